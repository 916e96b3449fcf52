#!/usr/bin/env python3
"""GPU box: does a timed 20-step block cost more when consecutive blocks replay DIFFERENT graphs (bench.py --steps 20 walks through five
20-node graphs, one per 20-tick slice of its 100-tick action table) than when they replay the same one?  Wall clock between two
synchronisations, median of 200 blocks; kernel time by events."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx

E, K, NG = 65536, 20, 5
env = bsx.parallel_env(n_agents=1, n_envs=E, auto_reset=True, seed=1)
env.reset()
acts = torch.randint(0, 4, (NG * K, E, 2), dtype=torch.int32, device="cuda")
for t in range(130):
    env.step_batch(acts[t % (NG * K)])
torch.cuda.synchronize()
graphs = [env.capture_steps(acts[i * K:(i + 1) * K], chains=1)[0] for i in range(NG)]
for g in graphs:
    g.replay()
torch.cuda.synchronize()


def timed(pick, reps=200, between=None):
    tot = []
    for r in range(reps):
        if between:
            between()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); pick(r).replay(); torch.cuda.synchronize(); t2 = time.perf_counter()
        tot.append((t2 - t0) * 1e6)
    return statistics.median(tot)


ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
graphs[0].replay(); ev0.record(); [graphs[i % NG].replay() for i in range(10)]; ev1.record(); torch.cuda.synchronize()
print(f"kernel time per step (events over 10 replays): {ev0.elapsed_time(ev1) * 1e3 / (10 * K):.2f} us")
print(f"same graph every block        {timed(lambda r: graphs[0]):7.1f} us per block")
print(f"five graphs in turn           {timed(lambda r: graphs[r % NG]):7.1f} us per block")
big = env.capture_steps(acts, chains=1)[0]
big.replay(); torch.cuda.synchronize()
print(f"five graphs in turn, a 100-node graph replayed (untimed) between blocks {timed(lambda r: graphs[r % NG], between=big.replay):7.1f} us per block")
def evs():
    big.replay(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record(); big.replay(); b.record(); torch.cuda.synchronize()
print(f"... and an event bracket with a second 100-node replay, as bench.py's timed_blocks does  {timed(lambda r: graphs[r % NG], between=evs):7.1f} us per block")
print(f"same graph every block, again (after all of the above)  {timed(lambda r: graphs[0]):7.1f} us per block")
print(f"five graphs in turn, again    {timed(lambda r: graphs[r % NG]):7.1f} us per block")
import gc
gc.disable()
print(f"five graphs in turn, gc off   {timed(lambda r: graphs[r % NG]):7.1f} us per block")
st = torch.cuda.current_stream()
def timed_stream(pick, reps=200):
    tot = []
    for r in range(reps):
        st.synchronize()
        t0 = time.perf_counter(); pick(r).replay(); st.synchronize(); t2 = time.perf_counter()
        tot.append((t2 - t0) * 1e6)
    return statistics.median(tot)
print(f"five graphs in turn, stream.synchronize() instead of device  {timed_stream(lambda r: graphs[r % NG]):7.1f} us per block")
