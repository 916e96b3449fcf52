#!/usr/bin/env python3
"""GPU box: what a timed block of K = 20 steps (the driver's `bench.py --steps 20`) costs beyond its kernels, per launch form:
one graph replay of 20 nodes / 20 step_batch() calls / 20 raw C-ABI launches.  Wall clock between two synchronisations, median of 200."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd import _lib

E, K = 65536, 20
env = bsx.parallel_env(n_agents=1, n_envs=E, auto_reset=True, seed=1)
env.reset()
acts = torch.randint(0, 4, (K, E, 2), dtype=torch.int32, device="cuda")
for t in range(130):
    env.step_batch(acts[t % K])
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for t in range(K):
        env.step_batch(acts[t])
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for t in range(K):
            env.step_batch(acts[t])
torch.cuda.synchronize()

def timed(fn, reps=200):
    tot, call = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        tot.append((t2 - t0) * 1e6); call.append((t1 - t0) * 1e6)
    return statistics.median(tot), statistics.median(call)

def eager():
    for t in range(K):
        env.step_batch(acts[t])
ptrs = [acts[t].data_ptr() for t in range(K)]
def raw():
    for t in range(K):
        env._launch(ptrs[t], _lib.ACT_I32, False, None, env._p_obs, env._p_rew, env._p_done)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); ev0.record(); [g.replay() for _ in range(10)]; ev1.record(); torch.cuda.synchronize()
print(f"kernel time per step (events over 10 replays): {ev0.elapsed_time(ev1) * 1e3 / (10 * K):.2f} us")
for name, fn in (("one graph replay of 20 nodes", g.replay), ("20 step_batch() calls", eager), ("20 raw C-ABI launches", raw)):
    tot, call = timed(fn)
    print(f"{name:32s} block {tot:7.1f} us = {tot / K:5.2f} us per step; host time inside the call(s) {call:6.1f} us")
