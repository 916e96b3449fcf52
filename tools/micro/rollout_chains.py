#!/usr/bin/env python3
"""GPU box: the configs[4] rollout as the graph of (actor -> step) kernel pairs, with the batch as 1 / 2 / 3 chains over game ranges
(PolicyRollout(chains=P)): does one range's kernel boundary hide behind the other's kernels at 1v1 too?  us per tick (wall clock over
`reps` replays of a T-tick graph, synchronised at both ends), forms alternating, medians.
    python tools/micro/rollout_chains.py [E] [n] [precision]"""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
A, D, T = 2 * n, 3 * n + 2, 32
torch.manual_seed(0)
actor = StackedActor(A, D, 4, device="cuda")
with torch.no_grad():
    actor.w3.mul_(100.0)
ros = {}
for P in (1, 2, 3):
    env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234)
    env.reset()
    ro = PolicyRollout(env, actor, T, noise_std=0.1, precision=prec, chains=P)
    ro.start(); ro.capture()
    for _ in range(10):
        ro.run()
    ros[P] = ro
torch.cuda.synchronize()
res = {P: [] for P in ros}
for rep in range(7):
    for P, ro in ros.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            ro.run()
        torch.cuda.synchronize()
        res[P].append((time.perf_counter() - t0) / 20 / T * 1e6)
print(json.dumps({"workload": f"{E} x {n}v{n}, graph of (actor -> step) pairs, {prec}, T = {T}", "us_per_tick_median": {P: round(statistics.median(v), 2) for P, v in res.items()},
                  "runs": {P: [round(x, 2) for x in v] for P, v in res.items()}}))
