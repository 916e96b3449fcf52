#!/usr/bin/env python3
"""GPU box: calls per second of the DROP-IN surface (one game, reference return types, stdlib random draws): what a maintainer
who only swaps the import gets.  Same loop as BASELINE.md section 3 / configs[0]: uniform random actions, reset on done."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import deep_rl_battlespace_amd as bsx

for n in (1, 4):
    random.seed(1234)
    env = bsx.parallel_env(n_agents=n)
    ids = env.possible_agents
    acts = np.random.default_rng(1234).integers(0, 4, size=(20000, len(ids))).tolist()
    env.reset()
    for k in range(200):
        if env.env_done:
            env.reset()
        env.step({a: acts[k][i] for i, a in enumerate(ids)})
    N = 3000
    t0 = time.perf_counter()
    for k in range(N):
        if env.env_done:
            env.reset()
        env.step({a: acts[k][i] for i, a in enumerate(ids)})
    dt = time.perf_counter() - t0
    print(f"{n}v{n}: {N / dt:8.0f} step() calls/s = {N * len(ids) / dt:9.0f} agent-steps/s  ({dt / N * 1e6:.0f} us per call)")
