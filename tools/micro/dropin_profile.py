#!/usr/bin/env python3
"""GPU box: where a drop-in step() call spends its time (cProfile over 3000 calls of the 1v1 game)."""
import cProfile, os, pstats, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import deep_rl_battlespace_amd as bsx

random.seed(1234)
env = bsx.parallel_env(n_agents=1)
ids = env.possible_agents
acts = np.random.default_rng(1234).integers(0, 4, size=(20000, len(ids))).tolist()
env.reset()
def loop(k0, N):
    for k in range(k0, k0 + N):
        if env.env_done:
            env.reset()
        env.step({a: acts[k][i] for i, a in enumerate(ids)})
loop(0, 300)
pr = cProfile.Profile(); pr.enable(); loop(300, 3000); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
