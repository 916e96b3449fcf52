#!/usr/bin/env python3
"""GPU box: from which graph LENGTH do chains over game ranges pay when every replay is synchronised (a rollout of T ticks, then the
learner: the realistic use)?  A multi-branch graph costs more to launch than a linear one, per node; the chains' gain is per tick.
Wall clock per tick, replay + synchronise, median of `reps` replays, forms alternating.
    python tools/micro/chains_by_length.py steps|rollout E n P"""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor

kind, E, n, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
A, D = 2 * n, 3 * n + 2
out = {}
for T in (16, 32, 64, 128):
    runs = {}
    for chains in (1, P):
        env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234)
        env.reset()
        if kind == "steps":
            acts = torch.randint(0, 4, (T, E, A), dtype=torch.int32, device="cuda")
            g = env.capture_steps(acts, chains=chains)[0]
            runs[chains] = g.replay
        else:
            torch.manual_seed(0)
            actor = StackedActor(A, D, 4, device="cuda")
            with torch.no_grad():
                actor.w3.mul_(100.0)
            ro = PolicyRollout(env, actor, T, noise_std=0.1, chains=chains)
            ro.start(); ro.capture()
            runs[chains] = ro.run
        for _ in range(max(2, 300 // T)):
            runs[chains]()
        torch.cuda.synchronize()
    res = {c: [] for c in runs}
    for rep in range(30):
        for c, fn in runs.items():
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / T * 1e6)
    out[T] = {c: round(statistics.median(v), 2) for c, v in res.items()}
    del runs
    torch.cuda.empty_cache()
print(json.dumps({"what": f"{kind}, {E} x {n}v{n}: wall us per tick, ONE synchronised replay of a T-tick graph, chains 1 vs {P}", "by_T": out}))
