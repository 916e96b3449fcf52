#!/usr/bin/env python3
"""Stress regime: every agent shoots every tick -> ~11 live bullets per agent, every ring slot in use.
Prints us/step and agent-steps/s for 65 536 x 1v1 and 4v4.  GPU box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
for n in (1, 4):
    E, A, G = 65536, 2 * n, 50
    env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1); env.reset()
    acts = torch.ones((G, E, A), dtype=torch.int32, device="cuda")
    g, _ = env.capture_steps(acts)
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (10 * G)
    live = env.export_state(("bl_live",))["bl_live"].float().sum(-1).mean().item()
    print(json.dumps({"workload": f"{E} x {n}v{n}, every agent shoots every tick", "us_per_step": round(dt * 1e6, 2),
                      "agent_steps_per_s": round(E * A / dt, 1), "mean_live_bullets_per_agent": round(live, 2)}))
