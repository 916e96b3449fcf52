#!/usr/bin/env python3
"""GPU box: the per-step kernel on int32 actions vs float32 [4] score vectors holding the SAME uniform random actions (one-hot +-1):
what the score-vector encoding itself costs (16 B instead of 4 B per agent + the in-kernel arg-max)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx

E, T = 65536, 100
g = torch.Generator(device="cuda"); g.manual_seed(1)
ai = torch.randint(0, 4, (T, E, 2), generator=g, device="cuda", dtype=torch.int32)
lg = torch.nn.functional.one_hot(ai.long(), 4).float() * 2 - 1
for name, acts in (("int32", ai), ("scores", lg.contiguous())):
    env = bsx.parallel_env(n_agents=1, n_envs=E, auto_reset=True, seed=3); env.reset()
    graph, _ = env.capture_steps(acts)
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    graph.replay(); e0.record()
    for _ in range(10):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    live = float(env.export_state(("bl_live",))["bl_live"].float().sum()) / (E * 2)
    print(f"{name:7s}: {e0.elapsed_time(e1) / (10 * T) * 1e3:6.2f} us per step, {live:.2f} live bullets per plane")
