#!/usr/bin/env python3
"""GPU box: what does ONE dependent kernel node of a HIP graph cost when the kernel does (almost) nothing?  The floor under
the per-step kernel: a 100-node graph of (a) a 1-element torch add, (b) the observe-only launch of this library on 65 536
games (reads the plane records, writes the observation rows: no bullets, no physics)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx


def per_node(fn, n=100, reps=30):
    g = torch.cuda.CUDAGraph()
    fn(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); ev0.record()
    for _ in range(reps):
        g.replay()
    ev1.record(); torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / (reps * n) * 1e3


x = torch.zeros(1, device="cuda")
print(f"1-element add          : {per_node(lambda: x.add_(1.0)):6.2f} us per graph node")
big = torch.zeros(131072 * 4, device="cuda")
print(f"2 MB elementwise add   : {per_node(lambda: big.add_(1.0)):6.2f} us per graph node")
env = bsx.parallel_env(n_agents=1, n_envs=65536, auto_reset=True, seed=1); env.reset()
print(f"bsx_observe, 65536 x 1v1: {per_node(lambda: env.observe('plane0')):6.2f} us per graph node")
