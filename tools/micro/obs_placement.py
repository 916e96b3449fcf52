#!/usr/bin/env python3
"""Is the step's run time a function of WHERE its output buffers sit?  One env of 65 536 games, its obs / rew / done tensors replaced by views
into one arena at chosen offsets from a 2 MB boundary (torch's allocator hands out 2 MB-aligned blocks, so left alone ALL of a job's buffers
start on the same alignment); HIP-event time per step of a 100-step graph per placement.  (Round 4: two processes of the same 4v4 build
measured 20.3 and 22.4 us; rew 1 MB off the others' alignment was the fast case.)
    python tools/micro/obs_placement.py [n_agents_per_team]"""
import os, sys, json, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx

dev = torch.device("cuda:0")
n, E, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 65536, 100
A, D = 2 * n, 3 * n + 2
M = 2 << 20
arena = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
base = (-arena.data_ptr()) % M
acts = torch.randint(0, 4, (T, E, A), dtype=torch.int32, device=dev)
K = 1 << 10
cases = [(0, 0, 0), (0, 1024 * K, 0), (0, 1024 * K, 512 * K), (0, 512 * K, 0), (0, 256 * K, 0), (0, 64 * K, 0), (0, 4 * K, 0), (0, 1 * K, 0), (0, 256, 0),
         (1024 * K, 0, 0), (1024 * K, 1024 * K, 1024 * K), (512 * K, 1024 * K, 1536 * K), (0, 1024 * K + 4 * K, 8 * K), (0, 1280 * K, 0), (0, 768 * K, 0),
         (0, 1024 * K, 1024 * K), (0, 0, 1024 * K), (0, 0, 0)]
regions = [0, 64 << 20, 128 << 20]          # obs, rew, done live in their own 64 MB regions of the arena
for oo, ro, do in cases:
    env = bsx.parallel_env(n_agents=n, n_envs=E, seed=1234, auto_reset=True, device=dev)
    def view(region, off, nbytes, dtype, shape):
        return arena[base + region + off: base + region + off + nbytes].view(dtype).view(shape)
    env._obs = view(regions[0], oo, E * A * D * 4, torch.float32, (E, A, D))
    env._rew = view(regions[1], ro, E * A * 4, torch.float32, (E, A))
    env._done = view(regions[2], do, E * A, torch.uint8, (E, A))
    env._p_obs, env._p_rew, env._p_done = env._obs.data_ptr(), env._rew.data_ptr(), env._done.data_ptr()
    env.reset()
    g, _ = env.capture_steps(acts)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (2 * T) * 1e3)
    print(json.dumps({"n": n, "obs_off": oo, "rew_off": ro, "done_off": do, "us": round(sorted(ts)[2], 3), "state%2M": env._p_state % M}), flush=True)
    del env, g
