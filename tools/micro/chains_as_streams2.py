#!/usr/bin/env python3
"""GPU box: chains as P linear graphs on P streams WITH the fork / join a drop-in replacement of one graph needs (the side streams wait for
the caller's stream, the caller's stream waits for the side streams, per replay), queued (10 replays, then one synchronisation) and with
every replay synchronised.  us per tick by the wall clock, median of 15, forms alternating.
    python tools/micro/chains_as_streams2.py E n P [T]"""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd import _lib

E, n, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = int(sys.argv[4]) if len(sys.argv) > 4 else 100
A = 2 * n
forms = {}
acts = torch.randint(0, 4, (T, E, A), dtype=torch.int32, device="cuda")
nb = acts[0].numel() * acts.element_size()
keep = []
for name in ("1 chain", f"{P} chains, one graph", f"{P} linear graphs, fork/join per replay"):
    env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234)
    env.reset()
    if name == "1 chain":
        forms[name] = env.capture_steps(acts, chains=1)[0].replay
    elif "one graph" in name:
        forms[name] = env.capture_steps(acts, chains=P)[0].replay
    else:
        ranges = env.chain_ranges(P)
        streams = [torch.cuda.Stream() for _ in ranges]
        graphs = []
        for s, games in zip(streams, ranges):
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for t in range(T):
                    env._launch(acts.data_ptr() + t * nb, _lib.ACT_I32, False, None, env._p_obs, env._p_rew, env._p_done, games=games)
            graphs.append(g)
        keep.append((env, streams, graphs))

        def run(streams=streams, graphs=graphs):
            main = torch.cuda.current_stream()
            for s, g in zip(streams, graphs):
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    g.replay()
            for s in streams:
                main.wait_stream(s)
        forms[name] = run
    for _ in range(4):
        forms[name]()
    torch.cuda.synchronize()
res = {"queued x10": {k: [] for k in forms}, "synchronised": {k: [] for k in forms}}
for rep in range(15):
    for k, fn in forms.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        res["queued x10"][k].append((time.perf_counter() - t0) / (10 * T) * 1e6)
    for k, fn in forms.items():
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        res["synchronised"][k].append((time.perf_counter() - t0) / T * 1e6)
print(json.dumps({"what": f"{E} x {n}v{n}, T = {T}: wall us per tick", **{m: {k: round(statistics.median(v), 2) for k, v in r.items()} for m, r in res.items()}}))
