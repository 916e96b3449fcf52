#!/usr/bin/env python3
"""GPU box: the chains as P LINEAR graphs replayed on P streams instead of one multi-branch graph (which launches node by node at
~6 us each): does that keep the chains' gain when every replay is synchronised?  Wall us per tick of one synchronised replay of T
ticks: 1 chain / P chains in one graph (capture_steps(chains=P)) / P linear graphs on P streams.
    python tools/micro/chains_as_streams.py E n P"""
import json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd import _lib

E, n, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
A = 2 * n
out = {}
for T in (20, 64, 128):
    forms = {}
    acts = torch.randint(0, 4, (T, E, A), dtype=torch.int32, device="cuda")
    nb = acts[0].numel() * acts.element_size()
    for name in ("1 chain", f"{P} chains, one graph", f"{P} linear graphs on {P} streams"):
        env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234)
        env.reset()
        if name == "1 chain":
            g = env.capture_steps(acts, chains=1)[0]
            forms[name] = g.replay
        elif "one graph" in name:
            g = env.capture_steps(acts, chains=P)[0]
            forms[name] = g.replay
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
            keep = (env, streams, graphs)

            def run(streams=streams, graphs=graphs):
                for s, g in zip(streams, graphs):
                    with torch.cuda.stream(s):
                        g.replay()
            forms[name] = run
        for _ in range(max(2, 300 // T)):
            forms[name]()
        torch.cuda.synchronize()
    res = {k: [] for k in forms}
    for rep in range(30):
        for k, fn in forms.items():
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / T * 1e6)
    out[T] = {k: round(statistics.median(v), 2) for k, v in res.items()}
print(json.dumps({"what": f"{E} x {n}v{n}: wall us per tick, one synchronised replay of T ticks", "by_T": out}))
