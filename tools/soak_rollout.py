#!/usr/bin/env python3
"""Long-run check (GPU box): the one-launch rollout (bsx_rollout_discrete) against the graph of separate kernels per tick,
full size (65 536 x 1v1), several configurations, many launches: every transition tensor and the final game state must be
bit-identical.  Prints one JSON line; exit code 1 on any mismatch."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd import instinct
from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536); ap.add_argument("--T", type=int, default=32); ap.add_argument("--runs", type=int, default=40)
args = ap.parse_args()
E, T = args.envs, args.T
torch.manual_seed(11)
actor = StackedActor(2, 5, 4, device="cuda")
with torch.no_grad():
    actor.w3.mul_(80.0); actor.g1.uniform_(0.5, 1.5); actor.h2.uniform_(-0.3, 0.3)
configs = {"gaussian f32": dict(noise_std=0.2), "ou bf16x3": dict(ou_scale=0.3, precision="bf16x3"),
           "scripted blue f32": dict(noise_std=0.2, scripted="blue"), "scripted red bf16x3": dict(noise_std=0.1, scripted="red", precision="bf16x3")}
out, t0 = {}, time.time()
for name, kw in configs.items():
    kw = dict(kw); scripted = kw.pop("scripted", None)
    ros = []
    for one in (False, True):
        env = bsx.parallel_env(n_agents=1, n_envs=E, seed=77, auto_reset=True); env.reset()
        opp = None
        if scripted == "blue": opp = instinct.Team(env.possible_blue, env.possible_red, env)
        if scripted == "red": opp = instinct.Team(env.possible_red, env.possible_blue, env)
        ro = PolicyRollout(env, actor, T, seed=3, one_launch=one, opponent=opp, **kw); ro.start(); ro.capture()
        ros.append(ro)
    a, b = ros
    for r in range(args.runs):
        a.run(); b.run(); torch.cuda.synchronize()
        ok = torch.equal(a.obs, b.obs) and torch.equal(a.scores, b.scores) and torch.equal(a.rew, b.rew) and torch.equal(a.done, b.done)
        if ok and a.ou is not None and scripted is None:
            ok = torch.equal(a.ou["state"], b.ou["state"])
        if not ok:
            print(json.dumps({"mismatch": name, "run": r})); sys.exit(1)
    sa, sb = a.env.export_state(), b.env.export_state()
    live = sa["bl_live"].bool()
    for k in sa:
        same = torch.equal(sa[k][live], sb[k][live]) if k in ("bl_x", "bl_y", "bl_dir") else torch.equal(sa[k], sb[k])
        if not same:
            print(json.dumps({"mismatch": name, "state": k})); sys.exit(1)
    c = b.env.counters().sum(0)
    out[name] = {"agent_steps": E * 2 * T * args.runs, "games": int(c[0]), "ties": int(c[1]), "red_wins": int(c[2]), "blue_wins": int(c[3])}
print(json.dumps({"soak": "ok", "envs": E, "ticks_per_launch": T, "launches": args.runs, "configs": out, "seconds": round(time.time() - t0, 1)}))
