#!/usr/bin/env python3
"""Long-run parity soak (GPU box): the HIP step() path against the C oracle in the production configuration
(auto-reset, in-kernel Philox) for hundreds of millions of agent-steps.  Every call: dones and rewards must be equal,
observations within 1e-5 relative; every `--check` calls the complete game state (poses, hit points, bullet lists,
ticks, flags, counters) must be bit-identical.  Actions: uniform random with extra shooting, or the on-device scripted
opponent for both teams (decisive games).  Prints one JSON line; exit code 1 on any mismatch."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd import instinct
from oracle import cref

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536); ap.add_argument("--n-agents", type=int, default=1)
ap.add_argument("--steps", type=int, default=1500); ap.add_argument("--check", type=int, default=250)
ap.add_argument("--policy", choices=("random", "instinct"), default="random"); ap.add_argument("--seed", type=int, default=2024)
ap.add_argument("--wide", action="store_true", help="take the 64-bit-offset kernels (BSX_F_WIDE_OFFSETS)")
ap.add_argument("--many", type=int, default=0, help="K > 0: the HIP side runs K ticks per launch (bsx_step_many_discrete); random policy only")
ap.add_argument("--chains", type=int, default=0, help="P > 0 (with --many K): the HIP side replays ONE graph of K ticks whose launches are P chains over game ranges "
                                                      "(capture_steps(chains=P), bsx_step_discrete_range) instead of a K-tick launch")
args = ap.parse_args()
E, n, T = args.envs, args.n_agents, args.steps
A = 2 * n
env = bsx.parallel_env(n_agents=n, n_envs=E, seed=args.seed, auto_reset=True, wide_offsets=args.wide)
c = cref.CRefBatch(E, n_agents=n, seed=args.seed, auto_reset=True)
env.reset(); c.reset()
g = torch.Generator(device="cuda"); g.manual_seed(args.seed)
teams = [instinct.Team(env.possible_red, env.possible_blue, env), instinct.Team(env.possible_blue, env.possible_red, env)]
acts = torch.zeros((E, A), dtype=torch.int32, device="cuda")
t0 = time.time(); n_obs = n_exact = 0; max_rel = 0.0
K = args.many
if K and args.policy != "random":
    sys.exit("--many needs --policy random")
if args.chains and (not K or T % K):
    sys.exit("--chains needs --many K with K dividing --steps")
chunk = None
if args.chains:
    g_act = torch.zeros((K, E, A), dtype=torch.int32, device="cuda")
    g_graph, g_out = env.capture_steps(g_act, store=True, chains=args.chains)
for t in range(T):
    if args.policy == "instinct":
        for tm in teams:
            tm.write_actions(out=acts)
    else:
        acts = torch.randint(0, 4, (E, A), generator=g, device="cuda", dtype=torch.int32)
        acts = torch.where(torch.rand((E, A), generator=g, device="cuda") < 0.4, torch.ones_like(acts), acts)
    if K:
        if t % K == 0:                                   # K ticks of actions, one launch; compared tick by tick below
            k = min(K, T - t)
            ca = torch.randint(0, 4, (k, E, A), generator=g, device="cuda", dtype=torch.int32)
            ca = torch.where(torch.rand((k, E, A), generator=g, device="cuda") < 0.4, torch.ones_like(ca), ca)
            if args.chains:
                g_act.copy_(ca); g_graph.replay()
                chunk = (ca, g_out)
            else:
                chunk = (ca, env.step_many(ca, store=True))
        acts = chunk[0][t % K]
        obs, rew, done = (x[t % K] for x in chunk[1])
        if (t % args.check == args.check - 1 or t == T - 1) and (t % K != K - 1 and t != T - 1):
            sys.exit("--check must fall on launch boundaries (a multiple of --many)")
    else:
        obs, rew, done = env.step_batch(acts)
    co, cr, cd = c.step(acts.cpu().numpy())
    o = obs.cpu().numpy()
    if not np.array_equal(done.cpu().numpy(), cd) or not np.array_equal(rew.cpu().numpy().astype(np.float64), cr):
        print(json.dumps({"mismatch": "done/rew", "step": t})); sys.exit(1)
    diff = np.abs(o.astype(np.float64) - co)
    rel = diff / np.maximum(np.abs(co), 1e-30)
    bad = (diff > 1e-7) & (rel > 1e-5)
    if bad.any():
        print(json.dumps({"mismatch": "obs", "step": t, "count": int(bad.sum())})); sys.exit(1)
    n_obs += o.size; n_exact += int((o == co).sum())
    if t % args.check == args.check - 1 or t == T - 1:
        sh = {k: v.cpu().numpy() for k, v in env.export_state().items()}
        sc = c.export_state()
        for f in ("px", "py", "pdir", "php", "bhp", "tick", "env_done", "winner", "bl_live", "counters"):
            if not np.array_equal(sh[f], sc[f]):
                print(json.dumps({"mismatch": f, "step": t})); sys.exit(1)
        m = sc["bl_live"].astype(bool)
        for f in ("bl_x", "bl_y", "bl_dir"):
            if not np.array_equal(sh[f][m], sc[f][m]):
                print(json.dumps({"mismatch": f, "step": t})); sys.exit(1)
        print(f"step {t + 1}: ok ({time.time() - t0:.0f} s)", file=sys.stderr, flush=True)
cnt = env.counters().sum(0)
print(json.dumps({"soak": "ok", "policy": args.policy, "hip_launch": (f"one graph of {K} ticks, {args.chains} chains of range launches (bsx_step_discrete_range)" if args.chains else
                                 f"{K} ticks per launch (bsx_step_many_discrete)" if K else "one launch per step"), "envs": E, "n_per_team": n, "steps": T, "agent_steps": E * A * T,
                  "observation_values": n_obs, "observation_values_bit_identical": n_exact,
                  "games": int(cnt[0]), "ties": int(cnt[1]), "red_wins": int(cnt[2]), "blue_wins": int(cnt[3]),
                  "seconds": round(time.time() - t0, 1)}))
