#!/usr/bin/env python3
"""configs[4]: 65 536 games x 1v1 feeding an on-device actor, end-to-end agent-steps/s.
Prints one JSON line: rollout (actor + step, one HIP graph of T ticks), and for reference the env alone and the
actor alone at the same shapes.  Runs on the GPU box: python tools/bench_rollout.py [--envs E] [--n-agents n]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd.rollout import PolicyRollout, StackedActor

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536); ap.add_argument("--n-agents", type=int, default=1)
ap.add_argument("--T", type=int, default=32); ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--noise", type=float, default=0.1); ap.add_argument("--torch-actor", action="store_true")
ap.add_argument("--one-launch", action="store_true", help="all T ticks in one kernel (bsx_rollout_discrete; 1v1)")
ap.add_argument("--precision", default="f32", choices=("f32", "bf16x3", "bf16x6"), help="the actor's 64 x 64 layer")
ap.add_argument("--rollout-only", action="store_true", help="skip the env-alone and actor-alone graphs (counter passes: only the rollout's kernels run)")
ap.add_argument("--chains", default="1", help="graph form: the batch as this many chains of (actor -> step) pairs over game ranges, or 'auto' "
                                           "(PolicyRollout's default); 1 = one pair per tick over the whole batch: what the kernel-stats and traffic tools profile")
ap.add_argument("--scripted-blue", action="store_true", help="blue is the scripted opponent (instinct.Team), red the actor: main.py:119-122")
args = ap.parse_args()
E, n, T = args.envs, args.n_agents, args.T
A, D = 2 * n, 3 * n + 2
env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234)
env.reset()
torch.manual_seed(0)
actor = StackedActor(A, D, 4, device="cuda")
with torch.no_grad():
    actor.w3.mul_(100.0)
opp = None
if args.scripted_blue:
    from deep_rl_battlespace_amd import instinct
    opp = instinct.Team(env.possible_blue, env.possible_red, env)
ro = PolicyRollout(env, actor, T, noise_std=args.noise, fused=not args.torch_actor, one_launch=args.one_launch, precision=args.precision,
                   opponent=opp, chains=1 if args.one_launch or args.torch_actor or opp is not None else (args.chains if args.chains == "auto" else int(args.chains)))
ro.start(); ro.capture()


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t_roll = timed(ro.run, args.reps) / T
if args.rollout_only:
    print(json.dumps({"rollout_us_per_tick": round(t_roll * 1e6, 2), "rollout_agent_steps_per_s": round(E * A / t_roll, 1)}))
    sys.exit(0)
# env alone (same score-vector input path), actor alone
g_env, _ = env.capture_steps(ro.scores)
t_env = timed(g_env.replay, args.reps) / T
g_act = torch.cuda.CUDAGraph()
x = ro.obs[1]
with torch.no_grad():
    actor(x); torch.cuda.synchronize()
    with torch.cuda.graph(g_act):
        for t in range(T):
            if ro.fused is not None:
                ro.fused.forward_into(x, ro.scores[t], args.noise, seq=t)
            else:
                y = actor(x)
t_act = timed(g_act.replay, args.reps) / T
c = env.counters().sum(0)
print(json.dumps({"workload": f"{E} games x {n}v{n} + on-device actor (obs {D} -> 64 -> LN -> 64 -> LN -> 4, one per agent), T={T} ticks per graph",
                  "rollout_agent_steps_per_s": round(E * A / t_roll, 1), "rollout_us_per_tick": round(t_roll * 1e6, 2),
                  "env_only_us_per_tick": round(t_env * 1e6, 2), "actor_only_us_per_tick": round(t_act * 1e6, 2),
                  "actor": "torch ops" if args.torch_actor else f"fused HIP kernel (bsx_actor_forward), 64 x 64 layer in {args.precision}",
                  "launch": "one kernel for all T ticks (bsx_rollout_discrete)" if args.one_launch else "HIP graph of 2T kernels", "opponent": "blue = scripted instinct.Team, red = actor" if args.scripted_blue else "both teams act by their actors",
                  "noise_std": args.noise, "games_finished": int(c[0]), "ties": int(c[1]), "red_wins": int(c[2]), "blue_wins": int(c[3])}))
