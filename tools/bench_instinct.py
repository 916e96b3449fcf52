#!/usr/bin/env python3
"""Eventful games: both teams played by the on-device scripted opponent, 65 536 x n-v-n,
(opponent kernel -> step kernel) x T in one HIP graph.  Decisive games (wins, deaths, kill chains) load the bullet and
resolve paths far more than uniform random play does.  Prints one JSON line.  Runs on the GPU box."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import deep_rl_battlespace_amd as bsx
from deep_rl_battlespace_amd import instinct, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536); ap.add_argument("--n-agents", type=int, default=1)
ap.add_argument("--T", type=int, default=50); ap.add_argument("--reps", type=int, default=40)
args = ap.parse_args()
E, n, T = args.envs, args.n_agents, args.T
A = 2 * n
env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234)
env.reset()
red = instinct.Team(env.possible_red, env.possible_blue, env)
blue = instinct.Team(env.possible_blue, env.possible_red, env)
acts = torch.zeros((E, A), dtype=torch.int32, device="cuda")


def tick():
    red.write_actions(out=acts); blue.write_actions(out=acts)
    env._launch(acts.data_ptr(), _lib.ACT_I32, False, None, env._obs.data_ptr(), env._rew.data_ptr(), env._done.data_ptr())


for _ in range(300):
    tick()                                   # de-synchronise the games (they all start at tick 0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(T):
        tick()
for _ in range(3):
    g.replay()
torch.cuda.synchronize(); c0 = env.counters().sum(0); t0 = time.perf_counter()
for _ in range(args.reps):
    g.replay()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (args.reps * T)
c = env.counters().sum(0) - c0
live = env.export_state(("bl_live",))["bl_live"].float().sum(-1).mean().item()
print(json.dumps({"workload": f"{E} games x {n}v{n}, instinct vs instinct on device, T={T} ticks per graph",
                  "agent_steps_per_s": round(E * A / dt, 1), "us_per_tick": round(dt * 1e6, 2),
                  "games": int(c[0]), "ties": int(c[1]), "red_wins": int(c[2]), "blue_wins": int(c[3]),
                  "decisive_fraction": round(float(c[2] + c[3]) / max(1, int(c[0])), 3), "mean_live_bullets_per_agent": round(live, 2)}))
