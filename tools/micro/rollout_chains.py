"""PolicyRollout's graph form (two kernels per tick: bsx_actor_forward -> bsx_step_*) with the games as P chains (chains=P):
µs per tick of the whole batch for 1v1 ... 4v4, both teams on actors.  The one-launch form is printed beside it where it exists."""
import argparse
import importlib
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
bsx = importlib.import_module("deep-rl-battlespace_amd")
ro_mod = importlib.import_module("deep-rl-battlespace_amd.rollout")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--teams", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--ticks", type=int, default=32)
    ap.add_argument("--continuous", action="store_true")
    ap.add_argument("--only-one-launch", action="store_true")
    args = ap.parse_args()
    dev, E, T = torch.device("cuda:0"), args.envs, args.ticks
    for n in args.teams:
        forms = ([] if args.only_one_launch else [("graph", P) for P in args.chains]) + [("one_launch", 1)]
        for form, P in forms:
            env = bsx.parallel_env(n_agents=n, n_envs=E, auto_reset=True, seed=1234, device=dev, continuous_actions=args.continuous)
            env.reset()
            torch.manual_seed(0)
            actor = ro_mod.StackedActor(2 * n, 3 * n + 2, 3 if args.continuous else 4, device=dev)
            with torch.no_grad():
                actor.w3.mul_(100.0)
            ro = ro_mod.PolicyRollout(env, actor, T, noise_std=0.1, precision=args.precision, one_launch=form == "one_launch", chains=P)
            ro.start(); ro.capture()
            for _ in range(8):
                ro.run()
            samples = []
            for _ in range(5):
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                for _ in range(10):
                    ro.run()
                torch.cuda.synchronize(dev)
                samples.append((time.perf_counter() - t0) / (10 * T) * 1e6)
            us = statistics.median(samples)
            print(f"n={n} {form} chains={P} precision={args.precision}: us_per_tick={us:.3f} agent_steps_per_s={E * 2 * n / us * 1e6:.3e} "
                  f"samples={[round(x, 2) for x in samples]}", flush=True)
            del ro, env, actor
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
