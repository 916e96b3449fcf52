#!/usr/bin/env python3
"""Randomised differential run (GPU box): the HIP step() path against the C oracle over RANDOM configurations -- team size
1 ... 16, batch size (ragged, around the 256-game block boundaries), action encoding (int32 / score vectors / continuous
float32, float64, 4-wide rows), reward constants, env_offset, auto-reset or masked resets by hand, host-drawn jitter or in-kernel
Philox, 32- or 64-bit offset kernels, and every launch form (one call per step, a K-tick launch, a captured graph, a graph
whose launches are P chains over game ranges).  Per case: every call's rewards and flags equal, observations within 1e-5
relative, the complete game state bit-identical at the end (and at a few calls in between).

    python tools/fuzz_parity.py --seconds 240 --seed 1

Prints one JSON line (cases by feature, agent-steps, observation values compared / bit-identical); exit code 1 and the failing
case's parameters on the first mismatch."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from fuzz_util import draw_case, run_case, draw_dropin_case, run_dropin_case, draw_rollout_case, run_rollout_case   # the generators and checkers live with the tests (they drive the oracles)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0); ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-envs", type=int, default=6000); ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0 = by time)")
    ap.add_argument("--dropin-share", type=float, default=0.15, help="share of cases that play ONE game behind the reference's own surface "
                                                                     "(dict actions, reference return types, stdlib random) against the Python oracle")
    ap.add_argument("--rollout-share", type=float, default=0.1, help="share of cases that run a PolicyRollout in a random form and let the C oracle replay its games")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    by, total = {}, dict(cases=0, agent_steps=0, vals=0, exact=0)
    while (time.time() - t0 < args.seconds) and (not args.cases or total["cases"] < args.cases):
        r_kind = rng.random()
        dropin, rollout = r_kind < args.dropin_share, args.dropin_share <= r_kind < args.dropin_share + args.rollout_share
        case = draw_dropin_case(rng) if dropin else (draw_rollout_case(rng) if rollout else draw_case(rng, args.max_envs))
        try:
            bad, st = run_dropin_case(case) if dropin else (run_rollout_case(case) if rollout else run_case(case))
        except Exception as exc:                                 # an argument the build refuses is a finding too
            bad, st = f"{type(exc).__name__}: {str(exc)[:200]}", dict(vals=0, exact=0)
        if bad:
            print(json.dumps({"fuzz": "MISMATCH", "what": bad, "case": case})); sys.exit(1)
        total["cases"] += 1; total["vals"] += st["vals"]; total["exact"] += st["exact"]
        if dropin:
            total["agent_steps"] += 2 * case["n"] * case["T"]
            for key in ("drop-in surface", f"n={case['n']}", f"enc={case['encoding']}"):
                by[key] = by.get(key, 0) + 1
            continue
        if rollout:
            total["agent_steps"] += case["E"] * 2 * case["n"] * case["T"] * case["runs"]
            for key in ("policy rollout", f"rollout form={case['form']}", f"rollout noise={case['noise']}", f"rollout opponent={case['opponent']}",
                        f"rollout precision={case['precision']}", f"n={case['n']}"):
                by[key] = by.get(key, 0) + 1
            continue
        total["agent_steps"] += case["E"] * 2 * case["n"] * (case["T"] - case["T"] % case["K"])
        for key in (f"n={case['n']}", f"form={case['form']}", f"enc={case['enc']}", "wide" if case["wide"] else "narrow",
                    "auto_reset" if case["auto_reset"] else "masked_resets", "host_u" if case["host_u"] else "philox",
                    "checkpoint + resume in a new env" if case.get("resume") else "uninterrupted"):
            by[key] = by.get(key, 0) + 1
        if total["cases"] % 10 == 0:
            print(f"{total['cases']} cases, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    print(json.dumps({"fuzz": "ok", "seed": args.seed, "cases": total["cases"], "agent_steps": total["agent_steps"],
                      "observation_values": total["vals"], "observation_values_bit_identical": total["exact"],
                      "cases_by_feature": dict(sorted(by.items())), "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
