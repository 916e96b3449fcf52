#!/usr/bin/env python3
"""Condensed view of bench.py JSON lines: tools/show_bench.py gpurun_out/<tag>/<name>.out ..."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(f, "ERR", e); continue
    r = d["roofline"]
    print(f, "value %.2f G" % (d["value"] / 1e9), "ms/step", d["ms_per_step"], "launch_us", r["avg_launch_us"], "frac", r["frac"], "on_traffic", r.get("frac_on_traffic"), "live", r.get("live_bullets_per_agent"), "n_gpus", d["n_gpus"])
    for k, v in (d.get("other_workloads") or {}).items():
        print("    ", k[:58].ljust(58), v["avg_launch_us"], v["roofline_frac"], v.get("frac_on_traffic"), v["live_bullets_per_agent"])
    if d.get("multi_tick_launch"):
        print("     multi-tick us/tick", d["multi_tick_launch"]["us_per_tick"])
    if d.get("policy_rollouts"):
        for k, v in d["policy_rollouts"]["variants"].items():
            print("    ", k[:70].ljust(70), v.get("us_per_tick"), v.get("error"))
