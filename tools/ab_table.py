#!/usr/bin/env python3
"""Condensed table of the bench lines a tools/gpu_steps.sh call left under gpurun_out/<tag>/: kernel time per launch (all
samples), wall time per step, live bullets per agent.     python tools/ab_table.py <tag>"""
import glob
import json
import os
import sys

d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", sys.argv[1])
for f in sorted(glob.glob(os.path.join(d, "*.out"))):
    name = os.path.basename(f)[:-4]
    txt = open(f).read().strip()
    if not txt.startswith("{") and "\n{" not in txt:
        print(f"{name:18s} {txt.splitlines()[-1][:150] if txt else '(empty)'}")
        continue
    try:
        j = json.loads(txt.splitlines()[-1])
    except Exception:
        print(f"{name:18s} unparsable; stderr: {open(f[:-4] + '.err').read()[-200:]}")
        continue
    if "roofline" not in j:                                  # (a step that printed some other JSON line)
        print(f"{name:18s} {txt.splitlines()[-1][:200]}")
        continue
    r = j["roofline"]
    print(f"{name:18s} us={r['avg_launch_us']:8.3f} {j['timing']['avg_launch_us_samples']} wall_us={j['ms_per_step'] * 1e3:8.3f} live={r['live_bullets_per_agent']}")
