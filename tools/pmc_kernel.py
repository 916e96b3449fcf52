#!/usr/bin/env python3
"""Per-launch means of every counter a `rocprofv3 --kernel-trace --pmc ...` run collected, for the kernels whose name contains
<substr> (default bsx_step_kernel), second half of the launches (steady state); with SQ_WAVES present also per wave.
    python tools/pmc_kernel.py <rocprofv3 output dir> [substr]"""
import collections
import csv
import glob
import json
import os
import sys

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bsx_step_kernel")
out = {}
for cc in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        if sub in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), cs in agg.items():
        ent = {c: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for c, v in cs.items()}
        n = len(next(iter(cs.values())))
        w = ent.get("SQ_WAVES")
        row = {"launches": n, **{c: round(v, 1) for c, v in ent.items()}}
        if w:
            row["per_wave"] = {c: round(v / w, 1) for c, v in ent.items() if c != "SQ_WAVES"}
        out[f"{k} grid={g}"] = row
for kt in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    du = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        if sub in r["Kernel_Name"]:
            du[r["Kernel_Name"].split("(")[0][-60:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in du.items():
        v = v[len(v) // 2:]
        out.setdefault("kernel_ns_under_pmc", {})[k] = round(sum(v) / len(v), 1)
print(json.dumps(out, indent=1))
