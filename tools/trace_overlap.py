#!/usr/bin/env python3
"""How much do the step kernels of a `rocprofv3 --kernel-trace` run overlap in time?  For the dispatches whose name contains
<substr> (default bsx_step_kernel), second half of the trace (steady state): number of dispatches, mean duration, the time they
cover together (union of their [start, end] intervals), the sum of their durations, and concurrency = sum / union -- 1.0 for a
chain of dependent launches, above 1 when launches of different chains run at the same time (capture_steps(chains=P)).
    python tools/trace_overlap.py <rocprofv3 output dir> [substr]"""
import csv
import glob
import json
import os
import sys

d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bsx_step_kernel")
out = {}
for kt in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size") or r.get("Grid_Size_X") or "?") for r in csv.DictReader(open(kt)) if sub in r["Kernel_Name"])
    iv = iv[len(iv) // 2:]
    if not iv:
        continue
    union, cur_s, cur_e, overl = 0, iv[0][0], iv[0][1], 0
    for s, e, _ in iv[1:]:
        if s <= cur_e:
            overl += 1
            cur_e = max(cur_e, e)
        else:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
    union += cur_e - cur_s
    total = sum(e - s for s, e, _ in iv)
    span = iv[-1][1] - iv[0][0]
    grids = sorted({g for _, _, g in iv})
    out = {"dispatches": len(iv), "grid_sizes": grids, "mean_duration_ns": round(total / len(iv), 1), "sum_of_durations_us": round(total / 1e3, 1),
           "covered_us": round(union / 1e3, 1), "first_start_to_last_end_us": round(span / 1e3, 1), "concurrency": round(total / union, 3),
           "dispatches_starting_inside_another": overl}
print(json.dumps(out, indent=1))
