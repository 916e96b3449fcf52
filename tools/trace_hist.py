#!/usr/bin/env python3
"""GPU box: percentiles of a kernel's durations and of the gaps between consecutive launches in a rocprofv3 --kernel-trace CSV.
usage: tools/trace_hist.py <dir with *_kernel_trace.csv> <kernel name substring>"""
import csv, glob, os, sys
import numpy as np
f = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"]]
rows.sort()
a = np.array(rows, dtype=np.int64)
du = a[:, 1] - a[:, 0]
gap = a[1:, 0] - a[:-1, 1]
sp = a[1:, 0] - a[:-1, 0]
q = [1, 5, 25, 50, 75, 95, 99]
print(f"{len(du)} launches of {sys.argv[2]}")
print(" duration ns  mean %.0f  pct %s: %s" % (du.mean(), q, np.percentile(du, q).round().tolist()))
print(" end->next start ns  pct: %s" % np.percentile(gap, q).round().tolist())
print(" start->start ns (< 20 us only) mean %.0f  pct: %s" % (sp[sp < 20000].mean(), np.percentile(sp[sp < 20000], q).round().tolist()))
half = len(du) // 2
print(" duration mean, first / second half of the run: %.0f / %.0f" % (du[:half].mean(), du[half:].mean()))
