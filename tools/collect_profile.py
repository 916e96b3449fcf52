#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv, <tag>_kernel_stats_steps20.csv   rocprofv3 --kernel-trace --stats kernel summaries
  profiles/<tag>_bench_*.json                                       the bench lines
  profiles/<tag>_pmc_summary.json                                   per-launch means of every PMC counter, per workload
  profiles/traffic.json                                             HBM bytes per launch (per tick for the multi-tick launch) per
                                                                    workload, read by bench.py for roofline.traffic
Guide (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate passes; on
gfx950 FETCH_SIZE reads half of a wide coalesced stream, so the read side is doubled (calibrated on this kernel with a
known-bytes workload in round 1: profiles/r01_traffic_calibration.json, true/counter 1.993 and 1.001).
usage: tools/collect_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
if "--on-box" in sys.argv:                                    # GPU box: condense into gpurun_out/<tag>/condensed/ (raw CSVs are then deleted)
    dst = os.path.join(src, "condensed")
    os.makedirs(dst, exist_ok=True)
    if os.path.exists(os.path.join(ROOT, "profiles", "traffic.json")):
        shutil.copy(os.path.join(ROOT, "profiles", "traffic.json"), os.path.join(dst, "traffic.json"))
elif os.path.isdir(os.path.join(src, "condensed")):          # back home: the condensed files are the result
    dst = os.path.join(ROOT, "profiles")
    for f in glob.glob(os.path.join(src, "condensed", "*")):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
    print("copied", sorted(os.path.basename(f) for f in glob.glob(os.path.join(src, "condensed", "*"))))
    sys.exit(0)
else:
    dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
for sub, name in (("stats", "kernel_stats"), ("stats20", "kernel_stats_steps20")):
    ks = glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(dst, f"{tag}_{name}.csv"))


def kernel_filter(key):
    m = re.match(r"E(\d+)_n(\d+)(.*)", key)
    E, n, rest = int(m.group(1)), int(m.group(2)), m.group(3)
    cont, many = "_cont" in rest, "_many" in rest
    G = 2
    while G < 2 * n:
        G *= 2
    grid = ((E + 64 // G - 1) // (64 // G)) * 64
    narrow = E * 2 * n * 200 <= 0xFFFFFFFF
    name = (f"bsx_step_kernel<{n if n <= 4 else 0}, {'true' if cont else 'false'}, {'true' if many else 'false'}, false, false, "
            f"{'true' if narrow else 'false'}>")   # <N, CONT, MULTI, ACTOR, LG, OFF32>
    if n == 1 and not (cont and many) and E <= (65536 if many else (81920 if cont else 114688)):   # the two-wave 1v1 kernels (bsx_step_split.h): <LG, OFF32, MANY, CONT>, two waves per workgroup
        draw = not many and (cont or E <= 98304)               # <LG, OFF32, MANY, CONT, DRAW>: per-call launches of up to 98 304 games take the kernel whose geometry wave draws
        name, grid = (f"bsx_step_split_kernel<false, {'true' if narrow else 'false'}, {(2 if E > 32768 else 1) if many else 0}, {'true' if cont else 'false'}, "
                      f"{'true' if draw else 'false'}>"), grid * 2
    return name, grid, many


summary = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    key = re.match(r"pmc_(E\d+_n\d+(?:_[a-z]+)?)_", os.path.basename(d)).group(1)
    cc = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not cc:
        continue
    name, grid, many = kernel_filter(key)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if name in r["Kernel_Name"] and int(r["Grid_Size"]) == grid:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[len(v) // 2:]                                   # steady state: the second half of the run's launches
        summary[key][k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    kt = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))
    if kt:
        du = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if name in r["Kernel_Name"]]
        du = du[len(du) // 2:]
        if du:
            summary[key][f"avg_kernel_ns_under_pmc[{os.path.basename(d)}]"] = sum(du) / len(du)
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)

tpath = os.path.join(dst, "traffic.json")
tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
for key, cs in summary.items():
    if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    f_kib, w_kib = cs["FETCH_SIZE"]["mean_per_launch"], cs["WRITE_SIZE"]["mean_per_launch"]
    ent = {"hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024), "fetch_size_kib_raw": f_kib, "write_size_kib_raw": w_kib, "series": tag,
           "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, KiB units; read side x2 (MI355X_MICROARCH.md; calibrated "
                   "on this kernel in round 1: profiles/r01_traffic_calibration.json)"}
    if key.endswith("_many"):
        ticks = 100                                           # bench.py --mode many: ticks per launch
        ent.update(ticks_per_launch=ticks, hbm_bytes_per_tick=int((2 * f_kib + w_kib) * 1024 / ticks))
    tj[key] = ent
json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: round(v["mean_per_launch"], 1) if isinstance(v, dict) else round(v, 1) for c, v in cs.items()} for k, cs in summary.items()}, indent=1)[:4000])
