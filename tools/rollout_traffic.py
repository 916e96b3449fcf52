#!/usr/bin/env python3
"""GPU box: HBM bytes per TICK of the configs[4] rollouts (tools/bench_rollout.py --rollout-only), by the guide's recipe -- FETCH_SIZE and
WRITE_SIZE in SEPARATE `rocprofv3 --kernel-trace --pmc` passes, KiB units, read side x 2 on gfx950 -- per launch form:
the graph of two kernels per tick (actor kernel + step kernel: the two per-dispatch means are added) and the one-launch kernels
(per-dispatch mean / T).  Second half of each kernel's dispatches (steady state).
    python tools/rollout_traffic.py <tag>          -> gpurun_out/<tag>/rollout_traffic.json   (bench.py reads profiles/traffic.json:
    python tools/rollout_traffic.py --merge <tag>  copies the entries there as "rollout_<variant>", run in the build container)"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"graph": [], "one_launch": ["--one-launch"], "one_launch_bf16x6": ["--one-launch", "--precision", "bf16x6"],
            "one_launch_bf16x3": ["--one-launch", "--precision", "bf16x3"], "scripted_blue": ["--one-launch", "--scripted-blue"]}
T = 32

if sys.argv[1] == "--merge":
    tag = sys.argv[2]
    got = json.load(open(os.path.join(ROOT, "gpurun_out", tag, "rollout_traffic.json")))
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tp))
    for k, v in got.items():
        tj[f"rollout_{k}"] = dict(v, series=tag)
    json.dump(tj, open(tp, "w"), indent=1, sort_keys=True)
    print("merged", sorted(got))
    sys.exit(0)

tag = sys.argv[1]
out_dir = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(out_dir, exist_ok=True)
res = {}
for name, extra in VARIANTS.items():
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join("/tmp", f"bsx_rt_{name}_{ctr}")
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "tools", "bench_rollout.py"), "--rollout-only", "--reps", "10", "--T", str(T), *extra]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=280)
        files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if r.returncode != 0 or not files:
            per = {"error": f"{ctr} pass rc {r.returncode}: {(r.stderr or '')[-200:]}"}
            break
        by_kernel = collections.defaultdict(list)
        for row in csv.DictReader(open(files[0])):
            kn = row["Kernel_Name"]
            if row["Counter_Name"] == ctr and ("bsx_step_" in kn or "bsx_actor" in kn):    # bsx_step_kernel<...> and the two-wave bsx_step_split_kernel<...>
                by_kernel[kn.split("(")[0]].append(float(row["Counter_Value"]))
        per[ctr] = {k: (sum(v[len(v) // 2:]) / len(v[len(v) // 2:]), len(v)) for k, v in by_kernel.items() if len(v) >= 8}   # (reset-time launches etc. dropped)
        shutil.rmtree(d, ignore_errors=True)
    if "error" not in per and not any("bsx_step_" in k for k in per["FETCH_SIZE"]):
        per = {"error": f"no step kernel among the counted kernels of '{name}': {sorted(per['FETCH_SIZE'])}"}    # (a renamed kernel must not silently drop its bytes)
    if "error" in per:
        res[name] = per
        continue
    one = "--one-launch" in extra
    kib = {c: sum(m for m, _ in per[c].values()) / (T if one else 1) for c in ("FETCH_SIZE", "WRITE_SIZE")}
    res[name] = {"hbm_bytes_per_tick": int((2 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024),
                 "fetch_size_kib_raw_per_tick": round(kib["FETCH_SIZE"], 1), "write_size_kib_raw_per_tick": round(kib["WRITE_SIZE"], 1),
                 "kernels": {c: {k: {"mean_kib_per_dispatch": round(m, 1), "dispatches": n} for k, (m, n) in per[c].items()} for c in per},
                 "ticks_per_launch": T if one else 1,
                 "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB; read side x2 (MI355X_MICROARCH.md)"}
    print(name, res[name]["hbm_bytes_per_tick"], flush=True)
json.dump(res, open(os.path.join(out_dir, "rollout_traffic.json"), "w"), indent=1, sort_keys=True)
