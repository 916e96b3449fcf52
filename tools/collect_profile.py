#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into tracked files under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats kernel summary
  profiles/<tag>_bench_*.json          the bench lines
  profiles/<tag>_pmc_summary.json      per-launch means of the PMC counters for the step kernel
  profiles/traffic.json                HBM bytes per launch for bench.py's roofline.traffic (guide: FETCH_SIZE and
                                       WRITE_SIZE are in KiB, separate passes; on gfx950 FETCH_SIZE reads 1/2 of a wide
                                       coalesced stream, so the read side is doubled)
usage: tools/collect_profile.py <tag> [E n]"""
import csv, glob, json, os, shutil, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
E = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, os.path.join(dst, f"{tag}_{os.path.basename(f)}"))
ks = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
G = 2
while G < 2 * n:
    G *= 2
grid_threads = ((E + 64 // G - 1) // (64 // G)) * 64      # the step kernel's launch: one 64-lane workgroup per 64/G games
summary = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    cc = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not cc:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if f"bsx_step_kernel<{n if n <= 4 else 0}, false, false, false>" in r["Kernel_Name"] and int(r["Grid_Size"]) == grid_threads:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = glob.glob(os.path.join(src, sub, "*", "*_kernel_trace.csv"))
    du = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "bsx_step_kernel" in r["Kernel_Name"]]
    for k, v in agg.items():
        summary[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    summary[f"{sub}_avg_kernel_ns_under_pmc"] = sum(du) / max(1, len(du))
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
    fetch_kib, write_kib = summary["FETCH_SIZE"]["mean_per_launch"], summary["WRITE_SIZE"]["mean_per_launch"]
    tpath = os.path.join(dst, "traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    tj[f"E{E}_n{n}"] = {"hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024), "fetch_size_kib_raw": fetch_kib,
                       "write_size_kib_raw": write_kib, "series": tag,
                       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, KiB units; read side x2: calibrated on "
                               "this kernel with a known-bytes workload (profiles/r01_traffic_calibration.json: true/counter = "
                               "1.993 for FETCH_SIZE, 1.001 for WRITE_SIZE)"}
    json.dump(tj, open(tpath, "w"), indent=1)
# multi-tick launch and one-launch rollout (tools/profile_many.sh)
for sub, name in (("many_stats", "many_kernel_stats"), ("rollout_stats", "rollout_kernel_stats")):
    ks = glob.glob(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(dst, f"{tag}_{name}.csv"))
for f in ("many_rollout_pmc_summary.json", "rollout_graph.json", "rollout_one_launch.json", "rollout_graph_4v4.json",
          "rollout_graph_bf16x3.json", "rollout_one_launch_bf16x3.json", "rollout_graph_4v4_bf16x3.json"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
mp = os.path.join(src, "many_rollout_pmc_summary.json")
if os.path.exists(mp):
    mj = json.load(open(mp))
    if "FETCH_SIZE" in mj.get("many_pmc_fetch", {}) and "WRITE_SIZE" in mj.get("many_pmc_write", {}):
        ticks = 100                                          # bench.py --mode many: ticks per launch
        f_kib, w_kib = mj["many_pmc_fetch"]["FETCH_SIZE"]["mean_per_launch"], mj["many_pmc_write"]["WRITE_SIZE"]["mean_per_launch"]
        tpath = os.path.join(dst, "traffic.json")
        tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
        tj[f"E{E}_n{n}_many"] = {"hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024), "ticks_per_launch": ticks,
                                 "hbm_bytes_per_tick": int((2 * f_kib + w_kib) * 1024 / ticks), "fetch_size_kib_raw": f_kib,
                                 "write_size_kib_raw": w_kib, "series": tag,
                                 "note": "bench.py --mode many (bsx_step_many_discrete, 100 ticks per launch, every tick's outputs stored); "
                                         "same counters and corrections as the per-step entry"}
        json.dump(tj, open(tpath, "w"), indent=1)
print(json.dumps(summary, indent=1)[:1500])
