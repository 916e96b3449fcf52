#!/bin/bash
# Runs ON THE GPU BOX: PMC passes for the multi-tick launch (bench.py --mode many) and the one-launch rollout.
#   tools/profile_many.sh <tag>  ->  gpurun_out/<tag>/{many_*,rollout_*}
set -e
cd $GRAFT_REPO_ROOT
T=${1:-rXX}
O=gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python bench.py --mode many --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads --no-live-traffic"
R="python tools/bench_rollout.py --one-launch --reps 10"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/many_stats -- $P > $O/bench_C2_many_under_rocprof.json 2> $O/many_stats.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/many_pmc_fetch -- $P > /dev/null 2> $O/many_pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/many_pmc_write -- $P > /dev/null 2> $O/many_pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/many_pmc_sq -- $P > /dev/null 2> $O/many_pmc_sq.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rollout_stats -- $R > $O/rollout_under_rocprof.json 2> $O/rollout_stats.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/rollout_pmc_sq -- $R > /dev/null 2> $O/rollout_pmc_sq.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $O/rollout_pmc_mfma -- $R > /dev/null 2> $O/rollout_pmc_mfma.err
echo profile_many done
python - <<PY
import csv, glob, collections, json, os
O = "$O"
out = {}
for sub in sorted(glob.glob(os.path.join(O, "*_pmc_*"))):
    if not os.path.isdir(sub):
        continue
    cc = glob.glob(os.path.join(sub, "*", "*_counter_collection.csv"))
    if not cc:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if "bsx_step_kernel" in r["Kernel_Name"] and "true" in r["Kernel_Name"].split("<")[1].split(",")[2]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    out[os.path.basename(sub)] = {f"{k[1]}": {"mean_per_launch": sum(v) / len(v), "launches": len(v), "kernel": k[0]} for k, v in agg.items()}
json.dump(out, open(os.path.join(O, "many_rollout_pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    print(k, {a: round(b["mean_per_launch"]) for a, b in v.items()})
PY
