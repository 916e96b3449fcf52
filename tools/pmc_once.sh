#!/bin/bash
# GPU box: one PMC pass of the one-launch rollout with the counters given in $1 -> stdout means per launch
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_once; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $O/p -- python tools/bench_rollout.py --one-launch --reps 10 ${BR_ARGS:-} > /dev/null 2> $O/err.txt
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$O/p/*/*_counter_collection.csv")[0])):
    if "bsx_step_kernel<1, false, true, true, false, true>" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
