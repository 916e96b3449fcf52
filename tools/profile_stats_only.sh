#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): step 2 of tools/profile_round.sh alone -- rocprofv3 --kernel-trace --stats of the default bench
# command and of the driver's form -- keeping the kernel summaries and the percentiles of the step kernel's durations (tools/trace_hist.py).
#   tools/profile_stats_only.sh <tag> <kernel name substring>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-rXX}; KN=${2:-bsx_step_split}
O=gpurun_out/$T
mkdir -p $O
B="--no-cpu-baseline --no-other-workloads --no-live-traffic"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py $B > $O/bench_C2_under_rocprof.json 2> $O/stats.err
python tools/trace_hist.py $O/stats $KN > $O/kernel_trace_percentiles.txt
cp $O/stats/*/*_kernel_stats.csv $O/kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats20 -- python bench.py --steps 20 --warmup 5 $B > $O/bench_steps20_under_rocprof.json 2> $O/stats20.err
python tools/trace_hist.py $O/stats20 $KN > $O/kernel_trace_percentiles_steps20.txt
cp $O/stats20/*/*_kernel_stats.csv $O/kernel_stats_steps20.csv
rm -rf $O/stats $O/stats20
timeout -k 10 120 python bench.py $B > $O/bench_C2_plain.json
cat $O/kernel_trace_percentiles.txt $O/kernel_trace_percentiles_steps20.txt
echo "profile_stats_only done"
