#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of the configs[4] rollouts (tools/bench_rollout.py): the graph of two kernels per tick and
# the one-launch kernels in their three precisions; kernel summaries -> gpurun_out/<tag>/rollout_<variant>_kernel_stats.csv
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-rXX}; O=gpurun_out/$T; mkdir -p $O
run() {   # name, args...
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roll_$name -- python tools/bench_rollout.py "$@" > $O/rollout_$name.json 2> $O/rollout_$name.err
  cp $O/roll_$name/*/*_kernel_stats.csv $O/rollout_${name}_kernel_stats.csv
  rm -rf $O/roll_$name
}
run graph
run one_launch --one-launch
run one_launch_bf16x6 --one-launch --precision bf16x6
run one_launch_bf16x3 --one-launch --precision bf16x3
run scripted_blue --one-launch --scripted-blue
echo "profile_rollouts done"
