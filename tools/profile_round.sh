#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the evidence bundle for one measurement series.
#   tools/profile_round.sh <tag>      e.g. r02_a
# 1) plain bench lines (default run with every workload; the driver's `--steps 20 --warmup 5` form)
# 2) rocprofv3 --kernel-trace --stats of both forms (kernel summary CSV)
# 3) per workload, separate --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with other trace domains), plus SQ_* / TCC
#    passes for the headline.  Output: gpurun_out/<tag>/...; condensed into profiles/ by tools/collect_profile.py <tag>.
# A step that times out stops the script: nothing else touches the GPU afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-rXX}
O=gpurun_out/$T
mkdir -p $O
trap 'for d in $O/pmc_*/ $O/stats $O/stats20; do rm -rf $d; done' EXIT   # (the raw counter CSVs are tens of MB: never left behind, however the script ends)
B="--no-cpu-baseline --no-other-workloads --no-live-traffic"   # (the runs below are themselves under rocprofv3: no nested passes)
echo "[$(date +%T)] bench lines"
timeout -k 10 400 python bench.py > $O/bench_default.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json
echo "[$(date +%T)] rocprof stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py $B > $O/bench_C2_under_rocprof.json 2> $O/stats.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats20 -- python bench.py --steps 20 --warmup 5 $B > $O/bench_steps20_under_rocprof.json 2> $O/stats20.err
pmc() {   # key, counters, bench args...
  local key=$1 ctr=$2; shift 2
  echo "[$(date +%T)] pmc $key $ctr"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex bsx_step --output-format csv -d $O/pmc_${key}_$(echo $ctr | cut -d' ' -f1) -- python bench.py --steps 300 --warmup 30 --repeats 2 $B "$@" > /dev/null 2> $O/pmc_${key}.err
}
# Counters are collected for the step kernels only (--kernel-include-regex), and every pass launches EAGERLY: one dispatch record per step.
# The bullet-heavy workload used to be the exception on both counts -- its pass prepared the closed-loop trajectory under the profiler
# (~480 ticks of dense_policy = ~21 000 small torch dispatches, each under counter collection) and then replayed a 300-node HIP graph --
# and it is the only pass that ever failed to finish: killed at its limit on two boxes of one afternoon (the record: gpurun_out/r05at, rc 124
# after 120 s, rocprofv3 alive at SIGTERM, nothing written; the step kernel is not implicated -- round 4's one-wave kernel just the same).
# It was not a slow pass: the same command finished in 5 ... 7 s in ten series before and after (r05_a ... r05_k: ~0.2 ms per dispatch under
# collection), so 120 s is 20 x what its dispatch count explains (profiles/r06_traced_runs.json).  What the records leave is an intermittent
# stall of the collection itself on one of the two things only this pass did; neither is needed to count a step's bytes, so neither happens
# under the profiler any more: the trajectory is recorded by an UNPROFILED run (--dense-record) and the profiled one replays it call by call.
for spec in "E65536_n1|--mode eager" "E65536_n4|--mode eager --n-agents 4" "E1048576_n1|--mode eager --envs-per-gpu 1048576 --steps 100" \
            "E65536_n1_cont|--mode eager --continuous" "E65536_n4_cont|--mode eager --continuous --n-agents 4" \
            "E65536_n1_many|--mode many"; do
  key=${spec%%|*}; args=${spec#*|}
  pmc $key FETCH_SIZE $args
  pmc $key WRITE_SIZE $args
done
pmc E65536_n1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --mode eager
pmc E65536_n1 "TCC_HIT_sum TCC_MISS_sum" --mode eager
pmc E65536_n4 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --mode eager --n-agents 4
echo "[$(date +%T)] bullet-heavy trajectory (unprofiled)"
timeout -k 10 120 python bench.py --steps 300 --warmup 30 --repeats 1 $B --action-mix dense --dense-record /tmp/bsx_dense_$T.pt > /dev/null
pmc E65536_n1_dense FETCH_SIZE --mode eager --action-mix dense --dense-replay /tmp/bsx_dense_$T.pt
pmc E65536_n1_dense WRITE_SIZE --mode eager --action-mix dense --dense-replay /tmp/bsx_dense_$T.pt
rm -f /tmp/bsx_dense_$T.pt
# condense on the box (the raw counter CSVs are tens of MB; only summaries travel back), then drop the raw directories
python tools/collect_profile.py $T --on-box > $O/collect.log 2>&1 || cat $O/collect.log
for d in $O/pmc_*/ $O/stats $O/stats20; do rm -rf $d; done
for f in $O/*.err; do [ -s $f ] && { echo "== $f"; tail -3 $f; }; done
echo "[$(date +%T)] profile_round done"
