#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the evidence bundle for one measurement series.
#   tools/profile_round.sh <tag>      e.g. r01_b
# 1) plain bench lines (C2 graph + cpu baseline, C3, 1M, eager)   2) rocprofv3 --kernel-trace --stats of the default
# bench command   3) separate --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with other trace domains).
# Output: gpurun_out/<tag>/...; copy what should be judged into profiles/ with tools/collect_profile.py.
set -e
cd $GRAFT_REPO_ROOT
T=${1:-rXX}
O=gpurun_out/$T
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_C2.json
timeout -k 10 200 python bench.py --mode eager --no-cpu-baseline --no-other-workloads > $O/bench_C2_eager.json
timeout -k 10 200 python bench.py --n-agents 4 --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads > $O/bench_C3.json
timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-other-workloads --envs-per-gpu 1048576 > $O/bench_1M.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline --no-other-workloads > $O/bench_C2_under_rocprof.json 2> $O/stats.err
P="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-other-workloads --mode eager"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P > /dev/null 2> $O/pmc_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- $P > /dev/null 2> $O/pmc_sq.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_tcc -- $P > /dev/null 2> $O/pmc_tcc.err
echo profile_round done
