#!/bin/bash
# profiling helper (runs on the GPU box): ablation builds + PMC passes of the C2 bench.  Scratch output under gpurun_out/.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/diag
SRC=deep-rl-battlespace_amd/csrc
FLAGS="-O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -I include"
cp $SRC/libbattlespace_hip.so /tmp/product.so
for d in 1 2 4 7; do
  hipcc $FLAGS -DBSX_DIAG=$d $SRC/bsx_kernels.hip -o $SRC/libbattlespace_hip.so
  timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads > gpurun_out/diag/diag$d.json
done
cp /tmp/product.so $SRC/libbattlespace_hip.so
timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads > gpurun_out/diag/diag0.json
timeout -k 10 120 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-other-workloads --envs-per-gpu 1048576 > gpurun_out/diag/big1M.json
timeout -k 10 120 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-other-workloads --envs-per-gpu 262144 > gpurun_out/diag/big256k.json
timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads --envs-per-gpu 16384 > gpurun_out/diag/small16k.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --mode eager"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/diag/pmcA -- $P > /dev/null 2> gpurun_out/diag/pmcA.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/diag/pmcB -- $P > /dev/null 2> gpurun_out/diag/pmcB.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/diag/pmcC -- $P > /dev/null 2> gpurun_out/diag/pmcC.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/diag/pmcD -- $P > /dev/null 2> gpurun_out/diag/pmcD.err
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/diag/pmcE -- $P > /dev/null 2> gpurun_out/diag/pmcE.err
echo done
