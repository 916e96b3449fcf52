#!/bin/bash
# Timing-only ablation builds (GPU box): tools/diag_ablate.sh "<bits> <bits> ..." -> gpurun_out/diag/*.json
# (bits: 1 obs math, 2 bullets, 4 resolve, 8 shot Philox+sincos; results are WRONG with any bit set;
#  suffix "L" = build WITH machine LICM; BMODE="--mode many" benches the multi-tick launch)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/diag
SRC=deep-rl-battlespace_amd/csrc
cp $SRC/libbattlespace_hip.so /tmp/product.so
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I include -c $SRC/bsx_actor.hip -o /tmp/actor.o
for d in ${1:-"8 1 2"}; do
  LICM="-mllvm -disable-machine-licm"; case $d in *L) LICM="";; esac
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-kernarg-preload-count=11 $LICM -DBSX_DIAG=${d%L} -I include -c $SRC/bsx_kernels.hip -o /tmp/k.o
  hipcc --offload-arch=gfx950 -shared -fPIC /tmp/k.o /tmp/actor.o -o $SRC/libbattlespace_hip.so
  timeout -k 10 120 python bench.py ${BMODE:-} --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads > gpurun_out/diag/diag${d}_C2.json
  timeout -k 10 120 python bench.py ${BMODE:-} --steps 300 --warmup 100 --no-cpu-baseline --no-other-workloads --envs-per-gpu 1048576 > gpurun_out/diag/diag${d}_1M.json
done
cp /tmp/product.so $SRC/libbattlespace_hip.so
for f in gpurun_out/diag/*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['value']/1e9,3), 'G/s', d['roofline']['avg_launch_us'], 'us')" $f; done
