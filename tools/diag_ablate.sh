#!/bin/bash
# Timing-only ablation builds (GPU box): tools/diag_ablate.sh "<bits> <bits> ..." -> gpurun_out/diag/*.json
# (bits: 1 obs math, 2 bullets, 4 resolve; results are WRONG with any bit set).  The ablated libraries are built as VARIANTS
# (deep-rl-battlespace_amd/csrc/variants/lib_diag<bits>.so, tools/build_variant.py) and selected with BSX_LIB_PATH +
# BSX_ALLOW_DIAG=1 for these runs only: the product library is never touched, whatever fails or times out here.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/diag
for d in ${1:-"1 2 4"}; do
  V=deep-rl-battlespace_amd/csrc/variants/lib_diag$d.so
  [ -f $V ] || python tools/build_variant.py diag$d -DBSX_DIAG=$d
  BSX_LIB_PATH=$V BSX_ALLOW_DIAG=1 timeout -k 10 120 python bench.py ${BMODE:-} --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads --no-live-traffic > gpurun_out/diag/diag${d}_C2.json
  BSX_LIB_PATH=$V BSX_ALLOW_DIAG=1 timeout -k 10 120 python bench.py ${BMODE:-} --steps 300 --warmup 100 --no-cpu-baseline --no-other-workloads --no-live-traffic --envs-per-gpu 1048576 > gpurun_out/diag/diag${d}_1M.json
done
for f in gpurun_out/diag/*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['value']/1e9,3), 'G/s', d['roofline']['avg_launch_us'], 'us')" $f; done
