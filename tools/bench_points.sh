#!/bin/bash
# bench helper (GPU box): the standard four measurement points -> gpurun_out/$1_*.json
set -e
cd $GRAFT_REPO_ROOT
T=${1:-x}
mkdir -p gpurun_out
timeout -k 10 120 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-other-workloads --no-live-traffic > gpurun_out/${T}_C2.json
timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads --no-live-traffic --n-agents 4 > gpurun_out/${T}_C3.json
timeout -k 10 120 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-other-workloads --no-live-traffic --envs-per-gpu 1048576 > gpurun_out/${T}_1M.json
timeout -k 10 120 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads --no-live-traffic --envs-per-gpu 16384 > gpurun_out/${T}_16k.json
for f in gpurun_out/${T}_*.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['value']/1e9,3), 'G/s', d['roofline']['avg_launch_us'], 'us frac', d['roofline']['frac'])" $f; done
# multi-tick launches (bsx_step_many_*) and the one-launch rollout at the same points
timeout -k 10 120 python bench.py --mode many --steps 2000 --warmup 200 --no-cpu-baseline --no-other-workloads --no-live-traffic > gpurun_out/${T}_C2_many.json
timeout -k 10 120 python bench.py --mode many --steps 1000 --warmup 100 --no-cpu-baseline --no-other-workloads --no-live-traffic --n-agents 4 > gpurun_out/${T}_C3_many.json
timeout -k 10 120 python bench.py --mode many --steps 400 --warmup 100 --no-cpu-baseline --no-other-workloads --no-live-traffic --envs-per-gpu 1048576 > gpurun_out/${T}_1M_many.json
for f in gpurun_out/${T}_*_many.json; do python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d['value']/1e9,3), 'G/s', d['roofline']['avg_launch_us'], 'us frac', d['roofline']['frac'])" $f; done
timeout -k 10 200 python tools/bench_rollout.py --one-launch > gpurun_out/${T}_rollout_one.json
python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['rollout_us_per_tick'], 'us/tick', round(d['rollout_agent_steps_per_s']/1e9,3), 'G/s')" gpurun_out/${T}_rollout_one.json
