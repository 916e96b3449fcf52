#!/bin/bash
# GPU box: calibrate FETCH_SIZE / WRITE_SIZE on THIS kernel's access pattern (MI355X_MICROARCH.md: "calibrate on a known
# byte count in your own access pattern before trusting an absolute").  Workload: 1 048 576 x 1v1, nobody ever shoots, so
# per launch the kernel reads exactly  E*(16 env + 16 counter line) + EA*(16 plane + 4 action)  = 72 B per game and writes
# EA*(16 plane + 20 obs + 4 rew + 1 done) + E*(16 env + 2 flags) = 100 B per game; the state (0.7 GB) exceeds the
# Infinity Cache.  Output: gpurun_out/calib/{fetch,write}/..., summarised by tools/collect_profile.py --calib.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/calib; mkdir -p $O
P="python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-workloads --no-live-traffic --mode eager --envs-per-gpu 1048576 --action-mix forward"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $P > /dev/null 2> $O/fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $P > /dev/null 2> $O/write.err
python - <<PY
import csv, glob
E = 1048576
for name, exp in (("fetch", 72 * E), ("write", 100 * E)):
    f = glob.glob(f"$O/{name}/*/*_counter_collection.csv")[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "bsx_step_kernel<1, false, false, false, false, true>" in r["Kernel_Name"] and int(r["Grid_Size"]) == 2 * E]
    v = v[len(v) // 2:]          # steady state: after the first game's bullets... none here, but skip warm-up launches
    m = sum(v) / len(v) * 1024
    print(f"{name}: counter {m/1e6:.1f} MB per launch, known {exp/1e6:.1f} MB  -> true/counter = {exp/m:.3f}")
PY
