#!/bin/bash
# GPU box: run a list of steps, each under its own `timeout -k 10`, logging to gpurun_out/<tag>/<name>.{out,err}.
# A step that fails with an ordinary error does not stop the list; a step that TIMES OUT or is KILLED (rc 124 / 137) does --
# nothing further touches the GPU after that.   usage: tools/gpu_steps.sh <tag> <<'STEPS'
#   name|seconds|command ...
# STEPS
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; O=gpurun_out/$T; mkdir -p $O
: > $O/steps.log
while IFS='|' read -r name secs cmd; do
  [ -z "$name" ] && continue
  case "$name" in \#*) continue;; esac
  echo "[$(date +%T)] $name: $cmd" | tee -a $O/steps.log
  t0=$(date +%s)
  timeout -k 10 $secs bash -c "$cmd" > $O/$name.out 2> $O/$name.err
  rc=$?
  echo "[$(date +%T)] $name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "STOP: $name timed out / was killed" | tee -a $O/steps.log; exit 1; fi
done
echo "all steps done" | tee -a $O/steps.log
