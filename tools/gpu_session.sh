#!/bin/bash
# One GPU session = a list of shell commands run on the gpurun box, each with its own log under gpurun_out/<name>/.
# usage (from the container):  gpurun --timeout T -- bash tools/gpu_session.sh <name> '<cmd 1>' '<cmd 2>' ...
# Every command runs under its own `timeout` (STEP_TIMEOUT seconds, default 1800) so that a hang costs one step, not the box.
cd ${GRAFT_REPO_ROOT:-.}
NAME=$1; shift
O=gpurun_out/$NAME; mkdir -p $O
i=0
for CMD in "$@"; do
  i=$((i+1))
  echo "=== [$NAME step $i] $CMD"
  ( timeout ${STEP_TIMEOUT:-1800} bash -c "$CMD" ) > $O/step$i.log 2>&1
  echo "rc=$? ($(wc -l < $O/step$i.log) lines -> $O/step$i.log)"
  grep -v amdgpu.ids $O/step$i.log | tail -${TAIL:-40}
done
