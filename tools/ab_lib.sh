#!/bin/bash
# Developer tool: A/B several builds of libbjj_hip.so in one GPU session (interleaved rounds).  The builds are selected
# with BJJ_LIB_PATH -- the in-tree library is never overwritten.
# usage: [ROUNDS=2] [STEPS=8] tools/ab_lib.sh <alternative.so> [more.so ...] -- [workloads...]
cd ${GRAFT_REPO_ROOT:-.}
LIBS=(babyjubjub-rs_amd/csrc/libbjj_hip.so)
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for round in $(seq 1 ${ROUNDS:-2}); do
  for V in "${LIBS[@]}"; do
    echo "== round $round: $V"; BJJ_LIB_PATH=$(realpath $V) STEPS=${STEPS:-8} bash tools/bench_all.sh "$@" 2>&1 | grep -v amdgpu.ids
  done
done
