#!/bin/bash
# Developer tool: compare several builds of libbjj_hip.so in one GPU session (interleaved rounds).
# usage: WL="fixed_base var_base" tools/ab_multi.sh <alt1.bin> <alt2.bin> ...
cd ${GRAFT_REPO_ROOT:-.}
LIB=babyjubjub-rs_amd/csrc/libbjj_hip.so
cp $LIB /tmp/base.so
for round in 1 2 3; do
  for V in /tmp/base.so "$@"; do
    echo "== round $round: $V"; cp $V $LIB; STEPS=${STEPS:-10} bash tools/bench_all.sh ${WL:-fixed_base} 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/base.so $LIB
