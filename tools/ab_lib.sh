#!/bin/bash
# Developer tool: A/B two builds of libbjj_hip.so in one GPU session (interleaved rounds).
# usage: tools/ab_lib.sh <alternative .so/.bin> [workloads...]
cd ${GRAFT_REPO_ROOT:-.}
ALT=$1; shift
cp babyjubjub-rs_amd/csrc/libbjj_hip.so /tmp/base.so
for round in 1 2; do
  echo "== round $round: baseline"; cp /tmp/base.so babyjubjub-rs_amd/csrc/libbjj_hip.so; STEPS=8 bash tools/bench_all.sh "$@"
  echo "== round $round: alternative ($ALT)"; cp $ALT babyjubjub-rs_amd/csrc/libbjj_hip.so; STEPS=8 bash tools/bench_all.sh "$@"
done
cp /tmp/base.so babyjubjub-rs_amd/csrc/libbjj_hip.so
