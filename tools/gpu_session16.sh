#!/bin/bash
# Round-3 GPU session 16: what the Euclid sequence (lattice_short_pair) costs inside the verify kernel -- timing-only build.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s16; mkdir -p $O
for i in 1 2; do
  for L in babyjubjub-rs_amd/csrc/libbjj_hip.so tools/ab_verify_noeuclid.so; do
    echo "== round $i: $L"
    BJJ_LIB_PATH=$(realpath $L) timeout 300 python3 tools/power_probe.py verify 5 2>/dev/null | head -3
  done
done > $O/verify_euclid_share.txt 2>&1
cat $O/verify_euclid_share.txt
