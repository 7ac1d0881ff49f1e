#!/bin/bash
# Round-3 GPU session 19: would a verify kernel WITHOUT Poseidon / Euclid (they would move into the scan kernel) run better at
# 3 waves per SIMD?  Timing-only builds (wrong verdicts): main part only at 2 and at 3 waves per SIMD, against the shipped kernel.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s19; mkdir -p $O
for i in 1 2; do
  for L in babyjubjub-rs_amd/csrc/libbjj_hip.so tools/ab_verify_main2.so tools/ab_verify_main3.so; do
    echo "== round $i: $L"
    BJJ_LIB_PATH=$(realpath $L) timeout 300 python3 tools/power_probe.py verify 5 2>/dev/null | head -2
  done
done > $O/verify_main_only.txt 2>&1
BJJ_LIB_PATH=$(realpath babyjubjub-rs_amd/csrc/libbjj_hip.so) timeout 300 python3 tools/power_probe.py poseidon5 4 2>/dev/null | head -2 >> $O/verify_main_only.txt
cat $O/verify_main_only.txt
