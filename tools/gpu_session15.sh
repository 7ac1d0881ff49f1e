#!/bin/bash
# Round-3 GPU session 15: is K1's power (and hence its clock) a function of the BYTES a gather moves?  Timing-only builds that
# fetch 64 / 32 bytes of every 128-byte entry (wrong results) against the shipped build; sustained ms per launch, clock, power.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s15; mkdir -p $O
for i in 1 2; do
  for L in babyjubjub-rs_amd/csrc/libbjj_hip.so tools/ab_k1_gather64.so tools/ab_k1_gather32.so; do
    echo "== round $i: $L"
    BJJ_LIB_PATH=$(realpath $L) timeout 300 python3 tools/power_probe.py fixed_base 4 2>/dev/null | head -4
  done
done > $O/k1_gather_bytes.txt 2>&1
cat $O/k1_gather_bytes.txt
