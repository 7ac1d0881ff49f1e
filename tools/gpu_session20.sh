#!/bin/bash
# Round-3 GPU session 20: sustained clock / power per kernel at HEAD (one stream, launches back to back).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s20; mkdir -p $O
for K in fixed_base verify var_base poseidon5 sign decompress; do timeout 300 python3 tools/power_probe.py $K 5 2>/dev/null | head -4; done > $O/power_probe.txt 2>&1
cat $O/power_probe.txt
