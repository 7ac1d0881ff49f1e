#!/bin/bash
# Round-3 GPU session 8: tile-dispatched K2 -- smoke, A/B against the grid-strided form, then the full GPU suite.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s8; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import torch"
timeout 600 python3 -m pytest tests -m gpu -x -q -k "var_base or cfg3 or golden or wide or reference_api" > $O/pytest_k2.log 2>&1; grep -E "passed|failed|error" $O/pytest_k2.log | tail -3
grep -q " passed" $O/pytest_k2.log || exit 1
ROUNDS=3 STEPS=40 bash tools/ab_lib.sh tools/ab_k2_strided.so -- var_base > $O/ab_k2_tiles.log 2>&1; grep -E "^==|^var_base" $O/ab_k2_tiles.log
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
