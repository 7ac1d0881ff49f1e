#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s10; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import torch"
timeout 900 python3 -m pytest tests -m gpu -x -q -k "parity or boundary or schnorr or codec or cfg4 or cfg5" > $O/pytest_verify.log 2>&1; grep -E "passed|failed|error" $O/pytest_verify.log | tail -3
grep -q " passed" $O/pytest_verify.log || exit 1
echo "---- 2^20"; ROUNDS=2 STEPS=40 bash tools/ab_lib.sh tools/ab_persistent.so -- verify 2>&1 | grep -E "^==|^verify"
echo "---- 2^22"; BENCH_ARGS="--batch 4194304 --batches 2 --streams 1" ROUNDS=2 STEPS=8 bash tools/ab_lib.sh tools/ab_persistent.so -- verify 2>&1 | grep -E "^==|^verify"
