#!/bin/bash
# Round-3 GPU session 6: verify with the scan on a high-priority stream (A/B vs persistent waves), K1 as 2 x 256 lanes on two streams.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s6; mkdir -p $O
export TMPDIR=/tmp
python3 -c "import torch"
timeout 900 python3 -m pytest tests -m gpu -x -q -k "parity or boundary or schnorr or codec or cfg4" > $O/pytest_verify.log 2>&1; grep -E "passed|failed|error" $O/pytest_verify.log | tail -3
ROUNDS=3 STEPS=40 bash tools/ab_lib.sh tools/ab_persistent.so -- verify > $O/ab_groups_prio.log 2>&1; grep -E "^==|^verify" $O/ab_groups_prio.log
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace2s_verify -o t -- python3 bench.py --workload verify --steps 16 --no-cpu-baseline --no-also --no-strong > $O/trace2s_verify.log 2>&1
find $O -name "*.db" -delete
echo "---- K1: one stream"; ROUNDS=2 STEPS=200 bash tools/ab_lib.sh tools/ab_k1_256x2.so -- fixed_base 2>&1 | grep -E "^==|^fixed"
echo "---- K1: two streams"; BENCH_ARGS="--streams 2" ROUNDS=2 STEPS=200 bash tools/ab_lib.sh tools/ab_k1_256x2.so -- fixed_base 2>&1 | grep -E "^==|^fixed"
