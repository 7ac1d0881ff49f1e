#!/bin/bash
# Round-3 GPU session 7: full GPU suite on the build with K1's two shapes, then the bench lines.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s7; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
STEPS=100 bash tools/bench_all.sh fixed_base > $O/bench_all.txt 2>&1; STEPS=40 bash tools/bench_all.sh verify var_base poseidon5 >> $O/bench_all.txt 2>&1; cat $O/bench_all.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
