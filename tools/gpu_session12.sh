#!/bin/bash
# Round-3 GPU session 12: full GPU suite on the XCD-guard build, extended soak over every kernel form, bench lines.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s12; mkdir -p $O
export TMPDIR=/tmp
timeout 2000 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 1500 python3 tools/soak_more.py 1000 40 > $O/soak_w23.txt 2>&1; tail -3 $O/soak_w23.txt
W=28 timeout 1500 python3 tools/soak_more.py 2000 20 > $O/soak_w28.txt 2>&1; tail -3 $O/soak_w28.txt
STEPS=100 bash tools/bench_all.sh fixed_base > $O/bench_all.txt 2>&1; STEPS=40 bash tools/bench_all.sh verify var_base poseidon5 verify_compressed sign decompress point_add compress >> $O/bench_all.txt 2>&1; cat $O/bench_all.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
