#!/bin/bash
# Round-3 GPU session 17: full GPU suite at HEAD, rocprofv3 refresh of the four kernels, soak, final bench lines.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s17; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
for WL in fixed_base verify var_base poseidon5; do bash tools/profile_r.sh r03 $WL > $O/profile_$WL.log 2>&1; tail -1 $O/profile_$WL.log; done
timeout 1500 python3 tools/soak_more.py 3000 30 > $O/soak_w23.txt 2>&1; tail -2 $O/soak_w23.txt
STEPS=100 bash tools/bench_all.sh fixed_base > $O/bench_all.txt 2>&1; STEPS=40 bash tools/bench_all.sh verify var_base poseidon5 verify_compressed sign decompress point_add compress >> $O/bench_all.txt 2>&1; cat $O/bench_all.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
