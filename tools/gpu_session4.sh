#!/bin/bash
# Round-3 GPU session 4: full GPU suite, MFMA constant-half microbenchmark, rocprofv3 refresh of all four kernels at HEAD,
# default bench line.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s4; mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 tools/ubench/mfma_const_mul > $O/mfma_const_mul.txt 2>&1; cat $O/mfma_const_mul.txt
for WL in fixed_base verify var_base poseidon5; do bash tools/profile_r.sh r03 $WL > $O/profile_$WL.log 2>&1; tail -3 $O/profile_$WL.log; done
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
