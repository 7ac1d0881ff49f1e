#!/bin/bash
# Round-3 GPU session 23: full GPU suite on the build with the branch-free inversion and the constant-time signer option;
# one line per workload (did the inversion change cost anything?), sign with and without the option.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s23; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q -rs > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
STEPS=100 bash tools/bench_all.sh fixed_base > $O/bench_all.txt 2>&1; STEPS=40 bash tools/bench_all.sh verify var_base poseidon5 verify_compressed sign decompress point_add compress >> $O/bench_all.txt 2>&1
BENCH_ARGS=--signer-constant-time STEPS=40 bash tools/bench_all.sh sign | sed 's/^sign /sign_ct /' >> $O/bench_all.txt 2>&1
cat $O/bench_all.txt
timeout 900 python3 tools/soak_more.py 40000 60 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
