#!/bin/bash
# Round-3 GPU session 21: the signer's constant-time option -- parity (sign / Schnorr / public keys / C++ and Rust-shaped APIs),
# then what it costs.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s21; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_sign.py tests/test_schnorr.py tests/test_abi.py tests/test_gpu_cpp_api.py tests/test_gpu_reference_api.py tests/test_gpu_soak.py -x -q > $O/pytest_sign.log 2>&1; tail -5 $O/pytest_sign.log
STEPS=40 bash tools/bench_all.sh sign > $O/bench_sign.txt 2>&1
BENCH_ARGS=--signer-constant-time STEPS=40 bash tools/bench_all.sh sign >> $O/bench_sign.txt 2>&1
cat $O/bench_sign.txt
