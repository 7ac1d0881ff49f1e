#!/bin/bash
# Round-3 GPU session 14: K1 with TWO gathers in flight behind the running addition (BJJ_K1_PREFETCH2=1, the in-tree build)
# against one (tools/ab_k1_prefetch1.so); parity of the new build first.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s14; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batches.py -m gpu -x -q -k "fixed or cfg2 or golden or config0 or table" > $O/pytest_k1.log 2>&1; tail -3 $O/pytest_k1.log
ROUNDS=3 STEPS=200 bash tools/ab_lib.sh tools/ab_k1_prefetch1.so -- fixed_base > $O/ab_k1_prefetch2.txt 2>&1; cat $O/ab_k1_prefetch2.txt
