#!/bin/bash
# Round-3 GPU session 27: large launches queue behind the other scratch set (two streams never slower than one at >= 2^22).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s27; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "streams or large" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/throughput_vs_batch.sh > $O/throughput_vs_batch.txt 2>&1; cat $O/throughput_vs_batch.txt
