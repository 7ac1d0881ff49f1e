#!/bin/bash
# Round-3 GPU session 18: long randomised differential soak of every entry point and every kernel form against the oracle.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s18; mkdir -p $O
( time timeout 1500 python3 tools/soak_more.py 20000 500 ) > $O/soak_w23_500.txt 2>&1; tail -5 $O/soak_w23_500.txt
( time W=28 timeout 900 python3 tools/soak_more.py 30000 200 ) > $O/soak_w28_200.txt 2>&1; tail -5 $O/soak_w28_200.txt
