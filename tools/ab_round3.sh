#!/bin/bash
# Round-3 GPU session 1: (a) quick parity of the new host code (scratch sets, device guard, 64-lane verify workgroups),
# (b) interleaved A/B of the per-lane table layouts and of the verify workgroup size, (c) K1 with one 1024-lane workgroup.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_ab; mkdir -p $O
python3 -m pytest tests -m gpu -x -q -k "boundary or parity" > $O/pytest_quick.log 2>&1; tail -5 $O/pytest_quick.log
ROUNDS=2 STEPS=10 bash tools/ab_lib.sh tools/ab_base256.so tools/ab_pn0.so tools/ab_pn1.so -- verify var_base > $O/ab_pniels_vb64.log 2>&1
cat $O/ab_pniels_vb64.log
ROUNDS=3 STEPS=100 bash tools/ab_lib.sh tools/ab_k1_1024.so -- fixed_base > $O/ab_k1_1024.log 2>&1
cat $O/ab_k1_1024.log
