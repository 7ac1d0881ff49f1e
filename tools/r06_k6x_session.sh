#!/bin/bash
# Developer experiment (round 6): what about a K6 wave beside K2 costs K2 time?  tools/ab_k6x.so = this library with -DBJJ_K6_EXPERIMENT (k_var.hip):
#   BJJ_K6_XMODE=0 as shipped (wave priority 3)   1 no raised priority   2 resident, asleep ~4.9 ms (s_sleep), no arithmetic   3 resident, s_nop spin, no arithmetic
cd ${GRAFT_REPO_ROOT:-.}
for m in ${MODES:-0 1 2 3}; do echo "# BJJ_K6_XMODE=$m (exact kernel forced beside: BJJ_VB_SPLIT=1)"; BJJ_LIB_PATH=$(pwd)/tools/ab_k6x.so BJJ_VB_SPLIT=1 BJJ_K6_XMODE=$m python3 tools/vb_beside_ab.py 2>&1 | grep -v amdgpu; done
