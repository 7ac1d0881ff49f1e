#!/bin/bash
# Round 6: the whole GPU suite with the short-call kernels (csrc/k_small.hip) FORCED for every call size up to 2^20 items -- the existing parity corpus (golden vectors,
# every item of the 2^20-item batches, corruption, two streams, the host pipeline) through the several-lanes-per-item kernels.  Tests that assert WHICH form a call took are
# expected to fail (they are listed); everything else must pass.  Then the all-entry-point soak with more seeds at the shipped switch-overs.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06e; mkdir -p $O
BJJ_VB_QUAD_MAX=1048576 BJJ_P5_COOP_MAX=1048576 BJJ_VERIFY_SMALL_MAX=1048576 timeout 3000 python3 -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $O/forced_forms.txt; cat $O/forced_forms.txt
BJJ_SOAK_ALL_SEEDS=5000:100 timeout 1500 python3 -m pytest tests/test_gpu_soak.py -q -k all_entry_points 2>&1 | tail -4 > $O/soak_all.txt; cat $O/soak_all.txt
