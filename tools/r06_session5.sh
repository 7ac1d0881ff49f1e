#!/bin/bash
# Developer session (round 6): K6 on lane pairs -- parity first, then what it costs.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06; mkdir -p $O
python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_full_batches.py tests/test_gpu_boundary.py tests/test_gpu_soak.py tests/test_gpu_host_pipeline.py -x -q 2>&1 | grep -E "passed|failed|assert|Error" | tail -5
for e in "BJJ_VB_SPLIT=0" "BJJ_VB_SPLIT=1" ""; do echo "# env: ${e:-default: by the history of the context}"; env $e python3 tools/vb_beside_ab.py 2>&1 | grep -v amdgpu; done | tee $O/var_base_beside_ab_pair.txt
python3 tools/var_base_offcurve_probe.py 23 2>&1 | grep -v amdgpu | tee $O/offcurve_pair.txt
