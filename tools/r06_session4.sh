#!/bin/bash
# Developer session (round 6, after K6 became call-free / scratch-free): the off-curve records again, same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06; mkdir -p $O
bash tools/r06_vb_ab_session.sh r06 2>&1 | grep -v "^\[k2\]" > $O/vb_ab_session.log; cp gpurun_out/r06/offcurve_ab.txt $O/offcurve_ab_final.txt; cat $O/offcurve_ab_final.txt
for e in "BJJ_VB_SPLIT=0" "BJJ_VB_SPLIT=1" ""; do echo "# env: ${e:-default: by the history of the context}"; env $e python3 tools/vb_beside_ab.py 2>&1 | grep -v amdgpu; done | tee $O/var_base_beside_ab.txt
python3 -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
