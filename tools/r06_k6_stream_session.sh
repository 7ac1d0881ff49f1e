#!/bin/bash
# Developer experiment (round 6): the stream (= hardware queue) the scan + exact kernel of a device-pointer variable-base call run on.
cd ${GRAFT_REPO_ROOT:-.}
for w in 0 1 2 3; do echo "# BJJ_VB_EXACT_STREAM=$w (0 the set's scan stream, 1 copy-in stream, 2 copy-out stream, 3 second lane); exact kernel forced beside"; BJJ_VB_SPLIT=1 BJJ_VB_EXACT_STREAM=$w python3 tools/vb_beside_ab.py 2>&1 | grep -v amdgpu; done
