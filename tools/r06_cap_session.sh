#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
echo "== default (cap 2^17)"; python3 tools/host_all_entries_probe.py 23 2>&1 | grep -v amdgpu
echo "== BJJ_PIPE_FIRST_CHUNK=32768 BJJ_PIPE_CHUNK=262144 (the cap until round 6)"; BJJ_PIPE_FIRST_CHUNK=32768 BJJ_PIPE_CHUNK=262144 python3 tools/host_all_entries_probe.py 23 2>&1 | grep -v amdgpu
