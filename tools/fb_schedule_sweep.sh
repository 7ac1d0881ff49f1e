#!/bin/bash
# Developer sweep: explicit chunk lists (BJJ_PIPE_SCHEDULE) for the compressed fixed-base call on pinned memory; each in the sequence a,c,a,c of
# tools/fb_order_probe.py (the second c block = the fast copy-out state).
cd ${GRAFT_REPO_ROOT:-.}
for sch in "" \
  "32768,65536,131072,262144,262144,131072,65536,32768,32768" \
  "32768,65536,131072,262144,196608,131072,98304,65536,32768,32768" \
  "32768,65536,131072,262144,262144,163840,65536,32768" \
  "65536,131072,262144,262144,131072,98304,65536,32768" \
  "32768,65536,131072,196608,196608,196608,131072,65536,32768" \
  "32768,65536,131072,262144,327680,131072,65536,32768" ; do
  echo "== schedule: ${sch:-shipped}"
  BJJ_PIPE_SCHEDULE=$sch python3 tools/fb_order_probe.py ${1:-28} affine "a,c,a,c" 2>&1 | grep -v amdgpu | grep compressed | cut -c1-64
done
