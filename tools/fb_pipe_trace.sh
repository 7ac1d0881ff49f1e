#!/bin/bash
# Developer tool: device-side timeline (BJJ_PIPE_TRACE=1: event times behind every copy-in, chunk launch and copy-out) of the fixed-base host calls,
# 2^20 items on pinned memory, for the shipped schedules and for explicit chunk lists.  usage: tools/fb_pipe_trace.sh   (on a GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for half in 1 0; do for sched in "" "262144,262144,262144,262144" "131072,131072,131072,131072,131072,131072,131072,131072"; do for form in c a; do
  echo "== form $form BJJ_PIPE_K1_HALF=$half schedule ${sched:-shipped}"
  BJJ_PIPE_TRACE=1 BJJ_PIPE_K1_HALF=$half BJJ_PIPE_SCHEDULE=$sched python3 tools/fb_pipe_trace.py $form 2>&1 | grep -v "^\[pipe\]" | tail -14
done; done; done
