#!/bin/bash
# Developer session (round 6): variable base on host pointers, this build's forms against the round-5 forms ON ONE BOX (boxes differ by
# several per cent), each twice, interleaved; then the device-side timeline of one call of each.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/${1:-r06g}; mkdir -p $O
for rep in 1 2; do
  echo "# round-5 forms (BJJ_VB_SPLIT=0 BJJ_PIPE_VAR_BASE_SPLIT=0), run $rep"; BJJ_VB_SPLIT=0 BJJ_PIPE_VAR_BASE_SPLIT=0 python tools/var_base_offcurve_probe.py 23 2>&1 | grep -v amdgpu
  echo "# this build, run $rep"; python tools/var_base_offcurve_probe.py 23 2>&1 | grep -v amdgpu
done | tee $O/offcurve_ab.txt
for f in new old; do
  if [ $f = old ]; then export BJJ_PIPE_VAR_BASE_SPLIT=0 BJJ_VB_SPLIT=0; fi
  python tools/vb_pipe_trace.py 2> $O/trace_$f.txt; echo "== $f"; awk "/==== call 4/,0" $O/trace_$f.txt | grep -v "^\[pipe\] "
done
