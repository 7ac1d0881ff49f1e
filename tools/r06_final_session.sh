#!/bin/bash
# Developer session (round 6, final build): everything the round's records are made from, on ONE box.
#   gpurun --timeout 5000 -- bash tools/r06_final_session.sh [part ...]       parts: bench ab latency profiles scale (default: all)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06; mkdir -p $O
PARTS=${@:-bench ab latency profiles scale}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python3 bench.py > $O/bench_default_stdout.txt 2> $O/bench_default_stderr.txt; echo "bench rc=$?"; cp bench_detail.json $O/bench_default_detail.json
  wc -c $O/bench_default_stdout.txt
fi
if has ab; then
  (echo "# this library vs the round-5 library (tools/ab_r05.so = a8a9861 rebuilt), interleaved, tools/ab_lib.sh"; ROUNDS=2 STEPS=10 bash tools/ab_lib.sh tools/ab_r05.so -- fixed_base var_base verify) > $O/ab_r05_vs_r06.txt 2>&1
  cat $O/ab_r05_vs_r06.txt
  for e in "BJJ_VB_SPLIT=0" "BJJ_VB_SPLIT=1" ""; do echo "# env: ${e:-default: by the history of the context}"; env $e python3 tools/vb_beside_ab.py 2>&1 | grep -v amdgpu; done > $O/var_base_beside_ab.txt
  cat $O/var_base_beside_ab.txt
fi
if has latency; then
  python3 tools/single_call_latency.py 23 2>&1 | grep -v amdgpu > $O/single_call_latency.txt; cat $O/single_call_latency.txt
fi
if has profiles; then
  for WL in var_base fixed_base; do bash tools/profile_r.sh r06 $WL > $O/profile_r_$WL.log 2>&1; tail -3 $O/profile_r_$WL.log; done
  for WL in fixed_base var_base; do bash tools/profile_headline.sh r06 $WL > $O/profile_headline_$WL.log 2>&1; tail -3 $O/profile_headline_$WL.log; done
fi
if has scale; then
  bash tools/scale_session.sh r06 > $O/scale_session.log 2>&1; tail -15 $O/scale_session.log
fi
du -sh gpurun_out/r06 gpurun_out/r06_scale 2>/dev/null
