#!/bin/bash
# Round 6, third session: the K1 host entry points without a copy-in stage for short calls and with half-slot chunk launches.
# usage: tools/r06c_session.sh [tests] [ab] [latency] [bench]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/r06c; mkdir -p $OUT
for step in "$@"; do case $step in
  tests)   timeout 2400 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/tests.txt; cat $OUT/tests.txt ;;
  ab)      timeout 900 python3 tools/fb_host_half_ab.py 23 2 10,15,16,17,18,19,20,21,22,23 > $OUT/fb_host_half_ab.txt 2>&1; cut -c1-420 $OUT/fb_host_half_ab.txt ;;
  latency) timeout 900 python3 tools/single_call_latency.py > $OUT/single_call_latency.txt 2>&1; cat $OUT/single_call_latency.txt ;;
  bench)   timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cp bench_detail.json $OUT/bench_default_detail.json; cat $OUT/bench_default.json ;;
esac; done
