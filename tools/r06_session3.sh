#!/bin/bash
# Developer session (round 6): after the default chunk cap went from 2^18 to 2^17 -- the call-order probe, the tests, the final default bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06; mkdir -p $O
for S in "a,c,a,c,a" "c,a,c,a" "a,a,a" "c,c,c"; do echo "== $S"; python3 tools/fb_order_probe.py 28 affine $S 2>&1 | grep -v amdgpu | cut -c1-64; done | tee $O/fb_order_after_cap17.txt
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 bench.py > $O/bench_default_stdout.txt 2> $O/bench_default_stderr.txt; echo "bench rc=$?"; cp bench_detail.json $O/bench_default_detail.json; wc -c $O/bench_default_stdout.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_stdout.txt 2> $O/bench_driver_cmd_stderr.txt; echo "bench (driver command) rc=$?"; cp bench_detail.json $O/bench_driver_cmd_detail.json; cat $O/bench_driver_cmd_stdout.txt
