#!/bin/bash
# Developer session (round 6, after the last K2 edit): var-base profiles, library A/B, compressed fixed-base chunk sweep, tests.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06; mkdir -p $O
python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_host_pipeline.py tests/test_gpu_full_batches.py -x -q 2>&1 | tail -3
(echo "# this library vs the round-5 library (tools/ab_r05.so = a8a9861 rebuilt), interleaved, tools/ab_lib.sh"; ROUNDS=3 STEPS=10 bash tools/ab_lib.sh tools/ab_r05.so -- var_base) > $O/ab_r05_vs_r06_var_base.txt 2>&1; cat $O/ab_r05_vs_r06_var_base.txt
bash tools/r06_vb_ab_session.sh r06 2>&1 | grep -v "^\[k2\]" | tail -30
python3 tools/fb_compressed_sweep.py 23 2>&1 | tee $O/fb_compressed_sweep.txt
bash tools/profile_r.sh r06 var_base > $O/profile_r_var_base.log 2>&1; tail -2 $O/profile_r_var_base.log
bash tools/profile_headline.sh r06 var_base > $O/profile_headline_var_base.log 2>&1; tail -2 $O/profile_headline_var_base.log
