#!/bin/bash
# Round-3 GPU session 24: rocprofv3 refresh of the four kernels at HEAD (branch-free inversion), default bench line.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s24; mkdir -p $O
export TMPDIR=/tmp
for WL in fixed_base verify var_base poseidon5; do bash tools/profile_r.sh r03 $WL > $O/profile_$WL.log 2>&1; tail -1 $O/profile_$WL.log; done
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
