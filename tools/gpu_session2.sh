#!/bin/bash
# Round-3 GPU session 2: full GPU suite on the new build, kernel timeline of the two-stream verify protocol, default bench.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s2; mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
for WL in verify var_base; do
  rocprofv3 --output-format csv --kernel-trace -d $O/trace2s_$WL -o t -- python3 bench.py --workload $WL --steps 16 --no-cpu-baseline --no-also --no-strong > $O/trace2s_$WL.log 2>&1
  K=bjj_k_eddsa_verify\(; [ $WL == var_base ] && K=bjj_k_mul_var_base\(
  python3 tools/overlap_timeline.py $O/trace2s_$WL "$K" 20 > $O/timeline_$WL.txt 2>&1; cat $O/timeline_$WL.txt
  find $O/trace2s_$WL -name "*.csv" -size +3M -delete; find $O/trace2s_$WL -name "*.db" -delete
done
STEPS=40 bash tools/bench_all.sh verify var_base fixed_base point_add compress > $O/bench_all.txt 2>&1; cat $O/bench_all.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 2500 $O/bench_default.json
