#!/bin/bash
# Developer tool: bench + rocprofv3 kernel trace + HBM-traffic PMC passes on a GPU box.
# usage: tools/profile_r.sh <tag> [workload]
TAG=${1:-r01}; WL=${2:-fixed_base}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python3 babyjubjub-rs_amd/srchash.py > $OUT/source_hash.txt
python3 -c "import sys, json; sys.path.insert(0, 'babyjubjub-rs_amd'); import srchash; json.dump(srchash.kernel_hashes(), open('$OUT/code_hashes.json', 'w'), indent=1)"   # machine-code fingerprint of every kernel of the library that runs here   # the tree these counters describe (tools/summarize_profile.py, bench.py)
python3 bench.py --workload $WL --no-also --no-strong > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err; tail -c 3000 $OUT/bench_$WL.json
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --streams 1 --no-cpu-baseline --no-also --no-strong"   # same steps / warm-up as the default bench line; ONE stream: the profiler serialises launches anyway, and per-launch counters / durations mean one thing
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace_$WL -o trace -- python3 $ARGS > $OUT/trace_$WL.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$WL -o pmc -- python3 $ARGS > $OUT/pmc_fetch_$WL.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$WL -o pmc -- python3 $ARGS > $OUT/pmc_write_$WL.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq_$WL -o pmc -- python3 $ARGS > $OUT/pmc_sq_$WL.log 2>&1
find $OUT -name "*.csv" | head -30
# drop the bulky raw traces, keep stats + counter csv
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -size +2M -delete
du -sh $OUT
