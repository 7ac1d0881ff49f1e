#!/bin/bash
# Developer tool: rocprofv3 evidence for the protocol that makes `value` -- K timed launches alternating over two HIP streams
# (VERDICT r04 item 1).  Kernel trace first (begin / end timestamps of every dispatch: the concurrency span of the timed
# launches), then the --pmc passes of the same command (counters are per dispatch; the profiler serialises dispatches while it
# counts), and the un-profiled line of the same command on the same box before and after.
# usage: tools/profile_headline.sh <tag> [workload]     -> gpurun_out/<tag>/ts_<workload>_*
#   fixed_base (default): the driver's exact command  `python3 bench.py --gpus 1 --steps 20 --warmup 5`  (+ --no-cpu-baseline)
#   verify | var_base   : `python3 bench.py --workload <w> --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-strong`
TAG=${1:-r05}; WL=${2:-fixed_base}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
export TMPDIR=/tmp
python3 babyjubjub-rs_amd/srchash.py > $OUT/source_hash.txt
python3 -c "import sys, json; sys.path.insert(0, 'babyjubjub-rs_amd'); import srchash; json.dump(srchash.kernel_hashes(), open('$OUT/code_hashes.json', 'w'), indent=1)"   # machine-code fingerprint of every kernel of the library that runs here
if [ "$WL" = fixed_base ]; then
  ARGS="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"; SHORT="--no-also --no-strong"
else
  ARGS="bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-also --no-strong"; SHORT=""
fi
P=$OUT/ts_$WL
echo "python3 $ARGS" > ${P}_command.txt
python3 $ARGS > ${P}_unprofiled_before.json 2> ${P}_unprofiled_before.err
rocprofv3 --output-format csv --kernel-trace --stats -d ${P}_trace -o trace -- python3 $ARGS > ${P}_trace.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d ${P}_pmc_fetch -o pmc -- python3 $ARGS $SHORT > ${P}_pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d ${P}_pmc_write -o pmc -- python3 $ARGS $SHORT > ${P}_pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d ${P}_pmc_sq -o pmc -- python3 $ARGS $SHORT > ${P}_pmc_sq.log 2>&1
python3 $ARGS > ${P}_unprofiled_after.json 2> ${P}_unprofiled_after.err
find $OUT -name "*.db" -delete
find $OUT -name "pmc_kernel_trace.csv" -delete            # the PMC passes' own traces are not needed (counters are per dispatch)
find $OUT -name "trace_kernel_trace.csv" -exec gzip -f {} \;   # the evidence: a few thousand dispatches
du -sh $OUT
tail -c 300 ${P}_trace.log
