#!/bin/bash
# Round-3 GPU session 13: the new bench-watchdog / all-devices tests; where K1's parked wave time goes (cached-gather control,
# SQ wait counters).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s13; mkdir -p $O
export TMPDIR=/tmp
BJJ_TEST_ALL_DEVICES_ON_ONE=1 timeout 900 python3 -m pytest tests/test_gpu_bench.py::test_bench_headline_survives_unfinished_optional_sections tests/test_gpu_boundary.py::test_multi_all_devices_scatter_gather tests/test_gpu_bench.py::test_bench_gpus_2_self_launch_shared_gpu tests/test_gpu_bench.py::test_bench_one_gpu_line_has_every_block -x -q > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
for i in 1 2; do
  timeout 300 python3 tools/power_probe.py fixed_base 4 2>/dev/null | head -3
  REPEAT=1 timeout 300 python3 tools/power_probe.py fixed_base 4 2>/dev/null | head -3
done > $O/k1_cached_gathers.txt 2>&1
cat $O/k1_cached_gathers.txt
rocprofv3 -L > $O/counters_avail.txt 2>&1
ARGS="bench.py --workload fixed_base --streams 1 --steps 100 --warmup 20 --no-cpu-baseline --no-also --no-strong"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace -d $O/pmc_wait -o pmc -- python3 $ARGS > $O/pmc_wait.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --kernel-trace -d $O/pmc_insts -o pmc -- python3 $ARGS > $O/pmc_insts.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r03_s13"
for d in ("pmc_wait", "pmc_insts"):
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "mul_fixed_base" not in row["Kernel_Name"]:
                continue
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        print(d, {k: "%.4g x %d" % (v[0] / v[1], v[1]) for k, v in acc.items()})
PY
find $O -name "*.db" -delete; find $O -name "*_kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +4M -delete
