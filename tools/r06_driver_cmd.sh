#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out/r06
S=$(date +%s.%N)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_cmd_stdout.txt 2> gpurun_out/r06/bench_driver_cmd_stderr.txt; echo rc=$?
E=$(date +%s.%N); echo "wall $(echo "$E - $S" | bc) s"
cp bench_detail.json gpurun_out/r06/bench_driver_cmd_detail.json; wc -c gpurun_out/r06/bench_driver_cmd_stdout.txt
python3 - <<PY
import json; d=json.loads(open("gpurun_out/r06/bench_driver_cmd_stdout.txt").read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["also"]["host_api"]["verify_compressed"])
PY
