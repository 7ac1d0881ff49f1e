#!/bin/bash
# Developer session (round 6): the host-pointer rows of the bench line, the off-curve probe and the tests of the host pipeline on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/${1:-r06f}; mkdir -p $O
python bench.py --no-cpu-baseline --no-strong > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo rc=$?
python - <<PY
import json
d=json.load(open("bench_detail.json"))
h=d["also"]["host_api"]
for k in ("fixed_base","fixed_base_compressed","var_base","verify"):
    print(k, "pinned %.3f ms (%.1f M/s)"%(h[k]["ms_per_call"], h[k]["value"]/1e6), "chunks", h[k]["chunks"], "vs dev", h[k].get("vs_device_one_launch"), "| pageable %.3f ms"%h["pageable"][k]["ms_per_call"])
print("value", d["value"], "compressed dev", d["also"]["fixed_base_compressed"]["value"])
PY
python tools/var_base_offcurve_probe.py 23 2>&1 | grep -v amdgpu | tee $O/offcurve_default.txt
python -m pytest tests/test_gpu_host_pipeline.py tests/test_gpu_round6.py -x -q 2>&1 | tail -3
