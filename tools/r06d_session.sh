#!/bin/bash
# Round 6, last session at the final build: the short-call soak, the default and the driver-command bench records, the multi-GPU dry run.
#   gpurun --timeout 3600 -- bash tools/r06d_session.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r06d; mkdir -p $O
BJJ_SOAK_SHORT_VAR_BASE_SEEDS=1000:120 timeout 900 python3 -m pytest tests/test_gpu_soak.py -q -k short_var_base 2>&1 | tail -4 > $O/soak_short.txt; cat $O/soak_short.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cp bench_detail.json $O/bench_default_detail.json; wc -c $O/bench_default.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; echo "bench rc=$?"; cp bench_detail.json $O/bench_driver_command_detail.json; wc -c $O/bench_driver_command.json
bash tools/scale_session.sh r06 > $O/scale_session.log 2>&1; tail -12 $O/scale_session.log
timeout 300 python3 tools/small_call_probe.py 23 > $O/small_call_probe.txt 2>&1; cat $O/small_call_probe.txt
du -sh gpurun_out/r06d gpurun_out/r06_scale 2>/dev/null
