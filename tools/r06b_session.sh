mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -15 > gpurun_out/r06b/pytest_round6.txt; cat gpurun_out/r06b/pytest_round6.txt
(echo "# default forms"; python tools/var_base_offcurve_probe.py 23; echo "# round-5 forms: BJJ_VB_SPLIT=0 BJJ_PIPE_VAR_BASE_SPLIT=0"; BJJ_VB_SPLIT=0 BJJ_PIPE_VAR_BASE_SPLIT=0 python tools/var_base_offcurve_probe.py 23; echo "# forced beside: BJJ_VB_SPLIT=1"; BJJ_VB_SPLIT=1 python tools/var_base_offcurve_probe.py 23) > gpurun_out/r06b/offcurve.txt 2>&1; cat gpurun_out/r06b/offcurve.txt
