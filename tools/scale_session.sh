#!/bin/bash
# One command for the first box with more than one GPU: everything of the multi-GPU path that has never met
# real multi-rank RCCL, in order, each step under its own timeout, results under gpurun_out/<tag>_scale/ (what gpurun merges back; copy the
# products/*.json into profiles/ afterwards: tools/scale_collect.py <tag>).  Claims nothing by itself: the files it writes are the measurement.
#   gpurun --timeout 3000 -- bash tools/scale_session.sh r05          (on an N > 1 box)
# On a ONE-GPU box it runs as a DRY RUN of itself: the same steps with the transports that exist there (gloo ranks sharing the
# GPU for bench.py, hipMemcpyPeerAsync between 8 contexts on the one device for the native form, real RCCL with G = 1 for the
# failure path), a 16-bit table and small batches -- so that the script, the tools it calls and the environment overrides are
# tested before the day an 8-GPU node shows up.
#   8 ranks on one host: BJJ_BENCH_TELEMETRY_MS (hwmon polling period per rank, 0 = off) and BJJ_BENCH_WINDOW_BITS (table width
#   when the driver's fixed command line cannot carry --window-bits) are read by bench.py; both are recorded in the line.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=gpurun_out/${TAG}_scale; P=$O/products; mkdir -p $O $P
export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
NG=$(python3 -c "import torch; print(torch.cuda.device_count())")
STEP_TIMEOUT=${STEP_TIMEOUT:-900}
step() {  # step <name> <cmd...>
  local name=$1; shift
  echo "=== [$TAG scale] $name: $*"
  ( timeout $STEP_TIMEOUT "$@" ) > $O/$name.log 2> $O/$name.err
  local rc=$?
  echo "rc=$rc"; echo "{\"step\": \"$name\", \"rc\": $rc}" >> $O/steps.jsonl
  grep -v amdgpu.ids $O/$name.err | tail -5
  tail -c 400 $O/$name.log; echo
}
last_json() { grep '^{' $1 | tail -1; }
rm -f $O/steps.jsonl
echo "{\"gpus_visible\": $NG, \"dry_run\": $([ $NG -gt 1 ] && echo false || echo true)}" > $O/session.json
if [ "$NG" -gt 1 ]; then
  # ---- (1) the two tests that skip on one GPU
  step tests python3 -m pytest tests -m gpu -q -x -k "multi_all_devices or two_contexts"
  # ---- (2) the driver's bench at 1, 2, 4, 8 ranks (one process per GPU, RCCL)
  for N in 1 2 4 8; do
    [ $N -le $NG ] || continue
    step bench_n$N python3 bench.py --gpus $N --steps 20 --warmup 5
    last_json $O/bench_n$N.log > $P/${TAG}_scale_bench_n$N.json
  done
  DEVS=$(python3 -c "print(','.join(str(i) for i in range($NG)))")
  # ---- (3) one process, all GPUs, inside the C ABI: RCCL, pieces 1 and 4, even / ragged / n < G
  step native_cases python3 tools/native_multi_cases.py --devices $DEVS --transport rccl --out $P/${TAG}_scale_native_cases.json
  step native_bench python3 bench.py --gpus $NG --native-multi --transport rccl --chunks 4
  last_json $O/native_bench.log > $P/${TAG}_scale_native_bench.json
  # ---- (4) the failure path over live communicators: drain, ncclCommAbort, retired handle, fresh handle
  BJJ_LIB_PATH=$R/tests/hooks/libbjj_hip_hooks.so BJJ_MULTI_INJECT_FAIL_GROUP=2 step native_failure python3 tools/native_multi_cases.py --devices $DEVS --transport rccl --expect-failure --out $P/${TAG}_scale_native_failure.json
else
  export BJJ_BENCH_WINDOW_BITS=16 BJJ_BENCH_TELEMETRY_MS=20
  step tests python3 -m pytest tests -m gpu -q -x -k "multi_block_arithmetic or multi_dev_form_rccl_one_device"
  for N in 1 2 8; do
    BJJ_BENCH_SHARE_GPU=1 BJJ_BENCH_BACKEND=gloo step bench_n$N python3 bench.py --gpus $N --steps 5 --warmup 2 --batch 65536 --strong-total 262144 --no-cpu-baseline
    last_json $O/bench_n$N.log > $P/${TAG}_scale_dryrun_bench_n$N.json
  done
  step native_cases python3 tools/native_multi_cases.py --devices 0,0,0,0,0,0,0,0 --transport peer --window-bits 16 --per-gpu 32768 --out $P/${TAG}_scale_dryrun_native_cases.json
  step native_bench python3 bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --native-multi --transport peer --chunks 4 --batch 32768 --strong-total 1048576
  last_json $O/native_bench.log > $P/${TAG}_scale_dryrun_native_bench.json
  # real RCCL with one rank: the serial schedule's gather is group 1 of the call
  BJJ_LIB_PATH=$R/tests/hooks/libbjj_hip_hooks.so BJJ_MULTI_INJECT_FAIL_GROUP=1 step native_failure python3 tools/native_multi_cases.py --devices 0 --transport rccl --window-bits 16 --expect-failure --out $P/${TAG}_scale_dryrun_native_failure.json
fi
cp $O/steps.jsonl $P/${TAG}_scale_steps.jsonl; cp $O/session.json $P/${TAG}_scale_session.json
cat $O/steps.jsonl
