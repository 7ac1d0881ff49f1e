#!/bin/bash
# Round-3 GPU session 5: the verify kernel with one group per workgroup (per-XCD slot queues): smoke with timeouts, parity, A/B.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r03_s5; mkdir -p $O
export TMPDIR=/tmp
cat > /tmp/diag.py <<'PY'
import faulthandler, sys
faulthandler.enable()
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
from conftest import Oracle
orc = Oracle()
ctx = bjj.Context(0, 16)
for n in (64, 1000, 70000, 300000):
    A, R, S, msg = w.make_signatures(orc.mul_fixed_base, orc.poseidon5, n)
    bad = w.corrupt(A, R, S, msg, n)
    for rep in range(3):
        ok = ctx.eddsa_verify(A, R, S, msg)
        print("verify", rep, n, (ok == (~bad).astype(np.uint8)).all(), int(ok.sum()), flush=True)
ctx.close()
print("closed", flush=True)
PY
python3 -c "import torch"
timeout 180 stdbuf -o0 -e0 python3 /tmp/diag.py > $O/diag.log 2>&1; echo "diag rc=$?"; grep -v amdgpu.ids $O/diag.log | tail -16
grep -q closed $O/diag.log || exit 1
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "parity or boundary or schnorr or codec or cfg4 or cfg5 or reference_api or soak" > $O/pytest_verify.log 2>&1; grep -E "passed|failed|error" $O/pytest_verify.log | tail -3
ROUNDS=3 STEPS=40 bash tools/ab_lib.sh tools/ab_persistent.so -- verify > $O/ab_groups.log 2>&1; grep -E "^==|^verify" $O/ab_groups.log
timeout 600 rocprofv3 --output-format csv --kernel-trace -d $O/trace2s_verify -o t -- python3 bench.py --workload verify --steps 16 --no-cpu-baseline --no-also --no-strong > $O/trace2s_verify.log 2>&1
python3 tools/overlap_timeline.py $O/trace2s_verify "bjj_k_eddsa_verify_groups" 400 > $O/timeline_verify_all.txt 2>&1; grep -c . $O/timeline_verify_all.txt
find $O -name "*.db" -delete
