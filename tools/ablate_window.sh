#!/bin/bash
# Developer tool: fixed-base throughput vs window width W (table size / gather traffic trade-off),
# plus the PCIe-inclusive rate of the host-pointer API.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for W in ${WIDTHS:-12 14 16 17 18 20 21 23 26}; do
  python3 bench.py --workload fixed_base --window-bits $W --steps 10 --warmup 2 --no-cpu-baseline --no-also --no-strong 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); c=d['config']; print('W=%2d  windows=%2d  table %8.1f MB  %8.2f M/s  kernel %.3f ms  parity %s' % (c['window_bits'], -(-252//c['window_bits']), c['fixed_base_table_mb'], d['value']/1e6, d['roofline']['kernel_ms_avg'], d['parity_sample_ok']))
"
done
python3 - <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
ctx = bjj.Context(0)
n = 1 << 20
sc = w.scalars_254(n)
ctx.mul_fixed_base(sc)
ts = []
for _ in range(5):
    t = time.perf_counter(); ctx.mul_fixed_base(sc); ts.append(time.perf_counter() - t)
print("host-pointer API (pageable H2D 32 MB + kernel + D2H 64 MB): best %.2f ms -> %.1f M/s PCIe-inclusive" % (min(ts) * 1e3, n / min(ts) / 1e6))
A = ctx.mul_fixed_base(sc); S = sc; m = w.random_u256(w.SEED_MSGS, n, 0, 3)
ctx.eddsa_verify(A, A, S, m)
ts = []
for _ in range(3):
    t = time.perf_counter(); ctx.eddsa_verify(A, A, S, m); ts.append(time.perf_counter() - t)
print("verify host-pointer API (192 MB H2D + kernel + 1 MB D2H): best %.2f ms -> %.1f M/s PCIe-inclusive" % (min(ts) * 1e3, n / min(ts) / 1e6))
PY
