#!/usr/bin/env python3
"""Developer tool: kernel time in consecutive windows from a cold start -- how long does the bench have to warm up before
the timed region sees the sustained (power / thermally limited) clocks?   usage: tools/clock_drift_probe.py <workload> [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import babyjubjub_rs_amd as bjj
import bench

kind = sys.argv[1] if len(sys.argv) > 1 else "fixed_base"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
ctx = bjj.Context(0, int(os.environ.get("W", "28")))
n = 1 << 20
ctx.reserve(n)
st = torch.cuda.Stream(device=dev)
wl = bench.Workload(ctx, kind, n, 0, dev, st, nb=4 if kind == "fixed_base" else 2)
time.sleep(3.0)   # let the GPU idle down first
per = 100 if kind == "fixed_base" else 8
t0 = time.perf_counter()
rows = []
k = 0
while time.perf_counter() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(per):
        wl.launch(k); k += 1
    e1.record(st)
    st.synchronize()
    rows.append((time.perf_counter() - t0, e0.elapsed_time(e1) / per))
edges = [0.25, 0.5, 1, 1.5, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 30]
lo = 0.0
for hi in edges:
    v = [ms for t, ms in rows if lo < t <= hi]
    if v:
        print("%s  t in (%5.2f, %5.2f] s: %8.4f ms per launch (%d samples)" % (kind, lo, hi, float(np.mean(v)), len(v)))
    lo = hi
ctx.close()
