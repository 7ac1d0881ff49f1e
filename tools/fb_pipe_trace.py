#!/usr/bin/env python3
"""Developer tool: a few calls of bjj_mul_fixed_base_compressed (c) or bjj_mul_fixed_base (a), 2^20 items on pinned memory -- run it under
BJJ_PIPE_TRACE=1 for the device-side timeline of every chunk (tools/fb_pipe_trace.sh), or with a call count for min / median.
usage: fb_pipe_trace.py c|a [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
n = 1 << 20
ctx = bjj.Context(0, 23)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
h_sc, o32, o64 = ctx.host_empty(n * 32), ctx.host_empty(n * 32), ctx.host_empty(n * 64)
h_sc[:] = sc
which = sys.argv[1]
f = (lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base_compressed(ctx.handle, h_sc.ctypes.data, n, o32.ctypes.data), "c")) if which == "c" else (lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, n, o64.ctypes.data), "a"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ts = []
for _ in range(reps):
    t = time.perf_counter(); f(); ts.append((time.perf_counter() - t) * 1e3)
    if reps <= 4: print("call %.3f ms" % ts[-1], file=sys.stderr)
ts = sorted(ts[1:])
i = ctx.info()
print("%s: min %.3f ms  median %.3f ms  = %.1f M/s   (%d chunks, zero-copy %d, K1 shape %d)" % ("compressed" if which == "c" else "affine", ts[0], ts[len(ts) // 2], n / ts[0] / 1e3, i.last_host_chunks, i.last_host_zero_copy, i.last_fixed_base_shape))
