#!/usr/bin/env python3
"""Developer probe: device-side timeline (BJJ_PIPE_TRACE=1) of bjj_mul_var_base on pinned memory, 2^20 clean items -- set
BJJ_PIPE_VAR_BASE_SPLIT=0 for the round-5 form.  Prints the library's per-chunk lines of the LAST of a few calls and the wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BJJ_PIPE_TRACE"] = "1"
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

n = 1 << 20
every = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = bjj.Context(0, 16)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
pts = ctx.mul_fixed_base(sc.reshape(n, 32)).copy()
if every:
    pts[::every, 7] ^= 4
h_pts, h_sc, h_out = ctx.host_empty(n * 64), ctx.host_empty(n * 32), ctx.host_empty(n * 64)
h_pts[:] = pts.reshape(-1); h_sc[:] = sc
for k in range(6):
    sys.stderr.write("==== call %d\n" % k); sys.stderr.flush()
    t = time.perf_counter()
    ctx._ck(ctx.lib.bjj_mul_var_base(ctx.handle, h_pts.ctypes.data, h_sc.ctypes.data, n, h_out.ctypes.data), "vb")
    sys.stderr.write("==== call %d took %.3f ms\n" % (k, (time.perf_counter() - t) * 1e3)); sys.stderr.flush()
