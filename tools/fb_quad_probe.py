import os, subprocess, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
CHILD = r'''
import sys, time, ctypes as C
sys.path.insert(0, %r)
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
N = 1 << 16
ctx = bjj.Context(0, 23)
h_s, h_o, h_c = ctx.host_empty(N * 32), ctx.host_empty(N * 64), ctx.host_empty(N * 32)
h_s[:] = w.scalars_254(N, offset=3).reshape(-1)
for name, out in (("bjj_mul_fixed_base", h_o), ("bjj_mul_fixed_base_compressed", h_c), ("bjj_public_keys", h_o), ("bjj_public_keys_compressed", h_c)):
    row = []
    for n in (1, 64, 1024, 4096, 8192, 16384, 32768, 65536):
        f = lambda: ctx._ck(getattr(ctx.lib, name)(ctx.handle, h_s.ctypes.data, C.c_size_t(n), out.ctypes.data), name)
        t0 = time.perf_counter(); f()
        while time.perf_counter() - t0 < 0.15: f()
        ts = []
        for _ in range(15):
            t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
        ts.sort()
        row.append("%%d: %%.1f" %% (n, ts[len(ts) // 2] * 1e6))
    print("%%-32s %%s" %% (name, "   ".join(row)))
''' % ROOT
for q in ("0", "1048576"):
    print("# BJJ_FB_QUAD_MAX=%s: microseconds per call (median of 15), pinned" % q, flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BJJ_FB_QUAD_MAX=q), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print(r.stdout.rstrip() or r.stderr[-800:], flush=True)
