#!/usr/bin/env python3
"""Developer probe: bjj_mul_fixed_base_compressed (and the affine form beside it) on pinned memory, 2^20 items, over chunk schedules
(BJJ_PIPE_FIRST_CHUNK / BJJ_PIPE_CHUNK; a fresh process per schedule: the knobs are read once).  usage: fb_compressed_sweep.py [W]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
n = 1 << 20
ctx = bjj.Context(0, int(sys.argv[1]))
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
h_sc, o32, o64 = ctx.host_empty(n * 32), ctx.host_empty(n * 32), ctx.host_empty(n * 64)
h_sc[:] = sc
def best(f, reps=9, warm_s=0.5):
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < warm_s: f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3
c = best(lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base_compressed(ctx.handle, h_sc.ctypes.data, n, o32.ctypes.data), "c"))
cc = ctx.info().last_host_chunks
a = best(lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, n, o64.ctypes.data), "a"))
print("%%-28s compressed %%.3f ms (median %%.3f, %%2d chunks) = %%6.1f M/s   affine %%.3f ms (median %%.3f, %%2d chunks)" %% (sys.argv[2], c[0], c[1], cc, n / c[0] / 1e3, a[0], a[1], ctx.info().last_host_chunks))
''' % ROOT
W = sys.argv[1] if len(sys.argv) > 1 else "23"
for zc in ("0", "1"):
  print("# BJJ_FB_COMPRESSED_ZERO_COPY=%s" % zc, flush=True)
  os.environ["BJJ_FB_COMPRESSED_ZERO_COPY"] = zc
  for first, cap in [(None, None), (1 << 14, 1 << 16), (1 << 14, 1 << 17), (1 << 15, 1 << 16), (1 << 15, 1 << 17), (1 << 15, 1 << 18), (1 << 16, 1 << 17), (1 << 16, 1 << 18), (1 << 13, 1 << 17)]:
      env = dict(os.environ)
      label = "shipped schedule"
      if first:
          env["BJJ_PIPE_FIRST_CHUNK"], env["BJJ_PIPE_CHUNK"] = str(first), str(cap)
          label = "first 2^%d cap 2^%d" % (first.bit_length() - 1, cap.bit_length() - 1)
      r = subprocess.run([sys.executable, "-c", CHILD, W, label], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
      print(r.stdout.strip() or r.stderr[-400:], flush=True)
