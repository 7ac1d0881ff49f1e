#!/usr/bin/env python3
"""Developer probe: the fixed-base host entry points on pinned memory with and without the copy-in stage (BJJ_PIPE_ZERO_COPY_IN: the
first kernel of a chunk reads the caller's pinned scalars through their device mapping), 2^20 items over chunk schedules and small calls.
A fresh process per row (the knobs are read once).  usage: fb_zero_copy_in_ab.py [W]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
N = 1 << 20
ctx = bjj.Context(0, int(sys.argv[1]))
sc = np.ascontiguousarray(w.scalars_254(N)).reshape(-1)
h_sc, o32, o64 = ctx.host_empty(N * 32), ctx.host_empty(N * 32), ctx.host_empty(N * 64)
h_sc[:] = sc
def best(f, reps=11, warm_s=0.4):
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < warm_s: f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3
out = []
for n in [int(x) for x in sys.argv[3].split(",")]:
    c = best(lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base_compressed(ctx.handle, h_sc.ctypes.data, n, o32.ctypes.data), "c"))
    i = ctx.info(); cc, cz = i.last_host_chunks, i.last_host_zero_copy
    a = best(lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, n, o64.ctypes.data), "a"))
    i = ctx.info()
    print("%%-34s n %%7d  compressed %%.3f ms (median %%.3f, %%2d chunks, zc %%d) = %%6.1f M/s   affine %%.3f ms (median %%.3f, %%2d chunks, zc %%d) = %%6.1f M/s" %% (sys.argv[2], n, c[0], c[1], cc, cz, n / c[0] / 1e3, a[0], a[1], i.last_host_chunks, i.last_host_zero_copy, n / a[0] / 1e3))
''' % ROOT
W = sys.argv[1] if len(sys.argv) > 1 else "23"
SCHED = [(None, None), (1 << 15, 1 << 17), (1 << 15, 1 << 18), (1 << 16, 1 << 18), (1 << 16, 1 << 19), (1 << 17, 1 << 18), (1 << 17, 1 << 19), (1 << 18, 1 << 18)]
for rnd in range(2):
  for zi in ("0", "1"):
    print("# BJJ_PIPE_ZERO_COPY_IN=%s (round %d)" % (zi, rnd), flush=True)
    for first, cap in (SCHED if zi == "1" else SCHED[:3]):
      env = dict(os.environ, BJJ_PIPE_ZERO_COPY_IN=zi)
      label = "shipped schedule"
      if first:
          env["BJJ_PIPE_FIRST_CHUNK"], env["BJJ_PIPE_CHUNK"] = str(first), str(cap)
          label = "first 2^%d cap 2^%d" % (first.bit_length() - 1, cap.bit_length() - 1)
      sizes = "1048576" if first else "1,64,1024,4096,65536,1048576"
      r = subprocess.run([sys.executable, "-c", CHILD, W, label, sizes], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
      print(r.stdout.strip() or r.stderr[-400:], flush=True)
