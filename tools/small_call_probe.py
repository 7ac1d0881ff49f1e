#!/usr/bin/env python3
"""Developer probe: bjj_mul_var_base and bjj_poseidon5 on pinned host pointers over call sizes, the short-call kernels (k_small.hip: four lanes per item, six lanes
per hash) against K2 / K3 (BJJ_VB_QUAD_MAX=0, BJJ_P5_COOP_MAX=0) --
a fresh process per mode.  usage: small_call_probe.py [W]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, ctypes as C
sys.path.insert(0, %r)
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
N = 1 << 17
ctx = bjj.Context(0, int(sys.argv[1]))
pts = ctx.mul_fixed_base(w.scalars_254(N, offset=3))
sc = w.scalars_254(N, offset=900000)
h_p, h_s, h_o = ctx.host_empty(N * 64), ctx.host_empty(N * 32), ctx.host_empty(N * 64)
h_p[:] = pts.reshape(-1); h_s[:] = sc.reshape(-1)
row = []
for n in (1, 4, 16, 64, 256, 1024, 4096, 8192, 16384, 32768, 65536, 131072):
    f = lambda: ctx._ck(ctx.lib.bjj_mul_var_base(ctx.handle, h_p.ctypes.data, h_s.ctypes.data, C.c_size_t(n), h_o.ctypes.data), "vb")
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < 0.2: f()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    ts.sort()
    row.append("%%6d: %%7.1f (%%7.1f) form %%d" %% (n, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, ctx.info().last_var_base_form))
print("\n".join(row))
print("# bjj_eddsa_verify")
NV = 1 << 15
A, R, S, M = w.make_signatures(ctx.mul_fixed_base, ctx.poseidon5, NV)
hv = [ctx.host_empty(NV * k) for k in (64, 64, 32, 32)]; ov = ctx.host_empty(NV)
for b, a in zip(hv, (A, R, S, M)): b[:] = np.ascontiguousarray(a).reshape(-1)
for n in (1, 8, 64, 512, 2048, 4096, 8192, 16384, 32768):
    f = lambda: ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, hv[0].ctypes.data, hv[1].ctypes.data, hv[2].ctypes.data, hv[3].ctypes.data, C.c_size_t(n), ov.ctypes.data), "v")
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < 0.2: f()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    ts.sort()
    assert (np.asarray(ov[:n]) == 1).all()
    print("%%6d: %%7.1f (%%7.1f) form %%d" %% (n, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, ctx.info().last_verify_dispatch))
print("# bjj_sign")
hk, hm_, hr, hs, hok = ctx.host_empty(NV * 32), ctx.host_empty(NV * 32), ctx.host_empty(NV * 64), ctx.host_empty(NV * 32), ctx.host_empty(NV)
hk[:] = np.random.default_rng(6).integers(0, 256, NV * 32, dtype=np.uint8); hm_[:] = np.ascontiguousarray(M).reshape(-1)
for n in (1, 8, 64, 512, 2048, 4096, 8192, 16384):
    f = lambda: ctx._ck(ctx.lib.bjj_sign(ctx.handle, hk.ctypes.data, hm_.ctypes.data, C.c_size_t(n), hr.ctypes.data, hs.ctypes.data, hok.ctypes.data), "s")
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < 0.2: f()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    ts.sort()
    print("%%6d: %%7.1f (%%7.1f) form %%d" %% (n, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, ctx.info().last_sign_form))
print("# bjj_poseidon5")
h5 = ctx.host_empty(N * 160); h5[:] = np.random.default_rng(5).integers(0, 32, N * 160, dtype=np.uint8); o5 = ctx.host_empty(N * 32)
for n in (1, 8, 64, 512, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
    f = lambda: ctx._ck(ctx.lib.bjj_poseidon5(ctx.handle, h5.ctypes.data, C.c_size_t(n), o5.ctypes.data), "p5")
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < 0.2: f()
    ts = []
    for _ in range(15):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    ts.sort()
    print("%%6d: %%7.1f (%%7.1f) form %%d" %% (n, ts[len(ts) // 2] * 1e6, ts[0] * 1e6, ctx.info().last_poseidon_form))
''' % ROOT
W = sys.argv[1] if len(sys.argv) > 1 else "23"
for qmax in ("0", "1048576"):
    print("# BJJ_VB_QUAD_MAX = BJJ_P5_COOP_MAX = BJJ_VERIFY_SMALL_MAX = %s: microseconds per call, median (min)" % qmax, flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD, W], env=dict(os.environ, BJJ_VB_QUAD_MAX=qmax, BJJ_P5_COOP_MAX=qmax, BJJ_VERIFY_SMALL_MAX=qmax, BJJ_SIGN_SMALL_MAX=qmax), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    print(r.stdout.rstrip() or r.stderr[-800:], flush=True)
