#!/usr/bin/env python3
"""Developer probe: what if K1 read its scalars from, and stored its results to, PINNED HOST memory itself (no copy-in, no copy-out, one
launch)?  The *_dev entry points take any device-visible address; hipHostMalloc memory is one.  Beside it: the shipped host entry
points on the same arrays.  usage: fb_zero_copy_probe.py [W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

W = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n = 1 << 20
ctx = bjj.Context(0, W)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
h_sc, o32, o64, r32, r64 = (ctx.host_empty(n * 32), ctx.host_empty(n * 32), ctx.host_empty(n * 64), ctx.host_empty(n * 32), ctx.host_empty(n * 64))
h_sc[:] = sc
d_sc = torch.from_numpy(sc.copy()).cuda(); d32 = torch.empty(n * 32, dtype=torch.uint8, device="cuda"); d64 = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(4)]


def best(f, reps=15, warm_s=0.4):
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < warm_s: f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3


def host_c(): ctx._ck(ctx.lib.bjj_mul_fixed_base_compressed(ctx.handle, h_sc.ctypes.data, n, r32.ctypes.data), "c")
def host_a(): ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, n, r64.ctypes.data), "a")


def zc(compressed, src, dst, parts=1, nstreams=1):
    per = n // parts
    item = 32 if compressed else 64
    f = ctx.mul_fixed_base_compressed_dev if compressed else ctx.mul_fixed_base_dev
    def run():
        for p in range(parts):
            f(src + p * per * 32, per, dst + p * per * item, streams[p % nstreams].cuda_stream if nstreams else 0)
        if nstreams:
            for s in streams[:nstreams]: s.synchronize()
        ctx.sync()
    return run


host_c(); host_a()
rows = [("host entry, compressed (shipped)", host_c), ("host entry, affine (shipped)", host_a)]
for comp, dst_h, dst_d, name in ((True, o32, d32, "compressed"), (False, o64, d64, "affine")):
    H, D = h_sc.ctypes.data, d_sc.data_ptr()
    rows += [("%s: in HBM, out HBM (the kernel alone)" % name, zc(comp, D, dst_d.data_ptr())),
             ("%s: in HOST, out HBM" % name, zc(comp, H, dst_d.data_ptr())),
             ("%s: in HBM, out HOST" % name, zc(comp, D, dst_h.ctypes.data)),
             ("%s: in HOST, out HOST, 1 launch" % name, zc(comp, H, dst_h.ctypes.data)),
             ("%s: in HOST, out HOST, 2 launches 2 streams" % name, zc(comp, H, dst_h.ctypes.data, 2, 2)),
             ("%s: in HOST, out HOST, 4 launches 2 streams" % name, zc(comp, H, dst_h.ctypes.data, 4, 2)),
             ("%s: in HOST, out HOST, 8 launches 4 streams" % name, zc(comp, H, dst_h.ctypes.data, 8, 4))]
for rnd in range(2):
    print("# round %d" % rnd)
    for label, f in rows:
        b, m = best(f)
        print("%-58s %.3f ms (median %.3f) = %7.1f M/s" % (label, b, m, n / b / 1e3), flush=True)
zc(True, h_sc.ctypes.data, o32.ctypes.data)(); zc(False, h_sc.ctypes.data, o64.ctypes.data)()
print("parity: zero-copy results == host entry results:", bool((o32 == r32).all()), bool((o64 == r64).all()))
