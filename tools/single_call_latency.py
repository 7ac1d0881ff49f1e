#!/usr/bin/env python3
"""Developer probe: what ONE call of the reference's single-item API costs through the library (the Rust crate routes
Point::mul_scalar / verify / Poseidon::hash to the batch entry points with n = 1, rust/src/babyjubjub_hip.rs), and small batches:
host-pointer entry points on pinned and pageable memory, median of many calls.  (The CPU side of the comparison is `cpu_baseline.single_thread_value` of the
bench lines under profiles/: this tool does not touch oracle/.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

W = int(sys.argv[1]) if len(sys.argv) > 1 else 23
ctx = bjj.Context(0, W)
N = 4096
A, R, S, M = w.make_signatures(ctx.mul_fixed_base, ctx.poseidon5, N)
sc = w.scalars_254(N, offset=5)
h5 = np.concatenate([R, A, M], axis=1)


def med(f, reps):
    f(); f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e6, float(np.min(ts)) * 1e6


def rows(n, pinned):
    alloc = ctx.host_empty if pinned else (lambda nb: np.zeros(nb, np.uint8))
    def buf(a):
        b = alloc(a[:n].size); b[:] = np.ascontiguousarray(a[:n]).reshape(-1); return b
    hA, hR, hS, hM, hsc, hh = buf(A), buf(R), buf(S), buf(M), buf(sc), buf(h5)
    o64, o32, ok = alloc(n * 64), alloc(n * 32), alloc(n)
    L, H = ctx.lib, ctx.handle
    out = {
        "mul_fixed_base": med(lambda: L.bjj_mul_fixed_base(H, hsc.ctypes.data, n, o64.ctypes.data), 200),
        "mul_var_base": med(lambda: L.bjj_mul_var_base(H, hA.ctypes.data, hsc.ctypes.data, n, o64.ctypes.data), 100),
        "poseidon5": med(lambda: L.bjj_poseidon5(H, hh.ctypes.data, n, o32.ctypes.data), 200),
        "eddsa_verify": med(lambda: L.bjj_eddsa_verify(H, hA.ctypes.data, hR.ctypes.data, hS.ctypes.data, hM.ctypes.data, n, ok.ctypes.data), 100),
    }
    assert bool(np.asarray(ok).all())
    if pinned:
        for b in (hA, hR, hS, hM, hsc, hh, o64, o32, ok):
            ctx.host_free(b)
    return out


print("W = %d; microseconds per CALL, median (min)" % W)
print("%-16s %-9s %18s %18s %18s %18s" % ("items per call", "memory", "mul_fixed_base", "mul_var_base", "poseidon5", "eddsa_verify"))
for n in (1, 64, 1024, 4096):
    for pinned in (True, False):
        r = rows(n, pinned)
        print("%-16d %-9s " % (n, "pinned" if pinned else "pageable") + " ".join("%9.1f (%6.1f)" % r[k] for k in ("mul_fixed_base", "mul_var_base", "poseidon5", "eddsa_verify")), flush=True)
