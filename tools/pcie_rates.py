"""Developer tool: PCIe-inclusive rates of the host-pointer API (never bench.py's `value`)."""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
ctx = bjj.Context(0)
n = 1 << 20
sc = w.scalars_254(n)
def best(f, reps=4):
    f(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts)
t = best(lambda: ctx.mul_fixed_base(sc))
print("fixed-base host API: %.2f ms per 2^20 -> %.1f M/s PCIe-inclusive (32 MB in, 64 MB out)" % (t * 1e3, n / t / 1e6))
A = ctx.mul_fixed_base(sc); m = w.random_u256(w.SEED_MSGS, n, 0, 3)
t = best(lambda: ctx.eddsa_verify(A, A, sc, m), 3)
print("verify host API: %.2f ms per 2^20 -> %.1f M/s PCIe-inclusive (192 MB in, 1 MB out)" % (t * 1e3, n / t / 1e6))
pk = ctx.compress_points(A); sig = np.concatenate([pk, sc], axis=1)
t = best(lambda: ctx.eddsa_verify_compressed(pk, sig, m), 3)
print("verify (compressed) host API: %.2f ms per 2^20 -> %.1f M/s PCIe-inclusive (128 MB in, 1 MB out)" % (t * 1e3, n / t / 1e6))
# the C entry point alone, with caller buffers that are already paged in (no allocation / first-touch faults in the timed call)
import ctypes
out = np.empty(n * 64, np.uint8); out[:] = 0
scc = np.ascontiguousarray(sc).reshape(-1)
t = best(lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, scc.ctypes.data, n, out.ctypes.data), "x"), 6)
print("fixed-base C entry point, warm caller buffers: %.2f ms per 2^20 -> %.1f M/s" % (t * 1e3, n / t / 1e6))
ok = np.zeros(n, np.uint8); Af = np.ascontiguousarray(A).reshape(-1); mf = np.ascontiguousarray(m).reshape(-1)
t = best(lambda: ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, Af.ctypes.data, Af.ctypes.data, scc.ctypes.data, mf.ctypes.data, n, ok.ctypes.data), "x"), 3)
print("verify C entry point, warm caller buffers: %.2f ms per 2^20 -> %.1f M/s" % (t * 1e3, n / t / 1e6))
