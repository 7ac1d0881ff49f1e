#!/usr/bin/env python3
"""Developer tool: quick parity + timing probe on a GPU box (not a test, not the bench)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import babyjubjub_rs_amd as bjj
import bjj_oracle as o

R = ctypes.CDLL(os.path.join(ROOT, "oracle", "libbjj_oracle.so"))
W = int(os.environ.get("BJJ_W", "0"))
t0 = time.time(); ctx = bjj.Context(0, W); print("init %.2fs" % (time.time() - t0), "W", ctx.info().window_bits, "table MB", ctx.info().table_bytes / 1e6)
rng = np.random.default_rng(1)
NT = os.cpu_count()

def ref_fixed(sc):
    n = sc.size // 32; out = np.empty(n * 64, np.uint8)
    R.bjjref_mul_fixed_base_batch(sc.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), out.ctypes.data_as(ctypes.c_void_p), NT); return out.reshape(n, 64)
def ref_var(p, sc):
    n = sc.size // 32; out = np.empty(n * 64, np.uint8)
    R.bjjref_mul_var_base_batch(p.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), out.ctypes.data_as(ctypes.c_void_p), NT); return out.reshape(n, 64)
def ref_pos(a):
    n = a.size // 160; out = np.empty(n * 32, np.uint8)
    R.bjjref_poseidon5_batch(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), out.ctypes.data_as(ctypes.c_void_p), NT); return out.reshape(n, 32)
def ref_ver(pk, r, s, m):
    n = s.size // 32; out = np.empty(n, np.uint8)
    R.bjjref_verify_batch(pk.ctypes.data_as(ctypes.c_void_p), r.ctypes.data_as(ctypes.c_void_p), s.ctypes.data_as(ctypes.c_void_p), m.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), out.ctypes.data_as(ctypes.c_void_p), NT); return out

n = 3000
sc = rng.integers(0, 256, n * 32, dtype=np.uint8)
t0 = time.time(); g = ctx.mul_fixed_base(sc); t1 = time.time(); e = ref_fixed(sc); t2 = time.time()
print("fixed-base parity:", bool((g == e).all()), "mismatch rows", int((g != e).any(axis=1).sum()), "gpu %.3fs cpu %.3fs" % (t1 - t0, t2 - t1))
pts = ref_fixed(rng.integers(0, 256, n * 32, dtype=np.uint8))  # on-curve points
pts[5] = 0; pts[6, :] = 1  # off-curve rows
g = ctx.mul_var_base(pts.reshape(-1), sc); e = ref_var(pts.reshape(-1), sc)
print("var-base parity:", bool((g == e).all()), "mismatch rows", int((g != e).any(axis=1).sum()))
a = rng.integers(0, 256, n * 160, dtype=np.uint8); a.reshape(-1, 32)[:, 31] &= 0x1f
g = ctx.poseidon5(a); e = ref_pos(a)
print("poseidon parity:", bool((g == e).all()), "mismatch rows", int((g != e).any(axis=1).sum()))
# signatures: k, rho random; A = kB8, R = rho B8, S = rho + 8 hm k mod l
nv = 1500
ks = [int.from_bytes(rng.bytes(32), "little") % o.SUBORDER for _ in range(nv)]
rhos = [int.from_bytes(rng.bytes(32), "little") % o.SUBORDER for _ in range(nv)]
msgs = [int.from_bytes(rng.bytes(32), "little") % o.Q for _ in range(nv)]
A = ref_fixed(np.frombuffer(b"".join(o.to_le32(k) for k in ks), np.uint8))
Rp = ref_fixed(np.frombuffer(b"".join(o.to_le32(k) for k in rhos), np.uint8))
mb = np.frombuffer(b"".join(o.to_le32(m) for m in msgs), np.uint8).copy()
hin = np.concatenate([Rp, A, mb.reshape(nv, 32)], axis=1).reshape(-1)
hm = ref_pos(np.ascontiguousarray(hin))
S = [(rhos[i] + 8 * int.from_bytes(hm[i].tobytes(), "little") * ks[i]) % o.SUBORDER for i in range(nv)]
sb = np.frombuffer(b"".join(o.to_le32(s) for s in S), np.uint8).copy()
sb.reshape(nv, 32)[::7, 0] ^= 1  # corrupt some
A2 = A.copy(); A2[3] = 7  # off-curve pk
g = ctx.eddsa_verify(A2.reshape(-1), Rp.reshape(-1), sb, mb); e = ref_ver(A2.reshape(-1), Rp.reshape(-1), sb, mb)
print("verify parity:", bool((g == e).all()), "mismatch", int((g != e).sum()), "valid", int(e.sum()), "of", nv)

# ---- timing with device-resident buffers
import torch
dev = torch.device("cuda:0")
N = int(os.environ.get("BJJ_N", str(1 << 20)))
ctx.reserve(N)
tstream = torch.cuda.Stream()
stream = tstream.cuda_stream
assert stream != 0
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(tstream); fn(); e1.record(tstream); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), sorted(ts)[len(ts) // 2]
d_sc = torch.from_numpy(rng.integers(0, 256, N * 32, dtype=np.uint8)).to(dev)
d_out = torch.empty(N * 64, dtype=torch.uint8, device=dev)
mn, md = timeit(lambda: ctx.mul_fixed_base_dev(d_sc.data_ptr(), N, d_out.data_ptr(), stream))
print("fixed-base N=%d: min %.3f ms median %.3f ms -> %.1f M/s" % (N, mn, md, N / md / 1e3))
chk = d_out[:64 * 2000].cpu().numpy().reshape(-1, 64); e = ref_fixed(d_sc[:32 * 2000].cpu().numpy())
print("  spot parity:", bool((chk == e).all()))
Nv = N // 8
d_pts = d_out[:Nv * 64].clone()
d_out2 = torch.empty(Nv * 64, dtype=torch.uint8, device=dev)
mn, md = timeit(lambda: ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), Nv, d_out2.data_ptr(), stream), 3)
print("var-base N=%d: min %.3f ms median %.3f ms -> %.2f M/s" % (Nv, mn, md, Nv / md / 1e3))
d_in = torch.from_numpy(rng.integers(0, 256, Nv * 160, dtype=np.uint8)).to(dev)
d_h = torch.empty(Nv * 32, dtype=torch.uint8, device=dev)
mn, md = timeit(lambda: ctx.poseidon5_dev(d_in.data_ptr(), Nv, d_h.data_ptr(), stream), 3)
print("poseidon5 N=%d: min %.3f ms median %.3f ms -> %.2f M/s" % (Nv, mn, md, Nv / md / 1e3))
d_ok = torch.empty(Nv, dtype=torch.uint8, device=dev)
d_m = d_sc[:Nv * 32].clone(); d_m.view(-1, 32)[:, 31] &= 0x0f
mn, md = timeit(lambda: ctx.eddsa_verify_dev(d_pts.data_ptr(), d_pts.data_ptr(), d_sc.data_ptr(), d_m.data_ptr(), Nv, d_ok.data_ptr(), stream), 3)
print("verify N=%d: min %.3f ms median %.3f ms -> %.2f M/s" % (Nv, mn, md, Nv / md / 1e3))
