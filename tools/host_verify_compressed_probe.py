#!/usr/bin/env python3
"""Developer probe: bjj_eddsa_verify_compressed on pinned host memory, 2^20 wire-format signatures of the cfg-4 kind (1 in 64 corrupted), against one
device-pointer launch of the same inputs.  Run as is and with BJJ_PIPE_FIRST_CHUNK / BJJ_PIPE_CHUNK to compare chunk schedules."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
n = 1 << 20
ctx = bjj.Context(0, int(sys.argv[1]) if len(sys.argv) > 1 else 23)
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)
d_keys, d_m = up(w.random_u256(w.SEED_KEYS, n, 0)), up(w.random_u256(w.SEED_MSGS, n, 0, top_bits_cleared=3))
d_pk, d_r, d_s, d_f = (torch.empty(k, dtype=torch.uint8, device=dev) for k in (n * 64, n * 64, n * 32, n))
ctx.public_keys_dev(d_keys.data_ptr(), n, d_pk.data_ptr(), 0)
ctx.sign_dev(d_keys.data_ptr(), d_m.data_ptr(), n, d_r.data_ptr(), d_s.data_ptr(), d_f.data_ptr(), 0)
ctx.sync()
d_pkc, d_rc = torch.empty(n * 32, dtype=torch.uint8, device=dev), torch.empty(n * 32, dtype=torch.uint8, device=dev)
ctx.compress_points_dev(d_pk.data_ptr(), n, d_pkc.data_ptr(), 0); ctx.compress_points_dev(d_r.data_ptr(), n, d_rc.data_ptr(), 0); ctx.sync()
z = torch.zeros(n, 32, dtype=torch.uint8, device=dev)
A_t, R_t = torch.cat([d_pkc.view(n, 32), z], dim=1), torch.cat([z, d_rc.view(n, 32)], dim=1)
bad = w.corrupt(A_t, R_t, d_s.view(n, 32), d_m.view(n, 32), n, 0)
pk = A_t[:, :32].contiguous().reshape(-1)
sig = torch.cat([R_t[:, 32:], d_s.view(n, 32)], dim=1).contiguous().reshape(-1)
d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
def best(f, reps=5, warm_s=0.4):
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < warm_s: f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3
def dev_call():
    ctx.eddsa_verify_compressed_dev(pk.data_ptr(), sig.data_ptr(), d_m.data_ptr(), n, d_ok.data_ptr(), 0); ctx.sync()
t_dev = best(dev_call)
want = d_ok.cpu().numpy().copy()
h = [ctx.host_empty(t.numel()) for t in (pk, sig, d_m)]
for b, t in zip(h, (pk, sig, d_m)): b[:] = t.cpu().numpy()
ok = ctx.host_empty(n)
t_host = best(lambda: ctx._ck(ctx.lib.bjj_eddsa_verify_compressed(ctx.handle, h[0].ctypes.data, h[1].ctypes.data, h[2].ctypes.data, n, ok.ctypes.data), "vc"))
print("verify_compressed, 2^20 signatures (%d corrupted): one device launch %.3f ms   host call (pinned) %.3f ms (%d chunks)   equal: %s"
      % (int(bad.sum()), t_dev, t_host, ctx.info().last_host_chunks, bool((np.asarray(ok) == want).all())))
