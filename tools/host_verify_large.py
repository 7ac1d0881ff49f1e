#!/usr/bin/env python3
"""Developer probe: bjj_eddsa_verify on host pointers at a size that takes more than one super-batch of the default 1 GB device
staging (6 000 000 cfg-4 signatures = 1.16 GB of inputs): every verdict against device-pointer launches of the same inputs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000000
dev = torch.device("cuda", 0)
ctx = bjj.Context(0, 23)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1).view(np.uint8)).to(dev)
d_keys, d_msg = up(w.random_u256(w.SEED_KEYS, n, 0)), up(w.random_u256(w.SEED_MSGS, n, 0, top_bits_cleared=3))
d_pk, d_r, d_s, d_f = (torch.empty(k, dtype=torch.uint8, device=dev) for k in (n * 64, n * 64, n * 32, n))
ctx.public_keys_dev(d_keys.data_ptr(), n, d_pk.data_ptr(), 0)
ctx.sign_dev(d_keys.data_ptr(), d_msg.data_ptr(), n, d_r.data_ptr(), d_s.data_ptr(), d_f.data_ptr(), 0)
ctx.sync()
bad = w.corrupt(d_pk.view(n, 64), d_r.view(n, 64), d_s.view(n, 32), d_msg.view(n, 32), n, 0)
d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
t = time.perf_counter()
ctx.eddsa_verify_dev(d_pk.data_ptr(), d_r.data_ptr(), d_s.data_ptr(), d_msg.data_ptr(), n, d_ok.data_ptr(), 0)
ctx.sync()
t_dev = time.perf_counter() - t
want = d_ok.cpu().numpy()
assert (want == (~bad).astype(np.uint8)).all()
arrs = [x.cpu().numpy() for x in (d_pk, d_r, d_s, d_msg)]
for mem in ("pinned", "pageable"):
    alloc = ctx.host_empty if mem == "pinned" else (lambda nb: np.zeros(nb, np.uint8))
    h = [alloc(a.size) for a in arrs]
    for b, a in zip(h, arrs):
        b[:] = a
    ok = alloc(n)
    ok[:] = 0x77
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, h[0].ctypes.data, h[1].ctypes.data, h[2].ctypes.data, h[3].ctypes.data, n, ok.ctypes.data), "v")
        ts.append(time.perf_counter() - t)
    i = ctx.info()
    print("%-9s n = %d: best %.2f ms (%.1f M/s), chunks %d, direct %d staged %d; one device launch %.2f ms; every verdict equal: %s"
          % (mem, n, min(ts) * 1e3, n / min(ts) / 1e6, i.last_host_chunks, i.last_host_direct_arrays, i.last_host_staged_arrays, t_dev * 1e3,
             bool((np.asarray(ok) == want).all())), flush=True)
    if mem == "pinned":
        for b in h + [ok]:
            ctx.host_free(b)
