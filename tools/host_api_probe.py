#!/usr/bin/env python3
"""Developer probe: bjj_mul_fixed_base / bjj_eddsa_verify on pinned host memory inside a torch process, in the states bench.py
goes through (fresh context; after launches on two torch streams; after the verify kernels have created their scan streams)."""
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
dev = torch.device("cuda", 0)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
h_sc, out = ctx.host_empty(n * 32), ctx.host_empty(n * 64)
h_sc[:] = sc
out[:] = 0


def fb(label, reps=7):
    f = lambda: ctx._ck(ctx.lib.bjj_mul_fixed_base(ctx.handle, h_sc.ctypes.data, n, out.ctypes.data), "fb")
    f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    i = ctx.info()
    print("%-60s best %.3f ms  median %.3f ms  (direct %d staged %d chunks %d)" % (label, min(ts) * 1e3, float(np.median(ts)) * 1e3,
          i.last_host_direct_arrays, i.last_host_staged_arrays, i.last_host_chunks), flush=True)


fb("fresh context")
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
d_sc = torch.from_numpy(sc).to(dev)
d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for k in range(8):
    ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr(), (sa, sb)[k & 1].cuda_stream)
ctx.sync()
fb("after fixed-base launches alternating over two torch streams")
d_pk = d_out.clone()
d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
for k in range(4):
    ctx.eddsa_verify_dev(d_pk.data_ptr(), d_pk.data_ptr(), d_sc.data_ptr(), d_sc.data_ptr(), n, d_ok.data_ptr(), (sa, sb)[k & 1].cuda_stream)
ctx.sync()
fb("after verify launches on two torch streams (scan streams exist)")
c2 = bjj.Context(0, 16)
c2.mul_fixed_base(sc[:3200])
fb("after a second context has come and gone" if c2.close() is None else "")
fb("again")
