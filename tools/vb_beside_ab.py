#!/usr/bin/env python3
"""Developer probe: one device-pointer launch of 2^20 variable-base multiplications, clean and with 1 point in 4096 off the curve, in the
forced forms (BJJ_VB_SPLIT from the environment), interleaved and repeated -- what the exact kernel beside the batch kernel costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

n = 1 << 20
dev = torch.device("cuda", 0)
ctx = bjj.Context(0, 16)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
pts = ctx.mul_fixed_base(sc.reshape(n, 32)).copy()
d_sc = torch.from_numpy(sc).to(dev)
d_clean = torch.from_numpy(pts.reshape(-1)).to(dev)
d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
variants = {}
for every in (0, 65536, 4096, 97):
    p = pts.copy()
    if every:
        p[::every, 7] ^= 4
    variants[every] = torch.from_numpy(p.reshape(-1)).to(dev)
res = {k: [] for k in variants}
for rep in range(7):
    for every, d_p in variants.items():
        for k in range(2):      # the second of two identical launches is timed (history settled)
            torch.cuda.synchronize()
            t = time.perf_counter()
            ctx.mul_var_base_dev(d_p.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), 0)
            ctx.sync()
            dt = (time.perf_counter() - t) * 1e3
        if rep:
            res[every].append((dt, ctx.info().last_var_base_split))
for every, r in res.items():
    ts = sorted(x[0] for x in r)
    print("off-curve %-10s median %7.3f ms  min %7.3f  (exact kernel %s)" % ("none" if not every else "1 in %d" % every, ts[len(ts) // 2], ts[0], {0: "behind", 1: "beside"}[r[-1][1]]))
