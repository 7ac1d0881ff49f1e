#!/usr/bin/env python3
"""Developer probe: a few device-pointer launches of 2^20 variable-base multiplications with 1 in `every` points off the curve (argv[1],
default 4096) -- run under rocprofv3 --kernel-trace and read with tools/kernel_timeline.py to see where the exact kernel sits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

n = 1 << 20
every = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda", 0)
ctx = bjj.Context(0, 16)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
pts = ctx.mul_fixed_base(sc.reshape(n, 32)).copy()
if every:
    pts[::every, 7] ^= 4
d_sc, d_pts = torch.from_numpy(sc).to(dev), torch.from_numpy(pts.reshape(-1)).to(dev)
d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
for k in range(5):
    t = time.perf_counter()
    ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), 0)
    ctx.sync()
    print("launch %d: %.3f ms, exact kernel %s" % (k, (time.perf_counter() - t) * 1e3, {0: "behind", 1: "beside"}[ctx.info().last_var_base_split]))
