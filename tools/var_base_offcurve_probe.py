#!/usr/bin/env python3
"""Developer probe: what off-curve input points cost the variable-base path (they take the reference's bit-serial formula
sequence in a launch of their own, k_var.hip: bjj_k_mul_var_base_exact -- behind the main kernel, or beside it behind a scan once the
context has met an off-curve point): one device-pointer launch of 2^20 items with none / 1 in 4096 / 1 in 97 / 1 in 8 of the points off
the curve, and the host-pointer call on pinned memory.  BJJ_VB_SPLIT=0 + BJJ_PIPE_VAR_BASE_SPLIT=0 in the environment = the round-5 forms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

n = 1 << 20
dev = torch.device("cuda", 0)
ctx = bjj.Context(0, int(sys.argv[1]) if len(sys.argv) > 1 else 23)
sc = np.ascontiguousarray(w.scalars_254(n)).reshape(-1)
d_sc = torch.from_numpy(sc).to(dev)
d_pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_pts.data_ptr(), 0)
ctx.sync()
base = d_pts.clone()


def best(f, reps=5, warm_s=0.4):
    t_w = time.perf_counter()
    f()
    while time.perf_counter() - t_w < warm_s:
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


def dev_call():
    ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), 0)
    ctx.sync()


h_pts, h_sc, h_out = ctx.host_empty(n * 64), ctx.host_empty(n * 32), ctx.host_empty(n * 64)
h_sc[:] = sc
for every in (0, 4096, 97, 8):
    d_pts.copy_(base)
    if every:
        v = d_pts.view(n, 64)
        v[::every, 7] ^= 4
    h_pts[:] = d_pts.cpu().numpy()
    t_dev = best(dev_call)
    s_dev = ctx.info().last_var_base_split
    t_host = best(lambda: ctx._ck(ctx.lib.bjj_mul_var_base(ctx.handle, h_pts.ctypes.data, h_sc.ctypes.data, n, h_out.ctypes.data), "vb"))
    s_host = ctx.info().last_var_base_split
    same = bool((torch.from_numpy(np.asarray(h_out)).to(dev) == d_out).all())
    form = {0: "behind", 1: "beside", -1: "?"}
    print("off-curve points: %-12s one device launch %7.3f ms (exact kernel %s)   host call (pinned) %7.3f ms (%s, %d chunks)   equal: %s"
          % ("none" if not every else "1 in %d" % every, t_dev, form[s_dev], t_host, form[s_host], ctx.info().last_host_chunks, same), flush=True)
