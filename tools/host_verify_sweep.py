#!/usr/bin/env python3
"""Developer probe: bjj_eddsa_verify on pinned host memory against the SAME inputs through device pointers (one stream, two
streams), over chunk schedules of the host pipeline (BJJ_PIPE_FIRST_CHUNK / BJJ_PIPE_CHUNK, read when a context is made).
VERDICT r04 item 2 asks for the host call within 5 % of the device-pointer rate.
usage: python3 tools/host_verify_sweep.py [first:max ...]     (items, log2; default a grid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

n = 1 << 20
dev = torch.device("cuda", 0)
grid = sys.argv[1:] or ["15:18", "15:19", "15:20", "16:19", "16:20", "17:19", "17:20", "14:18", "15:17", "18:18", "20:20"]


def best(f, reps=5, warm_s=0.5):
    t_w = time.perf_counter()
    f()
    while time.perf_counter() - t_w < warm_s:
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3


# valid signatures, 1/64 corrupted: the workload of bench.py --workload verify
ctx0 = bjj.Context(0, 23)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1).view(np.uint8)).to(dev)
d_keys, d_msg = up(w.random_u256(w.SEED_KEYS, n, 0)), up(w.random_u256(w.SEED_MSGS, n, 0, top_bits_cleared=3))
d_pk, d_r, d_s, d_f = (torch.empty(k, dtype=torch.uint8, device=dev) for k in (n * 64, n * 64, n * 32, n))
ctx0.public_keys_dev(d_keys.data_ptr(), n, d_pk.data_ptr(), 0)
ctx0.sign_dev(d_keys.data_ptr(), d_msg.data_ptr(), n, d_r.data_ptr(), d_s.data_ptr(), d_f.data_ptr(), 0)
ctx0.sync()
assert bool(d_f.all())
w.corrupt(d_pk.view(n, 64), d_r.view(n, 64), d_s.view(n, 32), d_msg.view(n, 32), n, 0)
d = [d_pk, d_r, d_s, d_msg]
arrs = [x.cpu().numpy() for x in d]
d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
torch.cuda.synchronize()


def dev_call(ctx, streams, k=4):
    def f():
        for j in range(k):
            ctx.eddsa_verify_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), n, d_ok.data_ptr(), streams[j % len(streams)].cuda_stream)
        ctx.sync()
    return f


t1 = best(dev_call(ctx0, (sa,)), 3)
t2 = best(dev_call(ctx0, (sa, sb)), 3)
print("device pointers, one stream : %.3f ms per 2^20 (median %.3f)" % (t1[0] / 4, t1[1] / 4))
print("device pointers, two streams: %.3f ms per 2^20 (median %.3f)" % (t2[0] / 4, t2[1] / 4), flush=True)
ref_ok = d_ok.cpu().numpy().copy()
ctx0.close()
for g in grid:
    a, b = g.split(":")
    os.environ["BJJ_PIPE_FIRST_CHUNK"], os.environ["BJJ_PIPE_CHUNK"] = str(1 << int(a)), str(1 << int(b))
    ctx = bjj.Context(0, 23)
    h = [ctx.host_empty(x.size) for x in arrs]
    for hh, x in zip(h, arrs):
        hh[:] = x
    ok = ctx.host_empty(n)
    f = lambda: ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, h[0].ctypes.data, h[1].ctypes.data, h[2].ctypes.data, h[3].ctypes.data, n, ok.ctypes.data), "v")
    t = best(f, 5)
    same = bool((np.asarray(ok) == ref_ok).all())
    print("host pinned  first 2^%s max 2^%s : best %.3f ms  median %.3f ms  chunks %d  (vs two streams %+.1f %%, vs one %+.1f %%)  verdicts equal: %s"
          % (a, b, t[0], t[1], ctx.info().last_host_chunks, (t[0] / (t2[0] / 4) - 1) * 100, (t[0] / (t1[0] / 4) - 1) * 100, same), flush=True)
    for hh in h + [ok]:
        ctx.host_free(hh)
    ctx.close()
