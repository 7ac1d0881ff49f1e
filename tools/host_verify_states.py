#!/usr/bin/env python3
"""Developer probe: bjj_eddsa_verify on pinned host memory in the process states a caller can be in (fresh context; after
device-pointer launches on the caller's streams; after another context has come and gone; a second context of the process).
Which hardware queue a HIP stream lands on depends on what the process created before (host-pointer boundary)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w

W = int(sys.argv[1]) if len(sys.argv) > 1 else 23
n = 1 << 20
dev = torch.device("cuda", 0)


def best(f, reps=5, warm_s=0.5):
    t_w = time.perf_counter()
    f()
    while time.perf_counter() - t_w < warm_s:
        f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, float(np.median(ts)) * 1e3


ctx = bjj.Context(0, W)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1).view(np.uint8)).to(dev)
d_keys, d_msg = up(w.random_u256(w.SEED_KEYS, n, 0)), up(w.random_u256(w.SEED_MSGS, n, 0, top_bits_cleared=3))
d_pk, d_r, d_s, d_f = (torch.empty(k, dtype=torch.uint8, device=dev) for k in (n * 64, n * 64, n * 32, n))
ctx.public_keys_dev(d_keys.data_ptr(), n, d_pk.data_ptr(), 0)
ctx.sign_dev(d_keys.data_ptr(), d_msg.data_ptr(), n, d_r.data_ptr(), d_s.data_ptr(), d_f.data_ptr(), 0)
ctx.sync()
w.corrupt(d_pk.view(n, 64), d_r.view(n, 64), d_s.view(n, 32), d_msg.view(n, 32), n, 0)
d = [d_pk, d_r, d_s, d_msg]
arrs = [x.cpu().numpy() for x in d]
d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()


def host(c, label, trace=False):
    h = [c.host_empty(x.size) for x in arrs]
    for hh, x in zip(h, arrs):
        hh[:] = x
    ok = c.host_empty(n)
    f = lambda: c._ck(c.lib.bjj_eddsa_verify(c.handle, h[0].ctypes.data, h[1].ctypes.data, h[2].ctypes.data, h[3].ctypes.data, n, ok.ctypes.data), "v")
    t = best(f)
    i = c.info()
    print("%-72s best %.3f ms  median %.3f ms  chunks %d  bad verdicts %d" % (label, t[0], t[1], i.last_host_chunks, int((np.asarray(ok) == 0).sum())), flush=True)
    for hh in h + [ok]:
        c.host_free(hh)


def devp(c, streams, label, k=4):
    def f():
        for j in range(k):
            c.eddsa_verify_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), n, d_ok.data_ptr(), streams[j % len(streams)].cuda_stream)
        c.sync()
    t = best(f, 3)
    print("%-72s best %.3f ms  median %.3f ms per 2^20" % (label, t[0] / k, t[1] / k), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "trace":   # BJJ_PIPE_TRACE=1 in the environment: two calls, their timelines on stderr
    h = [ctx.host_empty(x.size) for x in arrs]
    for hh, x in zip(h, arrs):
        hh[:] = x
    ok = ctx.host_empty(n)
    for _ in range(2):
        print("---- call", file=sys.stderr, flush=True)
        ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, h[0].ctypes.data, h[1].ctypes.data, h[2].ctypes.data, h[3].ctypes.data, n, ok.ctypes.data), "v")
    sys.exit(0)
host(ctx, "host: first context, after the signer kernels on its own stream")
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
devp(ctx, (sa,), "device pointers, one torch stream")
devp(ctx, (sa, sb), "device pointers, two torch streams")
host(ctx, "host: after device-pointer launches on two torch streams")
c2 = bjj.Context(0, 16)
c2.mul_fixed_base(np.zeros((3200, 32), np.uint8))
c2.close()
host(ctx, "host: after a second context has come and gone")
c3 = bjj.Context(0, 20)
host(c3, "host: a second context while the first lives")
ctx.close()
host(c3, "host: the second context after the first was closed")
c3.close()
c4 = bjj.Context(0, 20)
host(c4, "host: a context made after all others were closed")
devp(c4, (sa, sb), "device pointers, two torch streams, that context")
