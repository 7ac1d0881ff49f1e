#!/usr/bin/env python3
"""Developer probe: every host-pointer batch entry point whose chunk schedule is the pipeline's default, 2^20 items on pinned memory, best / median of a
few calls after 0.4 s of them.  Run once as is and once with BJJ_PIPE_FIRST_CHUNK=32768 BJJ_PIPE_CHUNK=262144 (the cap until round 6)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import babyjubjub_rs_amd as bjj

n = 1 << 20
ctx = bjj.Context(0, int(sys.argv[1]) if len(sys.argv) > 1 else 23)
rng = np.random.default_rng(7)
keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8); msgs[:, 31] &= 0x1f
pk = ctx.public_keys(keys)
proj = np.concatenate([pk, rng.integers(0, 256, (n, 32), dtype=np.uint8)], axis=1)
comp = ctx.compress_points(pk)
h5 = rng.integers(0, 256, (n, 160), dtype=np.uint8)
calls = [("bjj_mul_fixed_base", [(keys, 32)], [64]), ("bjj_public_keys", [(keys, 32)], [64]), ("bjj_scalar_keys", [(keys, 32)], [32]),
         ("bjj_poseidon5", [(h5, 160)], [32]), ("bjj_point_add", [(pk, 64), (pk[::-1].copy(), 64)], [64]), ("bjj_proj_add", [(proj, 96), (proj[::-1].copy(), 96)], [96]),
         ("bjj_proj_affine", [(proj, 96)], [64]), ("bjj_compress_points", [(pk, 64)], [32]), ("bjj_decompress_points", [(comp, 32)], [64, 1]),
         ("bjj_sign", [(keys, 32), (msgs, 32)], [64, 32, 1]), ("bjj_sign_compressed", [(keys, 32), (msgs, 32)], [64, 1]),
         ("bjj_public_keys_compressed", [(keys, 32)], [32]), ("bjj_mul_fixed_base_compressed", [(keys, 32)], [32])]
for name, ins, outs in calls:
    a_in = []
    for arr, wdt in ins:
        b = ctx.host_empty(n * wdt); b[:] = np.ascontiguousarray(arr).reshape(-1); a_in.append(b)
    a_out = [ctx.host_empty(n * wdt) for wdt in outs]
    args = [ctx.handle] + [b.ctypes.data for b in a_in] + [C.c_size_t(n)] + [b.ctypes.data for b in a_out]
    f = lambda: getattr(ctx.lib, name)(*args)
    t0 = time.perf_counter(); assert f() == 0, ctx.lib.bjj_last_error()
    while time.perf_counter() - t0 < 0.4: f()
    ts = []
    for _ in range(7):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    print("%-32s %7.3f ms (median %7.3f)  %2d chunks   in %3d B  out %3d B per item" % (name, min(ts) * 1e3, sorted(ts)[3] * 1e3, ctx.info().last_host_chunks,
                                                                                      sum(w for _, w in ins), sum(outs)), flush=True)
    for b in a_in + a_out:
        ctx.host_free(b)
