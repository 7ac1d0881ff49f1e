#!/usr/bin/env python3
"""Developer A/B: the K1 host entry points (pinned memory) with the chunk launches on ONE workgroup slot per CU + equal 2^17-item chunks
(BJJ_PIPE_K1_HALF, default on) against the schedule of the rounds before (=0), and with / without the copy-in stage for short calls
(BJJ_PIPE_ZERO_COPY_IN).  A fresh process per row, the modes interleaved.  usage: fb_host_half_ab.py [W] [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, zlib
sys.path.insert(0, %r)
import ctypes as C
import numpy as np
import babyjubjub_rs_amd as bjj
N = 1 << max(int(x) for x in sys.argv[2].split(","))
ctx = bjj.Context(0, int(sys.argv[1]))
rng = np.random.default_rng(11)
h_in = ctx.host_empty(N * 32); h_in[:] = rng.integers(0, 256, N * 32, dtype=np.uint8)
o = ctx.host_empty(N * 64)
def best(f, reps=11, warm_s=0.3):
    t0 = time.perf_counter(); f()
    while time.perf_counter() - t0 < warm_s: f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3
for name, width in (("bjj_mul_fixed_base_compressed", 32), ("bjj_mul_fixed_base", 64), ("bjj_public_keys_compressed", 32), ("bjj_public_keys", 64)):
    row = []
    for lg in [int(x) for x in sys.argv[2].split(",")]:
        n = 1 << lg
        f = lambda: ctx._ck(getattr(ctx.lib, name)(ctx.handle, h_in.ctypes.data, C.c_size_t(n), o.ctypes.data), name)
        b = best(f)
        i = ctx.info()
        row.append("2^%%d %%.3f (%%d ch, zc %%d, crc %%08x)" %% (lg, b[0], i.last_host_chunks, i.last_host_zero_copy, zlib.crc32(o[:n * width].tobytes())))
    print("  %%-30s %%s" %% (name, "  ".join(row)), flush=True)
''' % ROOT
W = sys.argv[1] if len(sys.argv) > 1 else "23"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
SIZES = sys.argv[3] if len(sys.argv) > 3 else "10,15,16,17,18,19,20,22"
MODES = (("0", "0", None), ("1", "0", None), ("1", "1", None))
if len(sys.argv) > 4 and sys.argv[4] == "caps": MODES = (("1", "1", None), ("1", "1", "262144"), ("1", "1", "524288"))
FIRST = [None]
if len(sys.argv) > 4 and sys.argv[4] == "first": MODES, FIRST = (("1", "1", None),), ["0", "1", "2"]      # leading chunks of a long call that read in place
for rnd in range(rounds):
    for half, zi, cap, first in [m + (f,) for m in MODES for f in FIRST]:
        print("# round %d: BJJ_PIPE_K1_HALF=%s BJJ_PIPE_ZERO_COPY_IN=%s%s%s   (ms, best of 11 calls)" % (rnd, half, zi, "  BJJ_PIPE_FIRST_CHUNK=65536 BJJ_PIPE_CHUNK=" + cap if cap else "", "  BJJ_PIPE_ZERO_COPY_IN_FIRST=" + first if first else ""), flush=True)
        env = dict(os.environ, BJJ_PIPE_K1_HALF=half, BJJ_PIPE_ZERO_COPY_IN=zi)
        if first: env["BJJ_PIPE_ZERO_COPY_IN_FIRST"] = first
        if cap: env.update(BJJ_PIPE_FIRST_CHUNK="65536", BJJ_PIPE_CHUNK=cap)
        r = subprocess.run([sys.executable, "-c", CHILD, W, SIZES], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        print(r.stdout.rstrip() or r.stderr[-600:], flush=True)
