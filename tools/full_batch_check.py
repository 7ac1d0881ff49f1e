#!/usr/bin/env python3
"""Developer tool: EVERY item of a 2^20 batch against the oracle (the test suite samples at this size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import conftest
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
n = int(os.environ.get("N", 1 << 20))
orc = conftest.Oracle(); orc.threads = int(os.environ.get("THREADS", "64"))
ctx = bjj.Context(0, 0)
sc = w.scalars_254(n)
t = time.time(); got = ctx.mul_fixed_base(sc); want = orc.mul_fixed_base(sc)
print("fixed_base: %d items, mismatches %d (%.1f s)" % (n, int((got != want).any(axis=1).sum()), time.time() - t))
pts = got.copy(); pts[::97, 5] ^= 4   # some off-curve points
t = time.time(); g2 = ctx.mul_var_base(pts, sc); w2 = orc.mul_var_base(pts, sc)
print("var_base:   %d items, mismatches %d (%.1f s)" % (n, int((g2 != w2).any(axis=1).sum()), time.time() - t))
A, R, S, msg = w.make_signatures(ctx.mul_fixed_base, ctx.poseidon5, n)
bad = w.corrupt(A, R, S, msg, n)
t = time.time(); g3 = ctx.eddsa_verify(A, R, S, msg); w3 = orc.verify(A, R, S, msg)
print("verify:     %d items, mismatches %d, rejected %d (%.1f s)" % (n, int((g3 != w3).sum()), int((g3 == 0).sum()), time.time() - t))
h = np.concatenate([R, A, msg], axis=1)
t = time.time(); g4 = ctx.poseidon5(h); w4 = orc.poseidon5(h)
print("poseidon5:  %d items, mismatches %d (%.1f s)" % (n, int((g4 != w4).any(axis=1).sum()), time.time() - t))
