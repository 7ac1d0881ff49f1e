#!/usr/bin/env python3
"""Developer tool: a few n = 1 calls of the single-item entry points, for `rocprofv3 --kernel-trace --stats -- python3 tools/small_call_kernels.py`: the kernel times
inside a one-item call (what of a call's latency is kernel, and which kernel)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
ctx = bjj.Context(0, 23)
A, R, S, M = w.make_signatures(ctx.mul_fixed_base, ctx.poseidon5, 64)
keys = np.random.default_rng(1).integers(0, 256, (64, 32), dtype=np.uint8)
for _ in range(50):
    ctx.mul_fixed_base(S[:1]); ctx.mul_var_base(A[:1], S[:1]); ctx.poseidon5(np.concatenate([A[:1], R[:1], M[:1]], axis=1)); ctx.eddsa_verify(A[:1], R[:1], S[:1], M[:1])
    ctx.sign(keys[:1], M[:1]); ctx.public_keys(keys[:1])
