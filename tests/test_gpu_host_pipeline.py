"""The host-pointer boundary (what a caller of Point::mul_scalar / verify holding its data on the host binds to,
/root/reference src/lib.rs:149, :395): chunked pipeline, pinned arrays copied directly, pageable arrays staged by the
context's copy workers -- every byte compared with the oracle on both paths, mixed pinned / pageable arrays, ragged sizes
around the chunk schedule, and the kernel-form heuristic for callers that alternate over streams with and without a
synchronisation between their launches.  Needs a real MI355X: run with `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx16():
    import babyjubjub_rs_amd as bjj
    c = bjj.Context(0, 16)
    yield c
    c.close()


def _pinned_copy(ctx, a):
    p = ctx.host_empty(a.size)
    p[:] = np.ascontiguousarray(a).reshape(-1)
    return p


# 1 item .. beyond the first three chunks of the schedule (2^16, 2^17, 2^18, 2^18 ...), ragged by one item on both sides
SIZES = [1, 63, 4096 + 5, (1 << 16) - 1, (1 << 16) + 1, (1 << 16) + (1 << 17) + 1, (1 << 19) + 333]


@pytest.mark.parametrize("n", SIZES)
def test_fixed_base_pinned_and_pageable_agree_with_the_oracle(ctx16, oracle, n):
    from babyjubjub_rs_amd import workload as w
    sc = np.ascontiguousarray(w.scalars_254(n, offset=n)).reshape(-1)
    # (a) pageable in, pageable out: staged by the copy workers
    out_a = np.zeros(n * 64, np.uint8)
    ctx16._ck(ctx16.lib.bjj_mul_fixed_base(ctx16.handle, sc.ctypes.data, n, out_a.ctypes.data), "bjj_mul_fixed_base")
    ia = ctx16.info()
    assert (ia.last_host_direct_arrays, ia.last_host_staged_arrays) == (0, 2) and ia.host_copy_threads >= 1
    # (b) pinned in, pinned out: no staging at all
    psc, pout = _pinned_copy(ctx16, sc), ctx16.host_empty(n * 64)
    pout[:] = 0
    assert ctx16.host_is_pinned(psc) and ctx16.host_is_pinned(pout) and not ctx16.host_is_pinned(out_a)
    ctx16._ck(ctx16.lib.bjj_mul_fixed_base(ctx16.handle, psc.ctypes.data, n, pout.ctypes.data), "bjj_mul_fixed_base")
    ib = ctx16.info()
    assert (ib.last_host_direct_arrays, ib.last_host_staged_arrays) == (2, 0)
    assert ib.last_host_chunks == ia.last_host_chunks >= 1
    # (c) mixed: pinned in, pageable out
    out_c = np.zeros(n * 64, np.uint8)
    ctx16._ck(ctx16.lib.bjj_mul_fixed_base(ctx16.handle, psc.ctypes.data, n, out_c.ctypes.data), "bjj_mul_fixed_base")
    ic = ctx16.info()
    assert (ic.last_host_direct_arrays, ic.last_host_staged_arrays) == (1, 1)
    assert (out_a == pout).all() and (out_a == out_c).all()
    idx = np.unique(np.concatenate([np.arange(0, n, max(1, n // 300)), np.arange(max(0, n - 70), n)]))   # every chunk boundary region
    for edge in (1 << 16, (1 << 16) + (1 << 17), (1 << 16) + (1 << 17) + (1 << 18)):
        if edge < n:
            idx = np.unique(np.concatenate([idx, np.arange(edge - 2, min(n, edge + 2))]))
    assert (out_a.reshape(n, 64)[idx] == oracle.mul_fixed_base(sc.reshape(n, 32)[idx])).all()
    ctx16.host_free(psc)
    ctx16.host_free(pout)


def test_verify_pinned_registered_and_pageable_every_verdict(ctx16, oracle):
    """verify through the pipeline: chunk kernels alternate over the context's two compute streams (two scratch sets).
    Arrays: pk pinned by the library, R registered in place (bjj_host_register), s and msg pageable -> 2 direct + 2 staged
    inputs; the verdicts land in pinned memory.  Every verdict against the corruption mask, a sample against the oracle."""
    from babyjubjub_rs_amd import workload as w
    n = (1 << 18) + (1 << 16) + 123
    A, R, S, msg = w.make_signatures(ctx16.mul_fixed_base, ctx16.poseidon5, n)
    bad = w.corrupt(A, R, S, msg, n)
    pA = _pinned_copy(ctx16, A)
    Rr = np.ascontiguousarray(R).reshape(-1).copy()
    ctx16.host_register(Rr)
    assert ctx16.host_is_pinned(Rr)
    Sp, Mp = np.ascontiguousarray(S).reshape(-1), np.ascontiguousarray(msg).reshape(-1)
    ok = ctx16.host_empty(n)
    ok[:] = 7
    ctx16._ck(ctx16.lib.bjj_eddsa_verify(ctx16.handle, pA.ctypes.data, Rr.ctypes.data, Sp.ctypes.data, Mp.ctypes.data, n, ok.ctypes.data),
              "bjj_eddsa_verify")
    i = ctx16.info()
    assert (i.last_host_direct_arrays, i.last_host_staged_arrays) == (3, 2)
    assert (np.asarray(ok) == (~bad).astype(np.uint8)).all()
    idx = np.arange(0, n, 997)
    assert (np.asarray(ok)[idx] == oracle.verify(A[idx], R[idx], S[idx], msg[idx])).all()
    # the same call with everything pageable gives the same bytes
    ok2 = np.full(n, 7, np.uint8)
    ctx16._ck(ctx16.lib.bjj_eddsa_verify(ctx16.handle, np.ascontiguousarray(A).ctypes.data, np.ascontiguousarray(R).ctypes.data, Sp.ctypes.data,
                                         Mp.ctypes.data, n, ok2.ctypes.data), "bjj_eddsa_verify")
    assert (ok2 == np.asarray(ok)).all()
    ctx16.host_unregister(Rr)
    assert not ctx16.host_is_pinned(Rr)
    ctx16.host_free(pA)
    ctx16.host_free(ok)


def test_memory_pinned_by_somebody_else_is_recognised(ctx16, oracle):
    """torch's pin_memory (hipHostMalloc behind the library's back): found through the driver's pointer attributes"""
    import torch
    from babyjubjub_rs_amd import workload as w
    n = 70001
    sc = w.scalars_254(n, offset=3)
    t_in = torch.from_numpy(np.ascontiguousarray(sc).reshape(-1).copy()).pin_memory()
    t_out = torch.zeros(n * 64, dtype=torch.uint8).pin_memory()
    ctx16._ck(ctx16.lib.bjj_mul_fixed_base(ctx16.handle, t_in.data_ptr(), n, t_out.data_ptr()), "bjj_mul_fixed_base")
    i = ctx16.info()
    assert (i.last_host_direct_arrays, i.last_host_staged_arrays) == (2, 0)
    idx = np.arange(0, n, 211)
    assert (t_out.numpy().reshape(n, 64)[idx] == oracle.mul_fixed_base(sc[idx])).all()


def test_host_memory_api_argument_checks(ctx16):
    from babyjubjub_rs_amd import _lib
    lib, h = ctx16.lib, ctx16.handle
    p = ctypes.c_void_p()
    assert lib.bjj_host_alloc(None, 64, ctypes.byref(p)) == _lib.BJJ_E_INVALID
    assert lib.bjj_host_alloc(h, 0, ctypes.byref(p)) == _lib.BJJ_E_INVALID
    assert lib.bjj_host_alloc(h, 64, None) == _lib.BJJ_E_INVALID
    assert lib.bjj_host_free(h, None) == _lib.BJJ_OK
    a = np.zeros(4096, np.uint8)
    assert lib.bjj_host_free(h, a.ctypes.data) == _lib.BJJ_E_INVALID          # not ours
    assert lib.bjj_host_unregister(h, a.ctypes.data) == _lib.BJJ_E_INVALID    # never registered
    assert lib.bjj_host_alloc(h, 1 << 20, ctypes.byref(p)) == _lib.BJJ_OK and p.value
    assert lib.bjj_host_is_pinned(h, p.value, 1 << 20) == 1
    assert lib.bjj_host_is_pinned(h, p.value + 4096, (1 << 20) - 4096) == 1   # a sub-range
    assert lib.bjj_host_unregister(h, p.value) == _lib.BJJ_E_INVALID          # allocated, not registered
    assert lib.bjj_host_free(h, p.value) == _lib.BJJ_OK
    assert lib.bjj_host_free(h, p.value) == _lib.BJJ_E_INVALID                # twice


def test_forced_staging_and_chunk_schedule_from_the_environment(oracle):
    """BJJ_HOST_FORCE_STAGED / BJJ_PIPE_CHUNK / BJJ_PIPE_FIRST_CHUNK / BJJ_STAGE_THREADS, in a child process (the knobs are
    read when a context first runs a host-pointer call): pinned arrays are then staged too, 10 000 items in 1 024 / 2 048 /
    2 048 ... chunks recycle the 4-deep pinned rings, and the bytes do not change."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
c = bjj.Context(0, 16)
n = 10000
sc = np.ascontiguousarray(w.scalars_254(n, offset=9)).reshape(-1)
p_in, p_out = c.host_empty(n * 32), c.host_empty(n * 64)
p_in[:] = sc
c._ck(c.lib.bjj_mul_fixed_base(c.handle, p_in.ctypes.data, n, p_out.ctypes.data), "fb")
i = c.info()
print("INFO", i.last_host_direct_arrays, i.last_host_staged_arrays, i.last_host_chunks, i.host_copy_threads)
np.save(sys.argv[1], np.asarray(p_out).copy())
c.close()
''' % ROOT
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "out.npy")
        env = dict(os.environ, BJJ_HOST_FORCE_STAGED="1", BJJ_PIPE_CHUNK="2048", BJJ_PIPE_FIRST_CHUNK="1024", BJJ_STAGE_THREADS="2")
        r = subprocess.run([sys.executable, "-c", code, outp], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout
        info = [l for l in r.stdout.splitlines() if l.startswith("INFO")][0].split()[1:]
        assert [int(x) for x in info] == [0, 2, 5, 2], r.stdout      # 1024 + 2048 x 3 + 2832 (the 784-item remainder joins the last chunk): five chunks, two workers
        got = np.load(outp).reshape(10000, 64)
    from babyjubjub_rs_amd import workload as w
    assert (got == oracle.mul_fixed_base(w.scalars_254(10000, offset=9))).all()


def test_every_host_entry_point_on_pinned_memory_equals_the_staged_path(ctx16, oracle):
    """The 16 host-pointer batch entry points with EVERY array in pinned memory (several inputs, up to three outputs, key material that
    is wiped behind the call) against the same call on pageable memory, 140 001 items = three chunks over both lanes; the
    staged results of the multiplications, the hash, the signer and the verifier are checked against the oracle on a sample."""
    import ctypes as C
    from babyjubjub_rs_amd import workload as w
    n = 140001
    assert _schedule(n, 1 << 15, 1 << 18) == [32768, 65536, 41697]
    rng = np.random.default_rng(0x70696e)
    keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs[:, 31] &= 0x1f
    msgs[::97, 31] = 0xff                                        # a few msg > Q: Err rows
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    nonces = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    wide = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    h5 = rng.integers(0, 256, (n, 160), dtype=np.uint8)
    pk = ctx16.public_keys(keys)
    pts = pk.copy()
    pts[::53, 7] ^= 4                                            # some off-curve points (exact path)
    proj = np.concatenate([pts, rng.integers(0, 256, (n, 32), dtype=np.uint8)], axis=1)
    proj[::101, 64:] = 0                                         # z == 0
    r, s, okf = ctx16.sign(keys, msgs)
    pk_v, r_v = pk.copy(), r.copy()
    pk_v[::53, 7] ^= 4                                           # off-curve pk / R: the exact items of the verify pipeline,
    r_v[5::211, 40] ^= 1                                         # which run as ONE launch beside the chunks' bulk launches
    comp = ctx16.compress_points(pk)
    comp[::71, 3] ^= 1                                           # some that do not decompress
    sig = np.concatenate([ctx16.compress_points(r), s], axis=1)
    lib, hnd = ctx16.lib, ctx16.handle
    calls = [   # (function, inputs [(array, row bytes)], scalar args before n, output row bytes)
        ("bjj_mul_fixed_base", [(sc, 32)], [], [64]),
        ("bjj_mul_var_base", [(pts, 64), (sc, 32)], [], [64]),
        ("bjj_mul_var_base_wide", [(pts, 64), (wide, 64)], [64], [64]),
        ("bjj_poseidon5", [(h5, 160)], [], [32]),
        ("bjj_eddsa_verify", [(pk_v, 64), (r_v, 64), (s, 32), (msgs, 32)], [], [1]),
        ("bjj_schnorr_verify", [(pk_v, 64), (r_v, 64), (s, 32), (msgs, 32)], [], [1]),
        ("bjj_point_add", [(pts, 64), (pk, 64)], [], [64]),
        ("bjj_proj_add", [(proj, 96), (proj[::-1].copy(), 96)], [], [96]),
        ("bjj_proj_affine", [(proj, 96)], [], [64]),
        ("bjj_compress_points", [(pk, 64)], [], [32]),
        ("bjj_decompress_points", [(comp, 32)], [], [64, 1]),
        ("bjj_eddsa_verify_compressed", [(comp, 32), (sig, 64), (msgs, 32)], [], [1]),
        ("bjj_scalar_keys", [(keys, 32)], [], [32]),
        ("bjj_public_keys", [(keys, 32)], [], [64]),
        ("bjj_sign", [(keys, 32), (msgs, 32)], [], [64, 32, 1]),
        ("bjj_sign_schnorr", [(keys, 32), (msgs, 32), (nonces, 128)], [], [64, 160, 1]),
    ]
    results = {}
    for name, ins, extra, outs in calls:
        got = {}
        for mem in ("pageable", "pinned"):
            alloc = ctx16.host_empty if mem == "pinned" else (lambda nb: np.zeros(nb, np.uint8))
            a_in = []
            for arr, wdt in ins:
                b = alloc(n * wdt)
                b[:] = np.ascontiguousarray(arr).reshape(-1)
                a_in.append(b)
            a_out = [alloc(n * wdt) for wdt in outs]
            for b in a_out:
                b[:] = 0xAB
            args = [hnd] + [b.ctypes.data for b in a_in]
            if name == "bjj_mul_var_base_wide":
                args += [C.c_size_t(64)]
            args += [C.c_size_t(n)] + [b.ctypes.data for b in a_out]
            rc = getattr(lib, name)(*args)
            assert rc == 0, (name, mem, lib.bjj_last_error())
            i = ctx16.info()
            want = (len(ins) + len(outs), 0) if mem == "pinned" else (0, len(ins) + len(outs))
            # the verifiers and the variable-base multiplications bring their own schedule (first chunk 2^16: their kernels hide the copies, bjj_hip.hip)
            chunks = {"bjj_eddsa_verify": len(_schedule(n, 1 << 16, 1 << 19)), "bjj_schnorr_verify": len(_schedule(n, 1 << 16, 1 << 19)),
                      "bjj_eddsa_verify_compressed": len(_schedule(n, 1 << 16, 1 << 19)),      # round 6: the verifiers' schedule and split
                      "bjj_mul_var_base": len(_schedule(n, 1 << 16, 1 << 18)), "bjj_mul_var_base_wide": len(_schedule(n, 1 << 16, 1 << 18))}.get(name, 3)
            assert (i.last_host_direct_arrays, i.last_host_staged_arrays) == want and i.last_host_chunks == chunks, (name, mem, i.last_host_chunks)
            got[mem] = [np.asarray(b).copy() for b in a_out]
            if mem == "pinned":
                for b in a_in + a_out:
                    ctx16.host_free(b)
        for x, y in zip(got["pageable"], got["pinned"]):
            assert (x == y).all(), name
        results[name] = got["pageable"]
    idx = np.unique(np.concatenate([np.arange(0, n, 499), [32767, 32768, 98303, 98304, n - 1]]))   # incl. both chunk seams
    assert (results["bjj_mul_fixed_base"][0].reshape(n, 64)[idx] == oracle.mul_fixed_base(sc[idx])).all()
    assert (results["bjj_mul_var_base"][0].reshape(n, 64)[idx] == oracle.mul_var_base(pts[idx], sc[idx])).all()
    assert (results["bjj_poseidon5"][0].reshape(n, 32)[idx] == oracle.poseidon5(h5[idx])).all()
    vidx = np.unique(np.concatenate([idx, np.arange(0, n, 53)[::7], np.arange(5, n, 211)[::5]]))   # incl. off-curve items
    assert (results["bjj_eddsa_verify"][0][vidx] == oracle.verify(pk_v[vidx], r_v[vidx], s[vidx], msgs[vidx])).all()
    # every verdict of both verifiers against the one-launch device-pointer call (exact groups inside the launch)
    assert (results["bjj_eddsa_verify"][0] == _verify_one_launch(ctx16, False, pk_v, r_v, s, msgs)).all()
    assert (results["bjj_schnorr_verify"][0] == _verify_one_launch(ctx16, True, pk_v, r_v, s, msgs)).all()
    assert (results["bjj_public_keys"][0].reshape(n, 64)[idx] == oracle.public_keys(keys[idx])).all()
    ro, so, oko = oracle.sign(keys[idx], msgs[idx])
    rs = results["bjj_sign"]
    assert (rs[0].reshape(n, 64)[idx] == ro).all() and (rs[1].reshape(n, 32)[idx] == so).all() and (rs[2][idx] == oko).all()
    dp, dok = oracle.decompress(comp[idx])
    assert (results["bjj_decompress_points"][0].reshape(n, 64)[idx] == dp).all() and (results["bjj_decompress_points"][1][idx] == dok).all()


def _verify_one_launch(ctx, schnorr, pk, r, s, msg):
    """the device-pointer entry point on the whole batch: one scan + one launch that holds the exact groups and the bulk"""
    import torch
    dev = torch.device("cuda", 0)
    n = pk.shape[0]
    d = [torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev) for a in (pk, r, s, msg)]
    d_ok = torch.full((n,), 0xAB, dtype=torch.uint8, device=dev)
    f = ctx.schnorr_verify_dev if schnorr else ctx.eddsa_verify_dev
    f(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), n, d_ok.data_ptr(), 0)
    ctx.sync()
    return d_ok.cpu().numpy()


def _schedule(n, first, cap_chunk):
    """the chunk schedule of run_super_batch (bjj_hip.hip): first, doubling up to the cap, a remainder below half a chunk joins the last one"""
    out, lo, sz = [], 0, first
    while lo < n:
        take = min(sz, n - lo)
        if n - lo - take < sz // 2:
            take = n - lo
        out.append(take)
        lo += take
        if sz < cap_chunk:
            sz = min(sz * 2, cap_chunk)
    return out


def test_super_batches_when_the_device_staging_budget_is_small(oracle):
    """BJJ_PIPE_STAGING_MB=1: 50 000 fixed-base items (96 B each) no longer fit into the device staging at once and run as five
    consecutive super-batches of 10 240 items; pinned and pageable callers, every byte against the oracle."""
    import os
    import subprocess
    import sys
    import tempfile
    from conftest import ROOT
    from babyjubjub_rs_amd import workload as w
    n = 50000
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
c = bjj.Context(0, 16)
n = %d
sc = np.ascontiguousarray(w.scalars_254(n, offset=11)).reshape(-1)
p_in, p_out = c.host_empty(n * 32), c.host_empty(n * 64)
p_in[:] = sc
c._ck(c.lib.bjj_mul_fixed_base(c.handle, p_in.ctypes.data, n, p_out.ctypes.data), "fb")
i = c.info()
print("INFO", i.last_host_direct_arrays, i.last_host_staged_arrays, i.last_host_chunks)
out2 = np.zeros(n * 64, np.uint8)
c._ck(c.lib.bjj_mul_fixed_base(c.handle, sc.ctypes.data, n, out2.ctypes.data), "fb")
i = c.info()
print("INFO", i.last_host_direct_arrays, i.last_host_staged_arrays, i.last_host_chunks)
assert (np.asarray(p_out) == out2).all()
np.save(sys.argv[1], out2)
# verify across super-batches (193 B per item: 4 096 items each): the list of off-curve items is reset per super-batch and its
# batch-wide indices start at the super-batch; cfg-4 signatures, 1 in 64 corrupted, half of those off the curve
nv = 15000
A, R, S, msg = w.make_signatures(c.mul_fixed_base, c.poseidon5, nv)
bad = w.corrupt(A, R, S, msg, nv)
flat = [np.ascontiguousarray(a).reshape(-1) for a in (A, R, S, msg)]
ok_pg = np.full(nv, 9, np.uint8)
c._ck(c.lib.bjj_eddsa_verify(c.handle, flat[0].ctypes.data, flat[1].ctypes.data, flat[2].ctypes.data, flat[3].ctypes.data, nv, ok_pg.ctypes.data), "v")
print("VINFO", c.info().last_host_chunks, c.info().last_verify_dispatch)
pin = []
for a in flat:
    b = c.host_empty(a.size); b[:] = a; pin.append(b)
ok_pin = c.host_empty(nv); ok_pin[:] = 9
c._ck(c.lib.bjj_schnorr_verify(c.handle, pin[0].ctypes.data, pin[1].ctypes.data, pin[2].ctypes.data, pin[3].ctypes.data, nv, ok_pin.ctypes.data), "vs")
c._ck(c.lib.bjj_eddsa_verify(c.handle, pin[0].ctypes.data, pin[1].ctypes.data, pin[2].ctypes.data, pin[3].ctypes.data, nv, ok_pin.ctypes.data), "v")
assert (ok_pg == (~bad).astype(np.uint8)).all() and (np.asarray(ok_pin) == ok_pg).all(), (int((ok_pg != (~bad)).sum()), int((np.asarray(ok_pin) != ok_pg).sum()))
print("VOK", int(bad.sum()))
c.close()
''' % (ROOT, n)
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "out.npy")
        env = dict(os.environ, BJJ_PIPE_STAGING_MB="1", BJJ_PIPE_CHUNK="2048", BJJ_PIPE_FIRST_CHUNK="1024")
        r = subprocess.run([sys.executable, "-c", code, outp], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout
        cap = (1 << 20) // 96 // 2048 * 2048                      # items per super-batch: 10 240
        want = sum(len(_schedule(min(cap, n - lo), 1024, 2048)) for lo in range(0, n, cap))
        infos = [[int(x) for x in l.split()[1:]] for l in r.stdout.splitlines() if l.startswith("INFO")]
        assert infos == [[2, 0, want], [0, 2, want]] and want > 20, (infos, want)
        got = np.load(outp).reshape(n, 64)
        vcap = max((1 << 20) // 193 // 2048 * 2048, 2048)          # 4 096 verifications per super-batch
        vwant = sum(len(_schedule(min(vcap, 15000 - lo), 1024, 2048)) for lo in range(0, 15000, vcap))
        assert ("VINFO %d 1" % vwant) in r.stdout and "VOK" in r.stdout, r.stdout
        # ... and the per-chunk form of round 5 (exact items inside every chunk's launch) gives the same verdicts
        r2 = subprocess.run([sys.executable, "-c", code, outp], env=dict(env, BJJ_PIPE_VERIFY_SPLIT="0"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                            text=True, timeout=600)
        assert r2.returncode == 0 and "VOK" in r2.stdout, r2.stdout
    assert (got == oracle.mul_fixed_base(w.scalars_254(n, offset=11))).all()


def test_kernel_form_follows_pattern_and_state(oracle):
    """expect_overlap (bjj_hip.hip): launches that alternate over two streams WITHOUT a synchronisation between them get the
    forms for overlapping launches (K1: two 256-lane workgroups per CU, K2: grid-strided); the same alternation WITH a
    synchronisation after every launch -- nothing ever overlaps -- settles on the forms of a launch that runs alone from the
    second launch on; one stream never leaves them.  Results are checked in every regime."""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    dev = torch.device("cuda", 0)
    ctx = bjj.Context(0, 16)
    try:
        n = 1 << 18
        sc = w.scalars_254(n, offset=77)
        d_sc = torch.from_numpy(np.ascontiguousarray(sc).reshape(-1)).to(dev)
        d_pts = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_pts.data_ptr(), 0)
        ctx.sync()
        outs = [torch.zeros(n * 64, dtype=torch.uint8, device=dev) for _ in range(2)]
        sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        idx = np.arange(0, n, 1013)
        want_fb = oracle.mul_fixed_base(sc[idx])
        want_vb = oracle.mul_var_base(d_pts.cpu().numpy().reshape(n, 64)[idx], sc[idx])

        def fb(st, k):
            ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, outs[k].data_ptr(), st.cuda_stream)
            return ctx.info().last_fixed_base_shape

        def vb(st, k):
            ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), n, outs[k].data_ptr(), st.cuda_stream)
            return ctx.info().last_var_base_form

        # one stream: the form of a launch that runs alone (the very first launch on `sa` follows a call on the context's own
        # stream -- the set-up above -- which looks like the start of a ping-pong for exactly one launch)
        assert [fb(sa, 0) for _ in range(5)][1:] == [0, 0, 0, 0]
        ctx.sync()
        assert (outs[0].cpu().numpy().reshape(n, 64)[idx] == want_fb).all()
        assert [vb(sa, 0) for _ in range(3)] == [1, 1, 1]
        ctx.sync()
        assert (outs[0].cpu().numpy().reshape(n, 64)[idx] == want_vb).all()
        # ping-pong without synchronisation: from the second launch on the other set is busy, the pattern holds the shape
        shapes = [fb((sa, sb)[k & 1], k & 1) for k in range(8)]
        ctx.sync()
        assert shapes[1:] == [1] * 7, shapes
        # the first launch behind a synchronisation point of a caller that ping-pongs keeps the overlap shape (the pattern says
        # the next launch follows at once) ...
        assert fb(sa, 0) == 1
        assert fb(sb, 1) == 1
        ctx.sync()
        for o in outs:
            assert (o.cpu().numpy().reshape(n, 64)[idx] == want_fb).all()
        # ... but ping-pong WITH a synchronisation after every launch: the second call in a row that finds the other set idle
        # switches to the alone shape and stays there
        shapes = []
        for k in range(6):
            shapes.append(fb((sa, sb)[k & 1], k & 1))
            ctx.sync()
        assert shapes[0] == 1 and shapes[1:] == [0] * 5, shapes
        for o in outs:
            assert (o.cpu().numpy().reshape(n, 64)[idx] == want_fb).all()
        forms = []
        for k in range(5):
            forms.append(vb((sa, sb)[k & 1], k & 1))
            ctx.sync()
        assert forms[1:] == [1] * 4, forms
        for o in outs:
            assert (o.cpu().numpy().reshape(n, 64)[idx] == want_vb).all()
        # and back: without the synchronisation the overlap forms return
        forms = [vb((sa, sb)[k & 1], k & 1) for k in range(6)]
        ctx.sync()
        assert forms[1:] == [0] * 5, forms
    finally:
        ctx.close()


def test_host_verify_behind_an_unfinished_launch_that_owns_the_first_scratch_set(oracle):
    """The exact launch of a host-pointer verification of several chunks works in the first scratch set's per-lane tables without being one
    of the set's calls.  Variable-base launches the caller has enqueued on streams of their own and NOT waited for lay their tile tables
    over the same memory: the verification queues its exact launch behind the set's last foreign launch (VerifyPipe::begin).  The scenario
    -- both scratch sets owned by launches in flight -- is run and all three results are compared with runs that had the GPU to
    themselves.  (Whether a missing wait shows depends on how the hardware interleaves the launches: a build without it passed this test on
    the boxes of round 5, its exact launch running beside both others for 22 ms.  The wait is required by the set protocol all the same.)"""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    import os
    dev = torch.device("cuda", 0)
    # the tile form of K2 for every launch: its workgroups retire tile by tile, so the verification's exact workgroups DO get onto the
    # chip beside them (the grid-strided form holds every slot of the chip until it is done and would hide a missing wait)
    old_env = os.environ.get("BJJ_K2_VARIANT")
    os.environ["BJJ_K2_VARIANT"] = "1"
    try:
        ctx = bjj.Context(0, 16)
    finally:
        if old_env is None:
            del os.environ["BJJ_K2_VARIANT"]
        else:
            os.environ["BJJ_K2_VARIANT"] = old_env
    try:
        n = 1 << 20
        rng = np.random.default_rng(0x5e7)
        sc = w.scalars_254(n, offset=123)
        d_sc = torch.from_numpy(np.ascontiguousarray(sc).reshape(-1)).to(dev)
        d_pts, d_out, d_ref = (torch.empty(n * 64, dtype=torch.uint8, device=dev) for _ in range(3))
        X = torch.cuda.Stream(device=dev)
        ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_pts.data_ptr(), X.cuda_stream)          # first scratch use of the context: set 0 is X's
        ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), n, d_ref.data_ptr(), X.cuda_stream)
        ctx.sync()
        assert ctx.info().last_var_base_form == 1                                             # tiles: tables from the set's slot queue
        # signatures with many off-curve R: a long exact launch that writes tables in many slots
        m = 150000
        keys = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        msgs = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        msgs[:, 31] &= 0x1f
        pk = ctx.public_keys(keys)
        r, s, _ = ctx.sign(keys, msgs)
        r[::4, 40] ^= 2                  # R off the curve, pk on it: the exact item builds a window table of pk in its slot (verify_exact_t)
        pk[3::64, 5] ^= 2                # ... and a few of the other kind (bit-serial, no table)
        want = _verify_one_launch(ctx, False, pk, r, s, msgs)
        ctx.sync()
        hp = [_pinned_copy(ctx, a) for a in (pk, r, s, msgs)]
        ok = ctx.host_empty(m)
        ok[:] = 0xEE
        # once with the GPU to itself: the staging of the pipeline is sized now (growing it later would free device memory, which waits
        # for the whole device -- and with it for the launch this test wants to have in flight)
        ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, hp[0].ctypes.data, hp[1].ctypes.data, hp[2].ctypes.data, hp[3].ctypes.data, m, ok.ctypes.data), "bjj_eddsa_verify")
        assert (np.asarray(ok) == want).all()
        ok[:] = 0xEE
        # two launches on two streams of the caller take BOTH scratch sets (whichever is the least recently used first) and are left
        # running; the verification through host pointers follows at once
        Y = torch.cuda.Stream(device=dev)
        d_out2 = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        d_out.zero_()
        torch.cuda.synchronize()
        ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), X.cuda_stream)
        ctx.mul_var_base_dev(d_pts.data_ptr(), d_sc.data_ptr(), n, d_out2.data_ptr(), Y.cuda_stream)
        ctx._ck(ctx.lib.bjj_eddsa_verify(ctx.handle, hp[0].ctypes.data, hp[1].ctypes.data, hp[2].ctypes.data, hp[3].ctypes.data, m, ok.ctypes.data), "bjj_eddsa_verify")
        got = np.asarray(ok).copy()
        assert ctx.info().last_var_base_form == 1
        ctx.sync()
        assert (got == want).all(), int((got != want).sum())
        assert bool((d_out == d_ref).all()) and bool((d_out2 == d_ref).all())
        idx = np.arange(0, n, 40009)
        assert (d_out.cpu().numpy().reshape(n, 64)[idx] == oracle.mul_var_base(d_pts.cpu().numpy().reshape(n, 64)[idx], sc[idx])).all()
    finally:
        ctx.close()
