"""Round 6: (a) the wire-format OUTPUT path -- Point::compress / Signature::compress (src/lib.rs:166-178, 245-258) fused into the
producing kernels -- and (b) off-curve points on the variable-base path (Point has pub fields and no check, src/lib.rs:134-138)
with the exact kernel BESIDE the batch kernel: the device-pointer forms by the context's history, the host-pointer pipeline as one
exact launch per call whose results the host lays over the caller's array.  Needs a real MI355X: `pytest -m gpu`."""
import ctypes
import os

import numpy as np
import pytest

from conftest import pack, ints

pytestmark = pytest.mark.gpu

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
N = 1 << 20


def _ref_compress(x, y):
    """Point::compress, src/lib.rs:166-178, on plain integers (the test's own restatement: 32 bytes of y, bit 255 = x > Q >> 1)"""
    b = bytearray(int(y).to_bytes(32, "little"))
    if x > (Q >> 1):
        b[31] |= 0x80
    return bytes(b)


def _pinned_copy(ctx, a):
    p = ctx.host_empty(a.size)
    p[:] = np.ascontiguousarray(a).reshape(-1)
    return p


# ------------------------------------------------------------------------------------------------ compressed outputs
BOTH_TABLES = pytest.mark.parametrize("ctx_for_window", [23, 28], indirect=True, ids=["window_bits_23", "window_bits_28"])


@BOTH_TABLES
def test_fixed_base_compressed_is_compress_of_the_affine_result_every_item(ctx_for_window, oracle):
    """2^20 items: bjj_mul_fixed_base_compressed == bjj_compress_points(bjj_mul_fixed_base(..)) byte for byte -- pageable and pinned
    host arrays, one device-pointer launch in each of K1's shapes -- and a sample against the oracle's compress of the oracle's points"""
    import torch
    from babyjubjub_rs_amd import workload as w
    ctx = ctx_for_window
    sc = w.scalars_254(N)
    sc[7] = 0                                                    # the identity: (0, 1) -> 01 00 .. 00
    want = ctx.compress_points(ctx.mul_fixed_base(sc))
    got = ctx.mul_fixed_base_compressed(sc)
    assert got.shape == (N, 32) and (got == want).all(), np.nonzero((got != want).any(axis=1))[0][:8]
    assert bytes(got[7]) == _ref_compress(0, 1)
    psc, pout = _pinned_copy(ctx, sc), ctx.host_empty(N * 32)
    pout[:] = 0xAB
    ctx._ck(ctx.lib.bjj_mul_fixed_base_compressed(ctx.handle, psc.ctypes.data, N, pout.ctypes.data), "bjj_mul_fixed_base_compressed")
    i = ctx.info()
    assert (i.last_host_direct_arrays, i.last_host_staged_arrays) == (2, 0) and (np.asarray(pout).reshape(N, 32) == want).all()
    ctx.host_free(psc); ctx.host_free(pout)
    idx = np.arange(0, N, 2053)
    assert (got[idx] == oracle.compress(oracle.mul_fixed_base(sc[idx]))).all()
    dev = torch.device("cuda", 0)
    d_sc = torch.from_numpy(sc.reshape(-1)).to(dev)
    d_out = [torch.full((N * 32,), 0xCD, dtype=torch.uint8, device=dev) for _ in range(2)]
    st = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for _ in range(3):          # one stream, a synchronisation between the launches: the form for a launch that runs alone (one 512-lane workgroup per CU)
        ctx.mul_fixed_base_compressed_dev(d_sc.data_ptr(), N, d_out[0].data_ptr(), st[0].cuda_stream)
        ctx.sync()
    assert ctx.info().last_fixed_base_shape == 0
    assert (d_out[0].cpu().numpy().reshape(N, 32) == want).all()
    d_out[0].fill_(0xCD)
    for k in range(4):                                                                                  # two streams: two of 256 lanes
        ctx.mul_fixed_base_compressed_dev(d_sc.data_ptr(), N, d_out[k % 2].data_ptr(), st[k % 2].cuda_stream)
    assert ctx.info().last_fixed_base_shape == 1
    ctx.sync()
    for o in d_out:
        assert (o.cpu().numpy().reshape(N, 32) == want).all()


def test_compressed_outputs_small_and_ragged_sizes(gpu_ctx, oracle):
    from babyjubjub_rs_amd import workload as w
    for n in (1, 2, 63, 64, 65, 511, 513, 40001):
        sc = w.scalars_254(n, offset=3 * n)
        got = gpu_ctx.mul_fixed_base_compressed(sc)
        assert (got == oracle.compress(oracle.mul_fixed_base(sc))).all(), n
    assert gpu_ctx.mul_fixed_base_compressed(np.zeros((0, 32), np.uint8)).shape == (0, 32)


def test_circomlib_vector_through_the_fused_outputs(gpu_ctx, golden):
    """src/lib.rs:692-738: the circomlib key -> sk.public().compress() and sk.sign(msg).compress(); expected bytes = the reference's
    own pk / R8 / S values put through the definition of compress (:166-178, :245-258)"""
    k = golden["reference_kats"]["circomlib_testvector"]
    key = np.frombuffer(bytes.fromhex(k["key"]), np.uint8)
    pkx, pky = ints(k["pk"][0]), ints(k["pk"][1])
    rx, ry, s = ints(k["r_b8"][0]), ints(k["r_b8"][1]), ints(k["s"])
    msg = ints(k["msg"])
    assert bytes(gpu_ctx.public_keys_compressed(key)[0]) == _ref_compress(pkx, pky)
    sig, ok = gpu_ctx.sign_compressed(key, pack([msg]))
    assert ok[0] == 1 and bytes(sig[0]) == _ref_compress(rx, ry) + s.to_bytes(32, "little")
    # ... and the wire-format verifier accepts what the wire-format signer produced
    assert gpu_ctx.eddsa_verify_compressed(gpu_ctx.public_keys_compressed(key), sig, pack([msg]))[0] == 1


@pytest.mark.parametrize("ct", [False, True], ids=["table_indexed", "signer_constant_time"])
def test_public_keys_and_sign_compressed_equal_the_two_pass_form(oracle, ct):
    import babyjubjub_rs_amd as bjj
    ctx = bjj.Context(0, 16)
    try:
        if ct:
            ctx.set_signer_constant_time(True)
        n = 70001 if not ct else 9001
        rng = np.random.default_rng(0x636f6d70)
        keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        msgs[:, 31] &= 0x1f
        msgs[::89, 31] = 0xff                                    # msg > Q: the reference returns Err
        pk = ctx.public_keys(keys)
        assert (ctx.public_keys_compressed(keys) == ctx.compress_points(pk)).all()
        r, s, ok = ctx.sign(keys, msgs)
        sig, okc = ctx.sign_compressed(keys, msgs)
        assert (okc == ok).all() and (ok[::89] == 0).all() and ok.sum() == n - len(range(0, n, 89))
        want = np.concatenate([ctx.compress_points(r), s], axis=1)
        want[ok == 0] = 0                                        # Err rows leave all-zero (compress of the zeroed R would be 32 zero bytes too)
        assert (sig == want).all()
        idx = np.arange(0, n, 997)
        ro, so, oko = oracle.sign(keys[idx], msgs[idx])
        wo = np.concatenate([oracle.compress(ro), so], axis=1)
        wo[oko == 0] = 0
        assert (sig[idx] == wo).all() and (ctx.public_keys_compressed(keys)[idx] == oracle.compress(oracle.public_keys(keys[idx]))).all()
        good = ok == 1
        assert (ctx.eddsa_verify_compressed(ctx.public_keys_compressed(keys)[good], sig[good], msgs[good]) == 1).all()
    finally:
        ctx.close()


def _host_call(ctx, name, arrays_in, n, arrays_out):
    import ctypes as C
    args = [ctx.handle] + [a.ctypes.data for a in arrays_in] + [C.c_size_t(n)] + [a.ctypes.data for a in arrays_out]
    ctx._ck(getattr(ctx.lib, name)(*args), name)
    return ctx.info()


def test_k1_host_calls_on_pinned_memory_every_chunk_count(gpu_ctx, oracle):
    """The K1 entry points on pinned arrays (PipeSpec::zero_copy_in / k1_half, bjj_hip.hip): calls of one or two chunks read the
    caller's array through its device mapping (bjj_info.last_host_zero_copy bit 1), the 32-byte forms run their chunk launches on one
    workgroup slot per CU from two chunks on; every size byte for byte what the pageable call and one device-resident launch give"""
    from babyjubjub_rs_amd import workload as w
    ctx = gpu_ctx
    nmax = (1 << 19) + 70001
    keys = w.scalars_254(nmax, offset=77)
    keys[5] = 0
    p_in = _pinned_copy(ctx, keys)
    want_fb, want_pk = ctx.mul_fixed_base(keys), ctx.public_keys(keys)          # pageable: staged copies, full launches per chunk
    want = {"bjj_mul_fixed_base": (want_fb, 64), "bjj_mul_fixed_base_compressed": (ctx.compress_points(want_fb), 32),
            "bjj_public_keys": (want_pk, 64), "bjj_public_keys_compressed": (ctx.compress_points(want_pk), 32)}
    idx = np.arange(0, nmax, 4099)
    assert (want_fb[idx] == oracle.mul_fixed_base(keys[idx])).all() and (want_pk[idx] == oracle.public_keys(keys[idx])).all()
    for name, (exp, width) in want.items():
        p_out = ctx.host_empty(nmax * width)
        for n in (1, 63, 4097, 1 << 16, (1 << 16) + 1, (1 << 17) + 5, 3 << 16, nmax):
            p_out[:] = 0xEE
            i = _host_call(ctx, name, [p_in], n, [p_out])
            got = np.asarray(p_out[:n * width]).reshape(n, width)
            assert (got == exp[:n]).all(), (name, n, np.nonzero((got != exp[:n]).any(axis=1))[0][:8])
            assert (np.asarray(p_out[n * width:n * width + 64]) == 0xEE).all(), (name, n)                 # nothing beyond item n
            assert (i.last_host_direct_arrays, i.last_host_staged_arrays) == (2, 0)
            assert bool(i.last_host_zero_copy & 2) == (i.last_host_chunks <= 2), (name, n, i.last_host_chunks, i.last_host_zero_copy)
            if width == 32 and not any(k in os.environ for k in ("BJJ_PIPE_K1_HALF", "BJJ_PIPE_CHUNK", "BJJ_PIPE_FIRST_CHUNK", "BJJ_PIPE_SCHEDULE")):
                # 2^16, then 2^17 each (a remainder below half a chunk joins the last one); from two chunks on in the two-workgroup shape
                assert i.last_host_chunks == {1 << 16: 1, (1 << 16) + 1: 1, (1 << 17) + 5: 2, 3 << 16: 2, nmax: 5}.get(n, 1), (name, n, i.last_host_chunks)
                if i.last_host_chunks >= 2 and "BJJ_K1_VARIANT" not in os.environ:
                    assert i.last_fixed_base_shape == 1
        ctx.host_free(p_out)
    ctx.host_free(p_in)


def test_k1_host_calls_fall_back_to_copies_for_a_misaligned_pinned_array(gpu_ctx):
    """a pinned input that does not start on a 16-byte boundary cannot be read by the kernels in place (they move 16-byte words): the
    copy engines realign it into the staging area, as before"""
    from babyjubjub_rs_amd import workload as w
    ctx = gpu_ctx
    n = 5001
    sc = w.scalars_254(n, offset=9)
    raw = ctx.host_empty(n * 32 + 16)
    mis = raw[8:8 + n * 32]
    mis[:] = sc.reshape(-1)
    out = ctx.host_empty(n * 32)
    i = _host_call(ctx, "bjj_mul_fixed_base_compressed", [mis], n, [out])
    assert i.last_host_zero_copy == 0 and i.last_host_direct_arrays == 2
    assert (np.asarray(out).reshape(n, 32) == ctx.compress_points(ctx.mul_fixed_base(sc))).all()
    ctx.host_free(raw); ctx.host_free(out)


def test_compressed_entry_points_reject_bad_arguments(gpu_ctx):
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    buf = np.zeros(64, np.uint8)
    for name, args in (("bjj_mul_fixed_base_compressed", (None, 1, buf.ctypes.data)), ("bjj_mul_fixed_base_compressed", (buf.ctypes.data, 1, None)),
                       ("bjj_public_keys_compressed", (None, 1, buf.ctypes.data)), ("bjj_sign_compressed", (buf.ctypes.data, buf.ctypes.data, 1, None, buf.ctypes.data)),
                       ("bjj_mul_fixed_base_compressed_dev", (8, 1, 16, None)), ("bjj_sign_compressed_dev", (16, 16, 1, 16, None, None))):
        assert getattr(lib, name)(h, *args) == -1, name
    assert lib.bjj_mul_fixed_base_compressed(h, None, 0, None) == 0            # n == 0: nothing to do, like every entry point


# ------------------------------------------------------------------------------------------------ off-curve points, variable base
def _points(ctx, n, density, seed):
    """n group points; one in `density` pushed off the curve (density 0: none), at seeded positions"""
    from babyjubjub_rs_amd import workload as w
    pts = ctx.mul_fixed_base(w.scalars_254(n, offset=seed)).copy()
    bad = np.zeros(n, bool)
    if density:
        bad = (w.splitmix64(0x0ffc + seed, n, 0) % np.uint64(density)) == 0
        pts[bad, 0] ^= 1
    return pts, bad


def _one_launch(ctx, pts, sc, scalar_bytes=32):
    import torch
    dev = torch.device("cuda", 0)
    n = pts.shape[0]
    d_p, d_s = torch.from_numpy(pts.reshape(-1)).to(dev), torch.from_numpy(np.ascontiguousarray(sc).reshape(-1)).to(dev)
    d_o = torch.full((n * 64,), 0xEE, dtype=torch.uint8, device=dev)
    if scalar_bytes == 32:
        ctx.mul_var_base_dev(d_p.data_ptr(), d_s.data_ptr(), n, d_o.data_ptr())
    else:
        ctx.mul_var_base_wide_dev(d_p.data_ptr(), d_s.data_ptr(), scalar_bytes, n, d_o.data_ptr())
    ctx.sync()
    return d_o.cpu().numpy().reshape(n, 64)


def _oracle_wide(oracle, pts, sc, nbytes):
    want = np.empty((pts.shape[0], 64), np.uint8)
    for i in range(pts.shape[0]):
        oracle.lib.bjjref_mul_scalar(oracle._p(pts[i]), oracle._p(sc[i]), ctypes.c_size_t(nbytes), oracle._p(want[i]))
    return want


def test_device_calls_move_the_exact_kernel_beside_the_batch_kernel_by_history(oracle):
    """a context's history decides (bjj_get_info: last_var_base_split): clean -> K6 behind K2; the first call AFTER one that met an
    off-curve point -> scan + K6 beside K2; back once a completed call found none.  Every item of every call against the oracle."""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    ctx = bjj.Context(0, 16)
    try:
        n = 30011
        sc = w.scalars_254(n, offset=11)
        clean, _ = _points(ctx, n, 0, 1)
        dirty, bad = _points(ctx, n, 700, 2)
        assert 20 < bad.sum() < 80
        want_clean, want_dirty = oracle.mul_var_base(clean, sc), oracle.mul_var_base(dirty, sc)
        seq = [(clean, want_clean, 0), (dirty, want_dirty, 0), (dirty, want_dirty, 1), (clean, want_clean, 1), (clean, want_clean, 0),
               (dirty, want_dirty, 0), (clean, want_clean, 1), (dirty, want_dirty, 0)]
        for k, (p, want, split) in enumerate(seq):
            got = _one_launch(ctx, p, sc)                 # synchronises: the next call sees what this one met
            assert ctx.info().last_var_base_split == split, k
            assert (got == want).all(), (k, np.nonzero((got != want).any(axis=1))[0][:8])
    finally:
        ctx.close()


@pytest.mark.parametrize("force", ["0", "1"], ids=["exact_behind", "exact_beside"])
@pytest.mark.parametrize("k2", ["0", "1"], ids=["grid_strided", "tiles"])
def test_forced_forms_agree_with_the_oracle_at_every_density(oracle, monkeypatch, force, k2):
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    monkeypatch.setenv("BJJ_VB_SPLIT", force)
    monkeypatch.setenv("BJJ_K2_VARIANT", k2)
    ctx = bjj.Context(0, 16)
    monkeypatch.delenv("BJJ_VB_SPLIT"); monkeypatch.delenv("BJJ_K2_VARIANT")
    try:
        for n, density in ((1, 1), (1, 0), (64, 1), (777, 2), (20001, 3), (20001, 64), (50003, 4096), (50003, 0)):
            sc = w.scalars_254(n, offset=n + density)
            pts, bad = _points(ctx, n, density, n)
            got = _one_launch(ctx, pts, sc)
            assert ctx.info().last_var_base_split == int(force) and ctx.info().last_var_base_form == int(k2)
            want = oracle.mul_var_base(pts, sc)
            assert (got == want).all(), (n, density, np.nonzero((got != want).any(axis=1))[0][:8])
        # wide scalars take the same route (the exact kernel replays all n.bits() bits)
        n, nbytes = 3001, 96
        wsc = w.random_u256(0x77, 3 * n).reshape(n, nbytes).copy()
        pts, bad = _points(ctx, n, 5, 99)
        got = _one_launch(ctx, pts, wsc, nbytes)
        sel = np.unique(np.concatenate([np.nonzero(bad)[0][:150], np.arange(0, n, 37)]))
        assert (got[sel] == _oracle_wide(oracle, pts[sel], wsc[sel], nbytes)).all()
    finally:
        ctx.close()


def test_two_streams_with_off_curve_points_in_both_batches(oracle):
    """overlapping launches, both sets in the beside form (their scans and exact kernels on the sets' own priority streams)"""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    ctx = bjj.Context(0, 16)
    try:
        dev = torch.device("cuda", 0)
        n = 1 << 17
        batches = []
        for b in range(2):
            pts, bad = _points(ctx, n, 512, 40 + b)
            sc = w.scalars_254(n, offset=1000 * b)
            batches.append((pts, sc, torch.from_numpy(pts.reshape(-1)).to(dev), torch.from_numpy(sc.reshape(-1)).to(dev),
                            torch.zeros(n * 64, dtype=torch.uint8, device=dev)))
        st = [torch.cuda.Stream(device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        for rnd in range(3):               # after the first synchronisation: beside, on both sets
            for b in range(2):
                ctx.mul_var_base_dev(batches[b][2].data_ptr(), batches[b][3].data_ptr(), n, batches[b][4].data_ptr(), st[b].cuda_stream)
            # (round 0: the first launch knows nothing yet; the second may already see what the first met -- sizing the second scratch set synchronises)
            assert rnd == 0 or ctx.info().last_var_base_split == 1
            ctx.sync()
            for b in range(2):
                got = batches[b][4].cpu().numpy().reshape(n, 64)
                idx = np.unique(np.concatenate([np.arange(0, n, 257), np.nonzero((w.splitmix64(0x0ffc + 40 + b, n, 0) % np.uint64(512)) == 0)[0]]))
                assert (got[idx] == oracle.mul_var_base(batches[b][0][idx], batches[b][1][idx])).all(), (rnd, b)
                batches[b][4].zero_()
    finally:
        ctx.close()


DENSITIES = [0, 1 << 14, 4096, 256, 16, 3]


@pytest.mark.parametrize("mem", ["pinned", "pageable"])
def test_host_pipeline_every_density_every_item(oracle, mem):
    """bjj_mul_var_base / _wide on host pointers, none .. one off-curve point in 3, three chunks over both lanes (and a ragged tail):
    every item against ONE device-pointer launch of the same inputs, a sample (off-curve items included) against the oracle"""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    ctx = bjj.Context(0, 16)
    try:
        n = (1 << 16) + (1 << 17) + 70003
        for density in DENSITIES:
            pts, bad = _points(ctx, n, density, density + 5)
            sc = w.scalars_254(n, offset=density)
            wsc = np.concatenate([sc, w.random_u256(0x99 + density, n)], axis=1)
            alloc = ctx.host_empty if mem == "pinned" else (lambda nb: np.zeros(nb, np.uint8))
            for wide in (False, True):
                s_in = wsc if wide else sc
                a_p, a_s, a_o = alloc(n * 64), alloc(s_in.size), alloc(n * 64)
                a_p[:] = pts.reshape(-1); a_s[:] = s_in.reshape(-1); a_o[:] = 0xAB
                if wide:
                    rc = ctx.lib.bjj_mul_var_base_wide(ctx.handle, a_p.ctypes.data, a_s.ctypes.data, ctypes.c_size_t(64), ctypes.c_size_t(n), a_o.ctypes.data)
                else:
                    rc = ctx.lib.bjj_mul_var_base(ctx.handle, a_p.ctypes.data, a_s.ctypes.data, ctypes.c_size_t(n), a_o.ctypes.data)
                assert rc == 0, ctx.lib.bjj_last_error()
                i = ctx.info()
                assert i.last_host_chunks == 3 and i.last_var_base_split == 1
                assert (i.last_host_direct_arrays, i.last_host_staged_arrays) == ((3, 0) if mem == "pinned" else (0, 3))
                assert i.last_host_zero_copy == (1 if mem == "pinned" else 0)      # pinned output: the kernels (K2 and K6) store into it themselves
                got = np.asarray(a_o).reshape(n, 64).copy()
                ref = _one_launch(ctx, pts, s_in, 64 if wide else 32)
                assert (got == ref).all(), (density, wide, np.nonzero((got != ref).any(axis=1))[0][:8], int(bad.sum()))
                sel = np.unique(np.concatenate([np.nonzero(bad)[0][:200], np.arange(0, n, 4001), [65535, 65536, 196607, 196608, n - 1]]))
                want = _oracle_wide(oracle, pts[sel], wsc[sel], 64) if wide else oracle.mul_var_base(pts[sel], sc[sel])
                assert (got[sel] == want).all(), (density, wide)
                if mem == "pinned":
                    for b in (a_p, a_s, a_o):
                        ctx.host_free(b)
    finally:
        ctx.close()


def test_host_pipeline_super_batches_patch_their_own_range(oracle, monkeypatch):
    """a staging budget that cuts the call into super-batches: each runs its own scan list / exact launch / patch over ITS part of the array"""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    monkeypatch.setenv("BJJ_PIPE_STAGING_MB", "64")              # 224 B per item -> 262144-item super-batches
    ctx = bjj.Context(0, 16)
    try:
        n = 3 * 262144 + 777
        pts, bad = _points(ctx, n, 1000, 17)
        sc = w.scalars_254(n, offset=9)
        got = ctx.mul_var_base(pts, sc)
        assert ctx.info().last_host_chunks >= 6
        ref = _one_launch(ctx, pts, sc)
        assert (got == ref).all()
        sel = np.nonzero(bad)[0][::3]
        assert (got[sel] == oracle.mul_var_base(pts[sel], sc[sel])).all()
    finally:
        ctx.close()


def test_host_pipeline_through_the_copy_engines_when_zero_copy_is_off(oracle, monkeypatch):
    """BJJ_PIPE_ZERO_COPY=0: pinned outputs leave by D2H copies chunk by chunk and the exact kernel's results are laid over them by the
    host afterwards -- the path pageable outputs always take; a pinned INPUT with a pageable OUTPUT takes it too"""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    n = (1 << 16) + (1 << 17) + 70003
    for env in ("0", None):
        if env is not None:
            monkeypatch.setenv("BJJ_PIPE_ZERO_COPY", env)
        ctx = bjj.Context(0, 16)
        try:
            pts, bad = _points(ctx, n, 300, 7)
            sc = w.scalars_254(n, offset=3)
            a_p, a_s = _pinned_copy(ctx, pts), _pinned_copy(ctx, sc)
            a_o = ctx.host_empty(n * 64) if env is not None else np.zeros(n * 64, np.uint8)
            a_o[:] = 0xAB
            assert ctx.lib.bjj_mul_var_base(ctx.handle, a_p.ctypes.data, a_s.ctypes.data, ctypes.c_size_t(n), a_o.ctypes.data) == 0
            i = ctx.info()
            assert i.last_host_zero_copy == 0 and i.last_var_base_split == 1 and i.last_host_chunks == 3
            got = np.asarray(a_o).reshape(n, 64)
            assert (got == _one_launch(ctx, pts, sc)).all()
            sel = np.nonzero(bad)[0]
            assert sel.size > 500 and (got[sel] == oracle.mul_var_base(pts[sel], sc[sel])).all()
        finally:
            ctx.close()
        if env is not None:
            monkeypatch.delenv("BJJ_PIPE_ZERO_COPY")


# ------------------------------------------------------------------------------------------------ the wire-format verifier on host pointers
@pytest.mark.parametrize("mem", ["pinned", "pageable"])
def test_host_verify_compressed_across_chunks_every_verdict(oracle, mem):
    """bjj_eddsa_verify_compressed on host pointers at three chunks: per chunk decompress -> scan (one batch-wide list) -> bulk launch, ONE exact launch and ONE
    flag pass per call.  Inputs: valid wire-format signatures with, at random, a flipped bit in s / msg (plain wrong), in the compressed pk or R (decompresses
    to another point, does not decompress at all -> verdict 2, or lands off... a compressed point that decompresses is ON the curve), and msg > Q.  Every verdict
    against ONE device-pointer launch of the same inputs; a sample incl. every kind against the oracle; the per-chunk form (BJJ_PIPE_VERIFY_SPLIT=0) agrees."""
    import babyjubjub_rs_amd as bjj
    import torch
    ctx = bjj.Context(0, 16)
    try:
        n = (1 << 16) + (1 << 17) + 70001
        rng = np.random.default_rng(0x77697265)
        keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        msgs[:, 31] &= 0x1f
        pkc = ctx.public_keys_compressed(keys)
        sig, okf = ctx.sign_compressed(keys, msgs)
        assert okf.all()
        kind = rng.integers(0, 400, n)
        pkc[kind == 1, 3] ^= 1; pkc[kind == 2, 31] ^= 0x40            # pk: another y (may or may not decompress), a y >= r
        sig[kind == 3, 5] ^= 2; sig[kind == 4, 31] ^= 0x80            # R: another y; the sign bit of x
        sig[kind == 5, 40] ^= 1                                        # s
        msgs[kind == 6, 0] ^= 1; msgs[kind == 7, 31] = 0xff            # msg; msg > Q
        sig[kind == 8, :32] = 0xff                                     # R does not decompress
        dev = torch.device("cuda", 0)
        d = [torch.from_numpy(a.reshape(-1)).to(dev) for a in (pkc, sig, msgs)]
        d_ok = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
        ctx.eddsa_verify_compressed_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, d_ok.data_ptr())
        ctx.sync()
        want = d_ok.cpu().numpy()
        assert set(np.unique(want)) == {0, 1, 2} and (want[kind >= 9] == 1).all() and (want[kind == 8] == 2).all() and (want[kind == 7] == 0).all()
        alloc = ctx.host_empty if mem == "pinned" else (lambda nb: np.zeros(nb, np.uint8))
        bufs = []
        for a in (pkc, sig, msgs):
            b = alloc(a.size); b[:] = a.reshape(-1); bufs.append(b)
        ok = alloc(n); ok[:] = 0xCD
        assert ctx.lib.bjj_eddsa_verify_compressed(ctx.handle, bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data, ctypes.c_size_t(n), ok.ctypes.data) == 0
        i = ctx.info()
        assert i.last_host_chunks == 3 and i.last_verify_dispatch == 1
        assert (i.last_host_direct_arrays, i.last_host_staged_arrays) == ((4, 0) if mem == "pinned" else (0, 4))
        got = np.asarray(ok).copy()
        assert (got == want).all(), (int((got != want).sum()), np.nonzero(got != want)[0][:8], got[got != want][:8], want[got != want][:8])
        sel = np.unique(np.concatenate([np.nonzero(kind < 9)[0][:600], np.arange(0, n, 997), [65535, 65536, 196607, 196608, n - 1]]))
        assert (got[sel] == oracle.verify_compressed(pkc[sel], sig[sel], msgs[sel])).all()
    finally:
        ctx.close()
