"""Randomised differential soak: every C-ABI entry point against the oracle on batches of random size
with random corruption (bit flips in any input record), several rounds.  Seeded, so failures reproduce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _flip(rng, arr, frac):
    """flip one random bit in ~frac of the rows of a (n, k) uint8 array (in place); returns touched rows"""
    n, k = arr.shape
    rows = np.nonzero(rng.random(n) < frac)[0]
    for r in rows:
        bit = int(rng.integers(0, k * 8))
        arr[r, bit // 8] ^= np.uint8(1 << (bit % 8))
    return rows


def _more_seeds(default, var="BJJ_SOAK_HOST_VERIFY_SEEDS"):
    """BJJ_SOAK_HOST_VERIFY_SEEDS / BJJ_SOAK_HOST_VAR_BASE_SEEDS / BJJ_SOAK_ALL_SEEDS = first:count in the environment (developer): a longer run"""
    import os
    e = os.environ.get(var)
    if not e:
        return default
    a, b = e.split(":")
    return list(range(int(a), int(a) + int(b)))


@pytest.mark.parametrize("seed", _more_seeds([1, 2, 3, 4, 5, 6], "BJJ_SOAK_ALL_SEEDS"))
def test_soak_all_entry_points(gpu_ctx, oracle, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 1500))
    keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs[:, 31] &= 0x1f                       # mostly < Q ...
    _flip(rng, msgs, 0.05)                    # ... with a few pushed out of range or not
    # signer side
    r, s, ok = gpu_ctx.sign(keys, msgs)
    ro, so, oko = oracle.sign(keys, msgs)
    assert (ok == oko).all() and (r == ro).all() and (s == so).all()
    pk = gpu_ctx.public_keys(keys)
    assert (pk == oracle.public_keys(keys)).all()
    # the same results leaving in wire format, compression fused into the producing kernels (round 6)
    sigc, okc = gpu_ctx.sign_compressed(keys, msgs)
    wantc = np.concatenate([oracle.compress(ro), so], axis=1)
    wantc[oko == 0] = 0
    assert (okc == oko).all() and (sigc == wantc).all()
    assert (gpu_ctx.public_keys_compressed(keys) == oracle.compress(pk)).all()
    # scalar multiplications on (possibly corrupted = off-curve) points with arbitrary 256-bit scalars
    pts = pk.copy()
    _flip(rng, pts, 0.1)
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    assert (gpu_ctx.mul_fixed_base(sc) == oracle.mul_fixed_base(sc)).all()
    assert (gpu_ctx.mul_fixed_base_compressed(sc) == oracle.compress(oracle.mul_fixed_base(sc))).all()
    assert (gpu_ctx.mul_var_base(pts, sc) == oracle.mul_var_base(pts, sc)).all()
    assert (gpu_ctx.point_add(pts, pk) == oracle.point_add(pts, pk)).all()
    # raw projective add / affine on arbitrary (x, y, z) records, and scalars wider than 256 bits
    m = min(n, 300)
    pa = rng.integers(0, 256, (m, 96), dtype=np.uint8)
    pb = np.concatenate([pts[:m], rng.integers(0, 256, (m, 32), dtype=np.uint8)], axis=1)
    pa[rng.random(m) < 0.05, 64:] = 0                                        # some z == 0
    want_add = np.empty_like(pa)
    want_aff = np.empty((m, 64), np.uint8)
    for i in range(m):
        oracle.lib.bjjref_proj_add(oracle._p(pa[i]), oracle._p(pb[i]), oracle._p(want_add[i]))
        oracle.lib.bjjref_proj_affine(oracle._p(pa[i]), oracle._p(want_aff[i]))
    assert (gpu_ctx.proj_add(pa, pb) == want_add).all()
    assert (gpu_ctx.proj_affine(pa) == want_aff).all()
    import ctypes
    wb = 32 * int(rng.integers(2, 6))
    wsc = rng.integers(0, 256, (m, wb), dtype=np.uint8)
    want_w = np.empty((m, 64), np.uint8)
    for i in range(m):
        oracle.lib.bjjref_mul_scalar(oracle._p(pts[i]), oracle._p(wsc[i]), ctypes.c_size_t(wb), oracle._p(want_w[i]))
    assert (gpu_ctx.mul_var_base_wide(pts[:m], wsc, wb) == want_w).all()
    # verification with corruption anywhere
    A, R, S, M = pk.copy(), r.copy(), s.copy(), msgs.copy()
    for arr in (A, R, S, M):
        _flip(rng, arr, 0.08)
    got = gpu_ctx.eddsa_verify(A, R, S, M)
    assert (got == oracle.verify(A, R, S, M)).all()
    assert (gpu_ctx.schnorr_verify(A, R, S, M) == oracle.verify_schnorr(A, R, S, M)).all()
    # wire format
    comp = gpu_ctx.compress_points(pk)
    assert (comp == oracle.compress(pk)).all()
    _flip(rng, comp, 0.2)
    dp, dok = gpu_ctx.decompress_points(comp)
    op, ook = oracle.decompress(comp)
    assert (dok == ook).all() and (dp == op).all()
    sig = np.concatenate([gpu_ctx.compress_points(r), s], axis=1)
    _flip(rng, sig, 0.1)
    assert (gpu_ctx.eddsa_verify_compressed(comp, sig, msgs) == oracle.verify_compressed(comp, sig, msgs)).all()
    # hash
    h = rng.integers(0, 256, (n, 160), dtype=np.uint8)
    assert (gpu_ctx.poseidon5(h) == oracle.poseidon5(h)).all()


@pytest.mark.parametrize("seed", _more_seeds([11, 12, 13, 14]))
def test_soak_host_verify_across_chunks(gpu_ctx, oracle, seed):
    """The verifiers on host pointers at sizes of several chunks: a call's off-curve items run as ONE launch beside the chunks'
    bulk launches, its scans append to one batch-wide list, and that launch shares a slot queue with one lane's bulk launches
    (bjj_hip.hip: VerifyPipe).  Random size, random density of off-curve pk / R (none ... 1 in 3), each array pinned or pageable
    at random; every verdict against ONE device-pointer launch of the same inputs (exact groups inside the launch), a sample
    that contains off-curve items against the oracle."""
    import torch
    rng = np.random.default_rng(seed)
    n = int(rng.integers(100000, 400000))                        # two to three chunks of the verifiers' schedule (2^16, 2^17, 2^18 ...)
    dev = torch.device("cuda", 0)
    keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs[:, 31] &= 0x1f
    msgs[rng.random(n) < 0.01, 31] = 0xff                        # a few msg > Q (Schnorr: Err)
    pk = gpu_ctx.public_keys(keys)
    r, s, okf = gpu_ctx.sign(keys, msgs)
    dens = [0.0, 1 / 128, 1 / 16, 1 / 3][seed % 4]
    A, R, S = pk.copy(), r.copy(), s.copy()
    off_a, off_r = rng.random(n) < dens, rng.random(n) < dens / 2
    A[off_a, int(rng.integers(0, 31))] ^= np.uint8(1 << int(rng.integers(0, 8)))     # pk.x
    R[off_r, 32 + int(rng.integers(0, 31))] ^= np.uint8(1 << int(rng.integers(0, 8)))  # R.y
    S[rng.random(n) < 0.02, 3] ^= 1                               # plain wrong signatures (bulk path)
    arrays = [np.ascontiguousarray(a).reshape(-1) for a in (A, R, S, msgs)]
    d = [torch.from_numpy(a).to(dev) for a in arrays]
    d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
    for schnorr in (False, True):
        d_ok.fill_(0xAB)
        (gpu_ctx.schnorr_verify_dev if schnorr else gpu_ctx.eddsa_verify_dev)(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), n, d_ok.data_ptr(), 0)
        gpu_ctx.sync()
        want = d_ok.cpu().numpy()
        held, ptrs = [], []
        for a in arrays:                                          # pinned or pageable, per array
            if rng.random() < 0.5:
                b = gpu_ctx.host_empty(a.size)
                b[:] = a
                held.append(b)
                ptrs.append(b.ctypes.data)
            else:
                ptrs.append(a.ctypes.data)
        ok = gpu_ctx.host_empty(n) if rng.random() < 0.5 else np.empty(n, np.uint8)
        ok[:] = 0xCD
        f = gpu_ctx.lib.bjj_schnorr_verify if schnorr else gpu_ctx.lib.bjj_eddsa_verify
        gpu_ctx._ck(f(gpu_ctx.handle, ptrs[0], ptrs[1], ptrs[2], ptrs[3], n, ok.ctypes.data), "verify")
        i = gpu_ctx.info()
        assert i.last_host_chunks >= 2 and i.last_host_direct_arrays + i.last_host_staged_arrays == 5
        got = np.asarray(ok).copy()
        assert (got == want).all(), (schnorr, n, int((got != want).sum()))
        idx = np.unique(np.concatenate([np.arange(0, n, max(1, n // 40)), np.nonzero(off_a | off_r)[0][:40]]))
        ref = (oracle.verify_schnorr if schnorr else oracle.verify)(A[idx], R[idx], S[idx], msgs[idx])
        assert (got[idx] == ref).all()
        for b in held:
            gpu_ctx.host_free(b)
        if gpu_ctx.host_is_pinned(ok):
            gpu_ctx.host_free(ok)


@pytest.mark.parametrize("seed", _more_seeds([21, 22, 23, 24, 25, 26], "BJJ_SOAK_HOST_VAR_BASE_SEEDS"))
def test_soak_host_var_base_across_chunks(gpu_ctx, oracle, seed):
    """bjj_mul_var_base / _wide on host pointers at sizes of several chunks (round 6: per-chunk scans into one list, ONE exact launch per call beside
    the chunks' tile launches; a pinned output array is written by the kernels themselves, a pageable one gets the exact results laid over it by the
    host).  Random size, random density of off-curve points (none ... 1 in 3), each array pinned or pageable at random; every item against ONE
    device-pointer launch of the same inputs, the off-curve items and a stride against the oracle."""
    import ctypes
    import torch
    rng = np.random.default_rng(seed)
    n = int(rng.integers(100000, 400000))
    dev = torch.device("cuda", 0)
    pts = gpu_ctx.mul_fixed_base(rng.integers(0, 256, (n, 32), dtype=np.uint8)).copy()
    dens = [0.0, 1 / 4096, 1 / 128, 1 / 16, 1 / 3, 1 / 1000][seed % 6]
    off = rng.random(n) < dens
    pts[off, int(rng.integers(0, 64))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    wide = int(rng.integers(1, 4)) if seed % 2 else 1               # 32-byte or 64 / 96-byte scalars
    sc = rng.integers(0, 256, (n, 32 * wide), dtype=np.uint8)
    d_p, d_s = torch.from_numpy(pts.reshape(-1)).to(dev), torch.from_numpy(sc.reshape(-1)).to(dev)
    d_o = torch.full((n * 64,), 0xEE, dtype=torch.uint8, device=dev)
    if wide == 1:
        gpu_ctx.mul_var_base_dev(d_p.data_ptr(), d_s.data_ptr(), n, d_o.data_ptr())
    else:
        gpu_ctx.mul_var_base_wide_dev(d_p.data_ptr(), d_s.data_ptr(), 32 * wide, n, d_o.data_ptr())
    gpu_ctx.sync()
    want = d_o.cpu().numpy().reshape(n, 64)
    held, ptrs, pinned = [], [], []
    for a in (pts.reshape(-1), sc.reshape(-1)):
        if rng.random() < 0.5:
            b = gpu_ctx.host_empty(a.size); b[:] = a; held.append(b); ptrs.append(b.ctypes.data); pinned.append(True)
        else:
            ptrs.append(a.ctypes.data); pinned.append(False)
    out_pinned = bool(rng.random() < 0.5)
    out = gpu_ctx.host_empty(n * 64) if out_pinned else np.empty(n * 64, np.uint8)
    out[:] = 0xCD
    if wide == 1:
        rc = gpu_ctx.lib.bjj_mul_var_base(gpu_ctx.handle, ptrs[0], ptrs[1], ctypes.c_size_t(n), out.ctypes.data)
    else:
        rc = gpu_ctx.lib.bjj_mul_var_base_wide(gpu_ctx.handle, ptrs[0], ptrs[1], ctypes.c_size_t(32 * wide), ctypes.c_size_t(n), out.ctypes.data)
    assert rc == 0, gpu_ctx.lib.bjj_last_error()
    i = gpu_ctx.info()
    import os
    zc_on = os.environ.get("BJJ_PIPE_ZERO_COPY", "1") != "0"         # (developer runs of this soak with the copy-out stage forced)
    assert i.last_host_chunks >= 2 and i.last_var_base_split == 1 and i.last_host_zero_copy == (1 if out_pinned and zc_on else 0)
    assert i.last_host_direct_arrays == sum(pinned) + out_pinned
    got = np.asarray(out).reshape(n, 64).copy()
    assert (got == want).all(), (n, dens, wide, int((got != want).any(axis=1).sum()))
    idx = np.unique(np.concatenate([np.arange(0, n, max(1, n // 60)), np.nonzero(off)[0][:60]]))
    ref = np.empty((idx.size, 64), np.uint8)
    for k, j in enumerate(idx):
        oracle.lib.bjjref_mul_scalar(oracle._p(pts[j]), oracle._p(sc[j]), ctypes.c_size_t(32 * wide), oracle._p(ref[k]))
    assert (got[idx] == ref).all()
    for b in held:
        gpu_ctx.host_free(b)
    if out_pinned:
        gpu_ctx.host_free(out)


@pytest.mark.parametrize("seed", _more_seeds([31, 32, 33, 34], "BJJ_SOAK_HOST_VERIFY_COMPRESSED_SEEDS"))
def test_soak_host_verify_compressed_across_chunks(gpu_ctx, oracle, seed):
    """bjj_eddsa_verify_compressed on host pointers at sizes of two to three chunks (round 6: per-chunk decompressions, one batch-wide list, ONE exact launch
    and ONE flag pass per call): random size, random density of corruption anywhere in the wire-format records (a corrupted compressed point decompresses to
    another point or not at all), each array pinned or pageable at random; every verdict against ONE device-pointer launch, a sample against the oracle."""
    import ctypes
    import torch
    rng = np.random.default_rng(seed)
    n = int(rng.integers(100000, 400000))
    dev = torch.device("cuda", 0)
    keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs[:, 31] &= 0x1f
    pkc = gpu_ctx.public_keys_compressed(keys)
    sig, okf = gpu_ctx.sign_compressed(keys, msgs)
    dens = [0.0, 1 / 128, 1 / 16, 1 / 3][seed % 4]
    touched = np.zeros(n, bool)
    for arr in (pkc, sig, msgs):
        rows = np.nonzero(rng.random(n) < dens / 3)[0]
        for r_ in rows:
            bit = int(rng.integers(0, arr.shape[1] * 8))
            arr[r_, bit // 8] ^= np.uint8(1 << (bit % 8))
        touched[rows] = True
    d = [torch.from_numpy(a.reshape(-1)).to(dev) for a in (pkc, sig, msgs)]
    d_ok = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
    gpu_ctx.eddsa_verify_compressed_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), n, d_ok.data_ptr())
    gpu_ctx.sync()
    want = d_ok.cpu().numpy()
    held, ptrs = [], []
    for a in (pkc.reshape(-1), sig.reshape(-1), msgs.reshape(-1)):
        if rng.random() < 0.5:
            b = gpu_ctx.host_empty(a.size); b[:] = a; held.append(b); ptrs.append(b.ctypes.data)
        else:
            ptrs.append(a.ctypes.data)
    ok = gpu_ctx.host_empty(n) if rng.random() < 0.5 else np.empty(n, np.uint8)
    ok[:] = 0xCD
    assert gpu_ctx.lib.bjj_eddsa_verify_compressed(gpu_ctx.handle, ptrs[0], ptrs[1], ptrs[2], ctypes.c_size_t(n), ok.ctypes.data) == 0
    assert gpu_ctx.info().last_host_chunks >= 2
    got = np.asarray(ok).copy()
    assert (got == want).all(), (n, dens, int((got != want).sum()))
    assert (got[~touched] == 1).all()
    idx = np.unique(np.concatenate([np.arange(0, n, max(1, n // 60)), np.nonzero(touched)[0][:80]]))
    assert (got[idx] == oracle.verify_compressed(pkc[idx], sig[idx], msgs[idx])).all()
    for b in held:
        gpu_ctx.host_free(b)


@pytest.mark.parametrize("seed", _more_seeds([41, 42, 43, 44, 45, 46], "BJJ_SOAK_SHORT_VAR_BASE_SEEDS"))
def test_soak_short_var_base_calls(gpu_ctx, oracle, seed):
    """short Point::mul_scalar calls (four lanes per item, csrc/k_small.hip): a seeded size, points that are multiples of G (order 8l) rather than of B8,
    raw 256-bit scalars, a seeded share of off-curve and of small-order points -- every item against the oracle, host pointers (pageable) and a device launch"""
    import torch
    rng = np.random.default_rng(0x51ab0000 + seed)
    n = int(rng.integers(1, 1500))
    g = (995203441582195749578291179787384436505546430278305826713579947235728471134, 5472060717959818805561601436314318772137091100104008585924551046643952123905)
    G = np.frombuffer(g[0].to_bytes(32, "little") + g[1].to_bytes(32, "little"), np.uint8)
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x3f
    ks[rng.random(n) < 0.05] = 0                                                  # the identity as a base point
    l8 = np.frombuffer((2736030358979909402780800718157159386076813972158567259200215660948447373041).to_bytes(32, "little"), np.uint8)
    ks[rng.random(n) < 0.05] = l8                                                  # l * G: a point of order 8
    pts = oracle.mul_var_base(np.tile(G, (n, 1)), ks).copy()
    off = rng.random(n) < float(rng.choice([0.0, 0.01, 0.3]))
    pts[off, int(rng.integers(0, 31))] ^= 1 << int(rng.integers(0, 8))
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)                             # raw: bits 254 and 255 set on three quarters of them
    want = oracle.mul_var_base(pts, sc)
    got = gpu_ctx.mul_var_base(pts, sc)
    assert gpu_ctx.info().last_var_base_form == 2
    assert (got == want).all(), (seed, n, np.nonzero((got != want).any(axis=1))[0][:8])
    dev = torch.device("cuda", 0)
    d_p, d_s = torch.from_numpy(pts.reshape(-1)).to(dev), torch.from_numpy(sc.reshape(-1)).to(dev)
    d_o = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
    gpu_ctx.mul_var_base_dev(d_p.data_ptr(), d_s.data_ptr(), n, d_o.data_ptr())
    gpu_ctx.sync()
    assert (d_o.cpu().numpy().reshape(n, 64) == want).all(), (seed, n)
