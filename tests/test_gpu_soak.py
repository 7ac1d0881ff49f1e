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


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
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
    # scalar multiplications on (possibly corrupted = off-curve) points with arbitrary 256-bit scalars
    pts = pk.copy()
    _flip(rng, pts, 0.1)
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    assert (gpu_ctx.mul_fixed_base(sc) == oracle.mul_fixed_base(sc)).all()
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
