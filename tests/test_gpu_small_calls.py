"""Short calls of Point::mul_scalar (src/lib.rs:149-164): four lanes per item (csrc/k_small.hip: bjj_k_mul_var_base_quad) instead of K2's one --
the same group element through the same formulas, so the canonical output must be byte-identical to K2's and to the oracle's, for every
kind of point and scalar, at every batch size around the quad / wave / switch-over boundaries, on device and host pointers, with
off-curve points handed to the exact kernel K6 in both of its positions.  Needs a real MI355X: `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest

from conftest import pack

pytestmark = pytest.mark.gpu

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
L = 2736030358979909402780800718157159386076813972158567259200215660948447373041
QUAD_MAX = 1 << 14


def _launch(ctx, pts, sc):
    import torch
    dev = torch.device("cuda", 0)
    n = pts.shape[0]
    d_p, d_s = torch.from_numpy(np.ascontiguousarray(pts).reshape(-1)).to(dev), torch.from_numpy(np.ascontiguousarray(sc).reshape(-1)).to(dev)
    d_o = torch.full((n * 64 + 64,), 0xEE, dtype=torch.uint8, device=dev)
    ctx.mul_var_base_dev(d_p.data_ptr(), d_s.data_ptr(), n, d_o.data_ptr())
    ctx.sync()
    o = d_o.cpu().numpy()
    assert (o[n * 64:] == 0xEE).all()                       # nothing beyond item n - 1 (quads past the batch repeat the last item and store nothing)
    return o[:n * 64].reshape(n, 64)


def _group_points(ctx, n, seed):
    from babyjubjub_rs_amd import workload as w
    return ctx.mul_fixed_base(w.scalars_254(n, offset=seed)).copy()


@pytest.fixture(scope="module")
def k2_ctx():
    """a context whose short calls still run K2 (BJJ_VB_QUAD_MAX=0 is read at bjj_init)"""
    import os
    import babyjubjub_rs_amd as bjj
    old = os.environ.get("BJJ_VB_QUAD_MAX")
    os.environ["BJJ_VB_QUAD_MAX"] = "0"
    try:
        c = bjj.Context(0, 16)
    finally:
        if old is None:
            del os.environ["BJJ_VB_QUAD_MAX"]
        else:
            os.environ["BJJ_VB_QUAD_MAX"] = old
    yield c
    c.close()


def test_every_size_around_the_boundaries_against_k2_and_the_oracle(gpu_ctx, k2_ctx, oracle):
    from babyjubjub_rs_amd import workload as w
    nmax = QUAD_MAX + 70
    pts = _group_points(gpu_ctx, nmax, 5)
    sc = w.scalars_254(nmax, offset=40000)
    sc[:, 31] |= np.arange(nmax, dtype=np.uint8) & 0xC0        # bits 254 / 255 set on three quarters: the raw 256-bit scalar is reduced mod 8l
    want = oracle.mul_var_base(pts[:700], sc[:700])
    for n in (1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 63, 64, 65, 257, 700, 4099, QUAD_MAX - 1, QUAD_MAX, QUAD_MAX + 1, nmax):
        got = _launch(gpu_ctx, pts[:n], sc[:n])
        form = gpu_ctx.info().last_var_base_form
        assert form == (2 if n <= QUAD_MAX else form) and (n <= QUAD_MAX or form in (0, 1)), (n, form)
        m = min(n, 700)
        assert (got[:m] == want[:m]).all(), (n, np.nonzero((got[:m] != want[:m]).any(axis=1))[0][:8])
        if n > 700:
            ref = _launch(k2_ctx, pts[:n], sc[:n])
            assert k2_ctx.info().last_var_base_form in (0, 1)
            assert (got == ref).all(), (n, np.nonzero((got != ref).any(axis=1))[0][:8])
    # ... and through the host pointers (what a single p.mul_scalar(&n) of the crate is): pageable and pinned
    for n in (1, 7, 64, 700):
        assert (gpu_ctx.mul_var_base(pts[:n], sc[:n]) == want[:n]).all(), n
        assert gpu_ctx.info().last_var_base_form == 2
    p_p, p_s, p_o = gpu_ctx.host_empty(64), gpu_ctx.host_empty(32), gpu_ctx.host_empty(64)
    p_p[:], p_s[:] = pts[3], sc[3]
    gpu_ctx._ck(gpu_ctx.lib.bjj_mul_var_base(gpu_ctx.handle, p_p.ctypes.data, p_s.ctypes.data, ctypes.c_size_t(1), p_o.ctypes.data), "bjj_mul_var_base")
    assert (np.asarray(p_o) == want[3]).all()
    for b in (p_p, p_s, p_o):
        gpu_ctx.host_free(b)


def test_special_points_and_scalars(gpu_ctx, oracle):
    """the identity, the point of order two, points of order 4 and 8 (x = 0 / y = 0 and the 8-torsion the reference's cofactor clears), the generator
    and B8, against scalars 0, 1, 2, 7, 8, l - 1, l, l + 1, 8l - 1, 8l, 8l + 1, r - 1, 2^251, 2^254 - 1, 2^255, 2^256 - 1 and all-nibbles-8 (the recoding's carry chain)"""
    g = (995203441582195749578291179787384436505546430278305826713579947235728471134, 5472060717959818805561601436314318772137091100104008585924551046643952123905)
    b8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553, 16950150798460657717958625567821834550301663161624707787222815936182638968203)
    pts_i = [(0, 1), (0, Q - 1), g, b8, (Q - g[0], g[1]), (Q - b8[0], b8[1])]
    # points of order 4: y = 0, x^2 = 1/a (a = 168700); take them from the oracle: 2l * G has order dividing 4, l * G order dividing 8
    lg = oracle.mul_var_base(pack([g[0], g[1]]).reshape(1, 64), pack([L]))
    l2g = oracle.mul_var_base(pack([g[0], g[1]]).reshape(1, 64), pack([2 * L]))
    scal = [0, 1, 2, 7, 8, L - 1, L, L + 1, 8 * L - 1, 8 * L, 8 * L + 1, Q - 1, 1 << 251, (1 << 254) - 1, 1 << 255, (1 << 256) - 1,
            int("88" * 32, 16), int("77" * 32, 16), int("f" * 63, 16), int("08" * 32, 16)]
    base = np.concatenate([pack([c for p in pts_i for c in p]).reshape(len(pts_i), 64), lg, l2g])
    pts = np.repeat(base, len(scal), axis=0)
    sc = np.tile(pack(scal).reshape(len(scal), 32), (base.shape[0], 1))
    want = oracle.mul_var_base(pts, sc)
    got = _launch(gpu_ctx, pts, sc)
    assert gpu_ctx.info().last_var_base_form == 2
    assert (got == want).all(), np.nonzero((got != want).any(axis=1))[0][:8]


@pytest.mark.parametrize("force", [None, "0", "1"], ids=["by_history", "exact_behind", "exact_beside"])
def test_off_curve_points_in_short_calls_go_to_the_exact_kernel(oracle, monkeypatch, force):
    """Point has pub fields and no check (src/lib.rs:134-138): the quad kernel lists an off-curve item for K6 and leaves its slot alone, like K2 --
    K6 behind it (a clean history), beside it behind a scan (after a call that met one), and both forced"""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    if force is not None:
        monkeypatch.setenv("BJJ_VB_SPLIT", force)
    ctx = bjj.Context(0, 16)
    try:
        for rnd, n in enumerate((1, 5, 64, 333, 2000)):
            pts = _group_points(ctx, n, 100 + rnd)
            bad = (w.splitmix64(0x5ca1 + rnd, n, 0) % np.uint64(3)) == 0
            bad[0] = True                                                # every call meets one: the history stays dirty
            pts[bad, 0] ^= 1
            pts[bad & (np.arange(n) % 2 == 0), 33] ^= 0x40
            sc = w.scalars_254(n, offset=7000 + rnd)
            want = oracle.mul_var_base(pts, sc)
            got = _launch(ctx, pts, sc)
            i = ctx.info()
            assert i.last_var_base_form == 2 and (got == want).all(), (n, np.nonzero((got != want).any(axis=1))[0][:8])
            if force is None:
                assert i.last_var_base_split == (0 if rnd == 0 else 1)          # the first call found the history clean
            else:
                assert i.last_var_base_split == int(force)
            assert (ctx.mul_var_base(pts, sc) == want).all()                     # host pointers
    finally:
        ctx.close()


def test_wide_scalars_and_long_calls_keep_k2(gpu_ctx):
    from babyjubjub_rs_amd import workload as w
    pts = _group_points(gpu_ctx, 40, 9)
    wide = np.concatenate([w.scalars_254(40, offset=1), w.scalars_254(40, offset=2)], axis=1)
    gpu_ctx.mul_var_base_wide(pts, wide, 64)
    assert gpu_ctx.info().last_var_base_form in (0, 1)


# ------------------------------------------------------------------------------------------------ Poseidon, six lanes per hash
P5_COOP_MAX = 1 << 14


@pytest.fixture(scope="module")
def lane_ctx():
    """a context whose short Poseidon calls still run one hash per lane (BJJ_P5_COOP_MAX=0 is read at bjj_init)"""
    import os
    import babyjubjub_rs_amd as bjj
    old = os.environ.get("BJJ_P5_COOP_MAX")
    os.environ["BJJ_P5_COOP_MAX"] = "0"
    try:
        c = bjj.Context(0, 16)
    finally:
        if old is None:
            del os.environ["BJJ_P5_COOP_MAX"]
        else:
            os.environ["BJJ_P5_COOP_MAX"] = old
    yield c
    c.close()


def test_poseidon_short_calls_against_the_lane_kernel_and_the_oracle(gpu_ctx, lane_ctx, oracle, golden):
    """POSEIDON.hash(vec![a, b, c, d, e]) (src/lib.rs:400-404) in short calls: every call size around the group / wave / switch-over boundaries, inputs
    that are 0, 1, r - 1, and -- what the reference never feeds it, but the entry point accepts -- >= r"""
    rng = np.random.default_rng(0x9051)
    nmax = P5_COOP_MAX + 9
    h = rng.integers(0, 256, (nmax, 160), dtype=np.uint8)
    h[:, 31::32] &= 0x1f                                      # below 2^253 < r
    edge = [0, 1, 2, Q - 1, Q - 2, Q, Q + 1, (1 << 256) - 1, 1 << 255, (1 << 253)]
    for k, v in enumerate(edge):
        h[k, (k % 5) * 32:(k % 5) * 32 + 32] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
    h[len(edge)] = np.tile(np.frombuffer((Q - 1).to_bytes(32, "little"), np.uint8), 5)
    h[len(edge) + 1] = 0
    want_lane = lane_ctx.poseidon5(h)
    assert lane_ctx.info().last_poseidon_form == 0
    want = oracle.poseidon5(h[:3000])
    assert (want_lane[:3000] == want).all()
    for n in (1, 2, 7, 8, 9, 63, 64, 65, 1000, P5_COOP_MAX - 1, P5_COOP_MAX, P5_COOP_MAX + 1, nmax):
        got = gpu_ctx.poseidon5(h[:n])
        assert gpu_ctx.info().last_poseidon_form == (1 if n <= P5_COOP_MAX else 0), n
        assert got.shape == (n, 32) and (got == want_lane[:n]).all(), (n, np.nonzero((got != want_lane[:n]).any(axis=1))[0][:8])
    # the published known answers (tests/golden/reference_kats.json: poseidon_public) one call per hash -- what a single POSEIDON.hash costs and returns
    for c in golden["reference_kats"]["poseidon_public"]["cases"]:
        one = np.frombuffer(b"".join(int(v, 16).to_bytes(32, "little") if isinstance(v, str) else int(v).to_bytes(32, "little") for v in c["in"]), np.uint8)
        out = c["out"]
        exp = (int(out, 16) if isinstance(out, str) else int(out)).to_bytes(32, "little")
        assert bytes(gpu_ctx.poseidon5(one.reshape(1, 160))[0]) == exp
        assert gpu_ctx.info().last_poseidon_form == 1


# ------------------------------------------------------------------------------------------------ verify, eight lanes per signature
VERIFY_SMALL_MAX = 1 << 13


@pytest.mark.parametrize("window_bits", [16, 23, 13])
def test_verify_short_calls_every_verdict_against_the_lane_kernel_and_the_oracle(oracle, monkeypatch, window_bits):
    """verify(pk, sig, msg) (src/lib.rs:395-412) in short calls: valid signatures, every kind of corruption (a bit in pk, R, s, msg; pk or R off the curve:
    the exact launch's items; msg > Q; s >= l and s >= 2^255; the identity and small-order points as pk or R), every call size around the group / wave /
    switch-over boundaries, three table widths (the fixed-base windows are read one component per lane) -- verdicts equal the oracle's and K4's"""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    ctx = bjj.Context(0, window_bits)
    monkeypatch.setenv("BJJ_VERIFY_SMALL_MAX", "0")
    k4 = bjj.Context(0, window_bits)
    monkeypatch.delenv("BJJ_VERIFY_SMALL_MAX")
    try:
        nmax = VERIFY_SMALL_MAX + 37
        A, R, S, msg = w.make_signatures(oracle.mul_fixed_base, oracle.poseidon5, nmax)
        bad = w.corrupt(A, R, S, msg, nmax)                     # 1 in 64: a bit in one of the four records, half of those push pk / R off the curve
        rng = np.random.default_rng(0x7e51f)
        extra = rng.choice(np.nonzero(~bad)[0], 40, replace=False)
        S[extra[0:5]] = S[extra[0:5]] + 0                        # (valid)
        for k in extra[5:10]:                                    # s + l: B8 has order l, the reference multiplies by the raw integer -> still valid
            v = int.from_bytes(bytes(S[k]), "little") + L
            S[k] = np.frombuffer(v.to_bytes(32, "little"), np.uint8)
        msg[extra[10:13], 31] = 0xff                             # msg > Q -> false
        S[extra[13:16], 31] |= 0x80                              # s >= 2^255: a different integer -> false (unless it happens to be s + k l)
        A[extra[16]] = np.frombuffer((0).to_bytes(32, "little") + (1).to_bytes(32, "little"), np.uint8)            # pk = identity
        R[extra[17]] = np.frombuffer((0).to_bytes(32, "little") + (1).to_bytes(32, "little"), np.uint8)            # R = identity
        A[extra[18]] = np.frombuffer((0).to_bytes(32, "little") + (Q - 1).to_bytes(32, "little"), np.uint8)        # order 2
        R[extra[19]] = np.frombuffer((0).to_bytes(32, "little") + (Q - 1).to_bytes(32, "little"), np.uint8)
        A[extra[20:24], 5] ^= 0x10                               # pk off the curve (almost surely): exact path
        R[extra[24:28], 40] ^= 0x01                              # R.y changed: off the curve
        A[extra[28], :32] = 0xff                                 # coordinates >= r
        R[extra[29], 32:] = 0xff
        S[extra[30]] = 0
        msg[extra[31]] = 0
        want = oracle.verify(A, R, S, msg)
        assert 0 < (want == 0).sum() < nmax // 8 and (want[extra[5:10]] == 1).all()
        ref = k4.eddsa_verify(A, R, S, msg)
        assert k4.info().last_verify_dispatch in (0, 1) and (ref == want).all()
        for n in (1, 2, 7, 8, 9, 63, 64, 65, 1000, VERIFY_SMALL_MAX - 1, VERIFY_SMALL_MAX, VERIFY_SMALL_MAX + 1, nmax):
            got = ctx.eddsa_verify(A[:n], R[:n], S[:n], msg[:n])
            assert ctx.info().last_verify_dispatch == (2 if n <= VERIFY_SMALL_MAX else ctx.info().last_verify_dispatch), n
            assert (n <= VERIFY_SMALL_MAX) == (ctx.info().last_verify_dispatch == 2)
            assert (got == want[:n]).all(), (n, np.nonzero(got != want[:n])[0][:8])
        # one call per signature, the crate's `verify(pk, sig, msg)`: the interesting ones
        for k in list(extra) + list(np.nonzero(bad)[0][:24]):
            assert ctx.eddsa_verify(A[k:k + 1], R[k:k + 1], S[k:k + 1], msg[k:k + 1])[0] == want[k], k
    finally:
        ctx.close(); k4.close()


# ------------------------------------------------------------------------------------------------ sign, eight lanes per signature
SIGN_SMALL_MAX = 1 << 13


def test_sign_short_calls_against_the_lane_kernel_and_the_oracle(oracle, monkeypatch, golden):
    """PrivateKey::sign / Signature::compress (src/lib.rs:308-342, 245-258) in short calls: every call size around the group / wave / switch-over boundaries,
    msg > Q (the reference's Err: ok = 0, zeroed records), both output forms -- byte for byte what the lane kernel and the oracle give; the signatures verify"""
    import babyjubjub_rs_amd as bjj
    ctx = bjj.Context(0, 16)
    monkeypatch.setenv("BJJ_SIGN_SMALL_MAX", "0")
    lane = bjj.Context(0, 16)
    monkeypatch.delenv("BJJ_SIGN_SMALL_MAX")
    try:
        nmax = SIGN_SMALL_MAX + 21
        rng = np.random.default_rng(0x5199)
        keys = rng.integers(0, 256, (nmax, 32), dtype=np.uint8)
        msgs = rng.integers(0, 256, (nmax, 32), dtype=np.uint8)
        msgs[:, 31] &= 0x1f
        msgs[::97, 31] = 0xff                                   # msg > Q
        msgs[5] = 0
        keys[6] = 0
        r0, s0, ok0 = lane.sign(keys, msgs)
        assert lane.info().last_sign_form == 0
        sig0, okc0 = lane.sign_compressed(keys, msgs)
        ro, so, oko = oracle.sign(keys[:600], msgs[:600])
        assert (ok0[:600] == oko).all() and (r0[:600][oko == 1] == ro[oko == 1]).all() and (s0[:600][oko == 1] == so[oko == 1]).all()
        for n in (1, 2, 7, 8, 9, 63, 64, 65, 1000, SIGN_SMALL_MAX - 1, SIGN_SMALL_MAX, SIGN_SMALL_MAX + 1, nmax):
            r, s, ok = ctx.sign(keys[:n], msgs[:n])
            assert ctx.info().last_sign_form == (1 if n <= SIGN_SMALL_MAX else 0), n
            assert (ok == ok0[:n]).all() and (r == r0[:n]).all() and (s == s0[:n]).all(), n
            sig, okc = ctx.sign_compressed(keys[:n], msgs[:n])
            assert ctx.info().last_sign_form == (1 if n <= SIGN_SMALL_MAX else 0)
            assert (okc == okc0[:n]).all() and (sig == sig0[:n]).all(), n
        assert (ok0[::97] == 0).all() and (r0[::97] == 0).all() and (s0[::97] == 0).all()
        good = ok0[:2000] == 1
        pk = ctx.public_keys(keys[:2000])
        r, s, ok = ctx.sign(keys[:2000], msgs[:2000])
        assert (ctx.eddsa_verify(pk[good], r[good], s[good], msgs[:2000][good]) == 1).all()
        # the circomlib vector of the reference's own test (src/lib.rs:692-738) through a one-signature call
        from conftest import ints
        k = golden["reference_kats"]["circomlib_testvector"]
        key = np.frombuffer(bytes.fromhex(k["key"]), np.uint8).reshape(1, 32)
        r1, s1, ok1 = ctx.sign(key, pack([ints(k["msg"])]).reshape(1, 32))
        assert ctx.info().last_sign_form == 1 and ok1[0] == 1
        assert bytes(r1[0]) == ints(k["r_b8"][0]).to_bytes(32, "little") + ints(k["r_b8"][1]).to_bytes(32, "little")
        assert bytes(s1[0]) == ints(k["s"]).to_bytes(32, "little")
    finally:
        ctx.close(); lane.close()


# ------------------------------------------------------------------------------------------------ fixed base / public keys, four lanes per item
FB_QUAD_MAX = 1 << 15


@pytest.mark.parametrize("window_bits", [16, 23, 13, 8])
def test_fixed_base_and_public_keys_short_calls(oracle, monkeypatch, window_bits, golden):
    """B8.mul_scalar(n) / PrivateKey::public (src/lib.rs:149-164, 304-306) and their Point::compress forms in short calls: every call size around the quad / wave /
    switch-over boundaries, scalars 0, 1, l - 1, l, l + 1, 2^256 - 1, four table widths -- byte for byte what K1 and the oracle give"""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    ctx = bjj.Context(0, window_bits)
    monkeypatch.setenv("BJJ_FB_QUAD_MAX", "0")
    k1 = bjj.Context(0, window_bits)
    monkeypatch.delenv("BJJ_FB_QUAD_MAX")
    try:
        nmax = FB_QUAD_MAX + 19
        sc = w.random_u256(w.SEED_SCALARS, nmax, offset=31)
        for k, v in enumerate([0, 1, 2, L - 1, L, L + 1, 8 * L, (1 << 256) - 1, 1 << 255, Q - 1]):
            sc[k] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
        keys = w.random_u256(w.SEED_KEYS, nmax, offset=5)
        want, want_pk = k1.mul_fixed_base(sc), k1.public_keys(keys)
        assert k1.info().last_fixed_base_shape in (0, 1)
        want_c, want_pkc = k1.compress_points(want), k1.compress_points(want_pk)
        assert (want[:300] == oracle.mul_fixed_base(sc[:300])).all() and (want_pk[:300] == oracle.public_keys(keys[:300])).all()
        for n in (1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 1000, FB_QUAD_MAX - 1, FB_QUAD_MAX, FB_QUAD_MAX + 1, nmax):
            form = 2 if n <= FB_QUAD_MAX else None
            for got, exp in ((ctx.mul_fixed_base(sc[:n]), want), (ctx.mul_fixed_base_compressed(sc[:n]), want_c),
                             (ctx.public_keys(keys[:n]), want_pk), (ctx.public_keys_compressed(keys[:n]), want_pkc)):
                f = ctx.info().last_fixed_base_shape
                assert (f == 2) == (form == 2), (n, f)
                assert (got == exp[:n]).all(), (n, np.nonzero((got != exp[:n]).any(axis=1))[0][:8])
        from conftest import ints
        k = golden["reference_kats"]["circomlib_testvector"]
        key = np.frombuffer(bytes.fromhex(k["key"]), np.uint8).reshape(1, 32)
        assert bytes(ctx.public_keys(key)[0]) == ints(k["pk"][0]).to_bytes(32, "little") + ints(k["pk"][1]).to_bytes(32, "little")
    finally:
        ctx.close(); k1.close()


# ------------------------------------------------------------------------------------------------ short calls on pinned memory: no copies at all
def test_short_calls_on_pinned_memory_run_without_copies(gpu_ctx, oracle):
    """PipeSpec::small_direct_max (bjj_hip.hip): a call of at most 256 items whose arrays are all pinned and 16-byte aligned is launch + synchronise -- the short-call
    kernels read the caller's inputs and store into the caller's outputs through the device mappings (bjj_info.last_host_zero_copy == 3).  Results as ever; a pageable
    or misaligned array, or one item more, brings the copies back."""
    import ctypes as C
    from babyjubjub_rs_amd import workload as w
    ctx = gpu_ctx
    n = 200
    A, R, S, M = w.make_signatures(oracle.mul_fixed_base, oracle.poseidon5, 300)
    R[3, 2] ^= 1                                                  # one off-curve R: the exact launch stores its verdict into the host array too
    S[5, 0] ^= 1
    keys = w.random_u256(w.SEED_KEYS, 300, offset=9)
    h5 = np.concatenate([A, R, M], axis=1)

    def pinned(a):
        p = ctx.host_empty(a.size)
        p[:] = np.ascontiguousarray(a).reshape(-1)
        return p

    def call(name, ins, outs, count):
        args = [ctx.handle] + [b.ctypes.data for b in ins] + [C.c_size_t(count)] + [b.ctypes.data for b in outs]
        ctx._ck(getattr(ctx.lib, name)(*args), name)
        return ctx.info().last_host_zero_copy

    cases = [("bjj_mul_fixed_base", [S], [64], lambda k: [oracle.mul_fixed_base(S[:k])]),
             ("bjj_mul_fixed_base_compressed", [S], [32], lambda k: [oracle.compress(oracle.mul_fixed_base(S[:k]))]),
             ("bjj_mul_var_base", [A, S], [64], lambda k: [oracle.mul_var_base(A[:k], S[:k])]),
             ("bjj_poseidon5", [h5], [32], lambda k: [oracle.poseidon5(h5[:k])]),
             ("bjj_eddsa_verify", [A, R, S, M], [1], lambda k: [oracle.verify(A[:k], R[:k], S[:k], M[:k]).reshape(k, 1)]),
             ("bjj_public_keys", [keys], [64], lambda k: [oracle.public_keys(keys[:k])]),
             ("bjj_public_keys_compressed", [keys], [32], lambda k: [oracle.compress(oracle.public_keys(keys[:k]))])]
    for name, ins, widths, expect in cases:
        p_in = [pinned(a) for a in ins]
        p_out = [ctx.host_empty(300 * wd + 64) for wd in widths]
        for count, want_bits in ((1, 3), (n, 3), (256, 3), (257, None)):
            for b in p_out:
                b[:] = 0xEE
            bits = call(name, p_in, p_out, count)
            assert want_bits is None or bits == want_bits, (name, count, bits)
            assert want_bits is not None or not (bits & 1), (name, count, bits)       # one item more: the copy-out is back
            for b, wd, exp in zip(p_out, widths, expect(count)):
                got = np.asarray(b[:count * wd]).reshape(count, wd)
                assert (got == exp).all(), (name, count, np.nonzero((got != exp).any(axis=1))[0][:8])
                assert (np.asarray(b[count * wd:count * wd + 48]) == 0xEE).all(), (name, count)
        # a misaligned pinned input: copies (and the same bytes)
        raw = ctx.host_empty(p_in[0].size + 16)
        mis = raw[8:8 + p_in[0].size]
        mis[:] = p_in[0]
        bits = call(name, [mis] + p_in[1:], p_out, 7)
        assert bits == 0, (name, bits)
        for b, wd, exp in zip(p_out, widths, expect(7)):
            assert (np.asarray(b[:7 * wd]).reshape(7, wd) == exp).all(), name
        for b in p_in + p_out + [raw]:
            ctx.host_free(b)
    # sign: three outputs; the Err row stays zero
    msgs = M.copy()
    msgs[2, 31] = 0xff
    p_k, p_m = pinned(keys), pinned(msgs)
    p_r, p_s, p_ok = ctx.host_empty(300 * 64), ctx.host_empty(300 * 32), ctx.host_empty(304)
    assert call("bjj_sign", [p_k, p_m], [p_r, p_s, p_ok], 9) == 3
    ro, so, oko = oracle.sign(keys[:9], msgs[:9])
    ro[oko == 0] = 0; so[oko == 0] = 0
    assert (np.asarray(p_ok[:9]) == oko).all() and oko[2] == 0
    assert (np.asarray(p_r[:9 * 64]).reshape(9, 64) == ro).all() and (np.asarray(p_s[:9 * 32]).reshape(9, 32) == so).all()
    for b in (p_k, p_m, p_r, p_s, p_ok):
        ctx.host_free(b)


# ------------------------------------------------------------------------------------------------ verify_schnorr, eight lanes per signature
def test_schnorr_verify_short_calls_against_the_lane_kernel_and_the_oracle(oracle, monkeypatch):
    """verify_schnorr (src/lib.rs:364-385) in short calls: signatures from the device signer (s reduced mod 8l for the 32-byte record), corrupted in every record, pk or R
    off the curve (the exact launch's), msg > Q (verdict 2 = the reference's Err) -- every call size around the boundaries against K4's Schnorr kernel and the oracle"""
    import babyjubjub_rs_amd as bjj
    ctx = bjj.Context(0, 16)
    monkeypatch.setenv("BJJ_VERIFY_SMALL_MAX", "0")
    k4 = bjj.Context(0, 16)
    monkeypatch.delenv("BJJ_VERIFY_SMALL_MAX")
    try:
        nmax = VERIFY_SMALL_MAX + 11
        rng = np.random.default_rng(0x5c4)
        keys = rng.integers(0, 256, (nmax, 32), dtype=np.uint8)
        msgs = rng.integers(0, 256, (nmax, 32), dtype=np.uint8)
        msgs[:, 31] &= 0x1f
        nonces = rng.integers(0, 256, (nmax, 128), dtype=np.uint8)
        r, s, ok = ctx.sign_schnorr(keys, msgs, nonces)
        assert ok.all()
        sv = pack([int.from_bytes(s[i].tobytes(), "little") % (8 * L) for i in range(nmax)]).reshape(-1, 32)
        pk = ctx.public_keys(keys).copy()
        r = r.copy()
        sv[::7, 0] ^= 1
        msgs[3::41, 31] = 0xff                                   # msg > Q -> Err (2)
        pk[5::53, 3] ^= 0x04                                     # pk off the curve
        r[9::59, 35] ^= 0x20                                     # R off the curve
        msgs[11::61, 0] ^= 1
        want = k4.schnorr_verify(pk, r, sv, msgs)
        assert k4.info().last_verify_dispatch in (0, 1)
        assert (want[:1500] == oracle.verify_schnorr(pk[:1500], r[:1500], sv[:1500], msgs[:1500])).all()
        assert (want[3::41] == 2).all() and 0 < (want == 0).sum() and (want == 1).sum() > nmax // 2
        for n in (1, 2, 7, 8, 9, 63, 64, 65, 1000, VERIFY_SMALL_MAX - 1, VERIFY_SMALL_MAX, VERIFY_SMALL_MAX + 1, nmax):
            got = ctx.schnorr_verify(pk[:n], r[:n], sv[:n], msgs[:n])
            assert (ctx.info().last_verify_dispatch == 2) == (n <= VERIFY_SMALL_MAX), n
            assert (got == want[:n]).all(), (n, np.nonzero(got != want[:n])[0][:8])
    finally:
        ctx.close(); k4.close()
