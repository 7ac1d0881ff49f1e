"""Schnorr variant: schnorr_hash / verify_schnorr, src/lib.rs:364-385.  The reference
only has a randomised round trip for it (src/lib.rs:678-686); vectors here come from the oracle with the
nonce drawn from a seeded stream."""
import numpy as np
import pytest

from conftest import pack, unpack

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
L = 2736030358979909402780800718157159386076813972158567259200215660948447373041


def _rows(golden):
    sc = golden["oracle_vectors"]["schnorr"]
    return sc, pack([tuple(c["pk"]) for c in sc]), pack([tuple(c["r"]) for c in sc]), pack([c["s"] for c in sc]), \
        pack([c["msg"] for c in sc])


def test_oracle_schnorr_golden(oracle, pyoracle, golden):
    sc, pk, r, s, m = _rows(golden)
    assert list(oracle.verify_schnorr(pk, r, s, m)) == [c["ok"] for c in sc]
    assert any(c["s_unreduced_bits"] > 1000 for c in sc)      # the crate's s = k + x*h is ~1024+ bits wide


def test_emul_schnorr_golden(emul, golden):
    sc, pk, r, s, m = _rows(golden)
    for i, c in enumerate(sc):
        got = emul.emul_verify_schnorr(pk[64 * i:64 * i + 64].tobytes(), r[64 * i:64 * i + 64].tobytes(),
                                       s[32 * i:32 * i + 32].tobytes(), m[32 * i:32 * i + 32].tobytes(), 6)
        assert got == c["ok"], c["note"]


@pytest.mark.gpu
def test_gpu_schnorr_golden_and_random(gpu_ctx, oracle, golden):
    sc, pk, r, s, m = _rows(golden)
    assert list(gpu_ctx.schnorr_verify(pk, r, s, m)) == [c["ok"] for c in sc]
    # seeded random batch: valid signatures built from GPU kernels + host integers, some corrupted
    from babyjubjub_rs_amd import workload as w
    n = 1024
    keys = w.random_u256(w.SEED_KEYS ^ 0x33, n)
    msgs = w.random_u256(w.SEED_MSGS ^ 0x33, n, 0, top_bits_cleared=3)
    x = w.to_ints(gpu_ctx.scalar_keys(keys))
    pkp = gpu_ctx.public_keys(keys)
    k = [v % (8 * L) for v in w.to_ints(w.random_u256(w.SEED_NONCES ^ 0x33, n))]
    rp = gpu_ctx.mul_fixed_base(w.from_ints(k))
    h = w.to_ints(gpu_ctx.poseidon5(np.concatenate([pkp, rp, msgs], axis=1)))     # schnorr_hash order, lib.rs:369
    sv = w.from_ints([(k[i] + x[i] * h[i]) % (8 * L) for i in range(n)])
    sv[::9, 1] ^= 4
    msgs[5] = np.frombuffer((Q + 7).to_bytes(32, "little"), np.uint8)
    pkp[11, 40] ^= 1                                                                # off-curve pk
    got = gpu_ctx.schnorr_verify(pkp, rp, sv, msgs)
    want = oracle.verify_schnorr(pkp, rp, sv, msgs)
    assert (got == want).all() and got[5] == 2 and got[0] == 0 and got[1] == 1


@pytest.mark.gpu
def test_gpu_reference_api_schnorr(gpu_ctx, golden):
    """src/lib.rs:678-686 re-stated (nonce fixed instead of thread_rng)"""
    import babyjubjub_rs_amd as bjj
    from conftest import ints
    bjj.api._DEFAULT = gpu_ctx
    c = golden["gpu_expected"]["sign_schnorr_api"][0]                              # key 00..1f, nonce 2^1023 + 12345
    sk = bjj.PrivateKey(bytes.fromhex(c["key"]))
    pk = sk.public()
    msg = ints(c["msg"])
    r, s = ints(c["r"]), ints(c["s"])                                              # fixture: tests/golden/make_gpu_expected.py
    assert s.bit_length() > 1000
    assert bjj.verify_schnorr(pk, msg, bjj.Point(*r), s) is True                  # s reduced mod 8l by the mirror
    assert bjj.verify_schnorr(pk, msg + 1, bjj.Point(*r), s) is False
    with pytest.raises(ValueError):
        bjj.verify_schnorr(pk, Q + 1, bjj.Point(*r), s)


# ---- sign_schnorr (src/lib.rs:344-361) with caller-supplied nonces -------------------------------------
ORDER8 = 8 * 2736030358979909402780800718157159386076813972158567259200215660948447373041


def _schnorr_sign_cases():
    """(key, msg, nonce): edge nonces (0, 1, l, 2^1024 - 1, top bit only) and msgs (0, Q, Q + 1 = Err) + seeded random"""
    rng = np.random.default_rng(0x5C4E)
    rb = lambda n: int.from_bytes(rng.bytes(n), "little")  # noqa: E731
    L = ORDER8 // 8
    cases = [(bytes(range(32)), 0, 0), (bytes(range(32)), 1, 1), (b"\xff" * 32, Q, L), (b"\x00" * 32, Q + 1, 5),
             (bytes(rng.bytes(32)), rb(31), (1 << 1024) - 1), (bytes(rng.bytes(32)), Q - 1, 1 << 1023),
             (bytes(rng.bytes(32)), rb(31), L - 1), (bytes(rng.bytes(32)), rb(31), (1 << 261) - 1)]
    cases += [(bytes(rng.bytes(32)), rb(32) % Q, rb(128)) for _ in range(10)]
    return cases


def _schnorr_expect(pyoracle, cases):
    out = []
    for key, m, k in cases:
        res = pyoracle.sign_schnorr_with_nonce(key, m, k)
        out.append(None if res is None else (tuple(res[0]), res[1]))
    return out


def _schnorr_expect_golden(golden, cases):
    """the same expectations from tests/golden/gpu_expected.json (what the GPU box gets); the inputs stored there must be
    the seeded case list above"""
    from conftest import ints
    rows = golden["gpu_expected"]["sign_schnorr_cases"]
    assert [(bytes.fromhex(r["key"]), ints(r["msg"]), ints(r["nonce"])) for r in rows] == cases
    return [None if r["r"] is None else (ints(r["r"]), ints(r["s"])) for r in rows]


def test_schnorr_fixture_is_what_the_python_oracle_computes(pyoracle, golden):
    cases = _schnorr_sign_cases()
    assert _schnorr_expect_golden(golden, cases) == _schnorr_expect(pyoracle, cases)


def test_oracle_sign_schnorr_roundtrip(pyoracle):
    """the reference's own (and only) Schnorr test is sign -> verify == true, src/lib.rs:678-686"""
    for key, m, k in _schnorr_sign_cases()[:6]:
        res = pyoracle.sign_schnorr_with_nonce(key, m, k)
        if m > Q:
            assert res is None
            continue
        r, s = res
        assert s == k + pyoracle.scalar_key(key) * pyoracle.schnorr_hash(pyoracle.public(key), m, r)
        assert pyoracle.verify_schnorr(pyoracle.public(key), m, r, s) is True


def test_emul_sign_schnorr(emul, pyoracle):
    import ctypes
    cases = _schnorr_sign_cases()[:10]
    r, s = ctypes.create_string_buffer(64), ctypes.create_string_buffer(160)
    for (key, m, k), want in zip(cases, _schnorr_expect(pyoracle, cases)):
        ok = emul.emul_sign_schnorr(key, (m % (1 << 256)).to_bytes(32, "little"), k.to_bytes(128, "little"), 6, r, s)
        assert bool(ok) == (want is not None)
        if want is not None:
            assert unpack(r.raw, 2)[0] == want[0] and int.from_bytes(s.raw, "little") == want[1]
        else:
            assert r.raw == b"\x00" * 64 and s.raw == b"\x00" * 160


@pytest.mark.gpu
def test_gpu_sign_schnorr_vs_oracle_and_roundtrip(gpu_ctx, golden, oracle):
    cases = _schnorr_sign_cases()
    want = _schnorr_expect_golden(golden, cases)
    keys = np.frombuffer(b"".join(c[0] for c in cases), np.uint8).reshape(-1, 32)
    msgs = pack([c[1] for c in cases]).reshape(-1, 32)
    nonces = np.frombuffer(b"".join(c[2].to_bytes(128, "little") for c in cases), np.uint8).reshape(-1, 128)
    r, s, ok = gpu_ctx.sign_schnorr(keys, msgs, nonces)
    assert [bool(v) for v in ok] == [w is not None for w in want]
    for i, w in enumerate(want):
        if w is None:
            assert not r[i].any() and not s[i].any()
        else:
            assert unpack(r[i], 2)[0] == w[0] and int.from_bytes(s[i].tobytes(), "little") == w[1]
    # sign -> verify on the GPU (src/lib.rs:678-686), s reduced mod 8l for the 32-byte record
    good = [i for i, w in enumerate(want) if w is not None]
    sv = pack([want[i][1] % ORDER8 for i in good]).reshape(-1, 32)
    pk = gpu_ctx.public_keys(keys[good])
    assert (gpu_ctx.schnorr_verify(pk, r[good], sv, msgs[good]) == 1).all()
    assert (oracle.verify_schnorr(pk, r[good], sv, msgs[good]) == 1).all()
    # a 3 000-item seeded batch through the pipelined host API: every signature must verify
    rng = np.random.default_rng(77)
    n = 3000
    keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8); msgs[:, 31] &= 0x1f
    nonces = rng.integers(0, 256, (n, 128), dtype=np.uint8)
    r, s, ok = gpu_ctx.sign_schnorr(keys, msgs, nonces)
    assert ok.all()
    sv = pack([int.from_bytes(s[i].tobytes(), "little") % ORDER8 for i in range(n)]).reshape(-1, 32)
    pk = gpu_ctx.public_keys(keys)
    assert (gpu_ctx.schnorr_verify(pk, r, sv, msgs) == 1).all()
    sv[::7, 0] ^= 1
    got = gpu_ctx.schnorr_verify(pk, r, sv, msgs)
    assert (got[::7] == 0).all() and (np.delete(got, np.arange(0, n, 7)) == 1).all()


@pytest.mark.gpu
def test_gpu_reference_api_sign_schnorr(gpu_ctx, golden):
    import babyjubjub_rs_amd as bjj
    from conftest import ints
    sk = bjj.new_key()                                                           # lib.rs:387-393
    assert len(sk.key) == 32
    msg = 123456789012345678901234567890
    r, s = sk.sign_schnorr(msg)                                                  # random nonce, lib.rs:347-348
    assert bjj.verify_schnorr(sk.public(), msg, r, s) is True                    # lib.rs:678-686
    c = golden["gpu_expected"]["sign_schnorr_api"][1]                            # key 00..1f, nonce 2^1000 + 99: a fixture
    sk = bjj.PrivateKey.import_(bytes.fromhex(c["key"]))
    assert ints(c["msg"]) == msg
    r2, s2 = sk.sign_schnorr(msg, k=ints(c["nonce"]))
    assert (r2.x, r2.y) == ints(c["r"]) and s2 == ints(c["s"])
    with pytest.raises(ValueError):
        sk.sign_schnorr(Q + 1)
