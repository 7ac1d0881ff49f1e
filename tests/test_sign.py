"""Signer row: Blake-512, PrivateKey::scalar_key / public / sign
(src/lib.rs:226-237, 284-342).  CPU: oracle vs the circomlib KAT (src/lib.rs:689-738: digest, scalar
key, public key, R, S) and the product's bodies in the debug harness; GPU: the C ABI."""
import ctypes

import numpy as np
import pytest

from conftest import pack, unpack, le32

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def hexint(x):
    return int(x, 16) if isinstance(x, str) else int(x)


def test_oracle_circomlib_signer_kat(oracle, pyoracle, golden):
    v = golden["reference_kats"]["circomlib_testvector"]
    key = bytes.fromhex(v["key"])
    assert pyoracle.blake512(key).hex() == v["blake512"]                       # lib.rs:695-696
    d = ctypes.create_string_buffer(64)
    oracle.lib.bjjref_blake512(key, ctypes.c_size_t(32), d)
    assert d.raw.hex() == v["blake512"]
    assert pyoracle.scalar_key(key) == v["scalar_key"]                         # lib.rs:704-707
    assert pyoracle.public(key) == tuple(v["pk"])                              # lib.rs:710-718
    R, S = pyoracle.sign(key, v["msg"])                                        # lib.rs:721-735
    assert R == tuple(v["r_b8"]) and S == v["s"]
    r, s, ok = oracle.sign(np.frombuffer(key, np.uint8), pack([v["msg"]]))
    assert ok[0] == 1 and unpack(r, 2)[0] == tuple(v["r_b8"]) and unpack(s)[0] == v["s"]
    assert unpack(oracle.public_keys(np.frombuffer(key, np.uint8)), 2)[0] == tuple(v["pk"])
    assert pyoracle.blake512(b"").hex().startswith("a8cfbbd73726062df0c6864dda65defe")   # BLAKE spec vectors
    assert pyoracle.blake512(b"\x00" * 144).hex().startswith("313717d608e9cf758dcb1eb0f0c3cf9f")


def _sign_rows(golden):
    sg = golden["oracle_vectors"]["sign"]
    keys = np.frombuffer(b"".join(bytes.fromhex(c["key"]) for c in sg), np.uint8).reshape(-1, 32)
    return sg, keys, pack([c["msg"] for c in sg]).reshape(-1, 32)


def test_oracle_sign_golden(oracle, golden):
    sg, keys, msgs = _sign_rows(golden)
    r, s, ok = oracle.sign(keys, msgs)
    assert [bool(v) for v in ok] == [c["ok"] for c in sg]
    assert (r.reshape(-1) == pack([tuple(c["r_b8"]) for c in sg])).all()
    assert (s.reshape(-1) == pack([c["s"] for c in sg])).all()
    good = [i for i, c in enumerate(sg) if c["ok"]]
    pk = oracle.public_keys(keys[good])
    assert (pk.reshape(-1) == pack([tuple(sg[i]["pk"]) for i in good])).all()
    assert oracle.verify(pk, r[good], s[good], msgs[good]).all()              # sign -> verify, lib.rs:555-572


def test_emul_sign_golden(emul, golden):
    sg, keys, msgs = _sign_rows(golden)
    r, s, sk = ctypes.create_string_buffer(64), ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    for c, k, m in zip(sg, keys, msgs):
        ok = emul.emul_sign(k.tobytes(), m.tobytes(), 6, r, s)
        assert bool(ok) == c["ok"]
        assert unpack(r.raw, 2)[0] == tuple(hexint(v) for v in c["r_b8"]) and unpack(s.raw)[0] == hexint(c["s"])
        emul.emul_scalar_key(k.tobytes(), sk)
        assert unpack(sk.raw)[0] == hexint(c["scalar_key"])


@pytest.mark.gpu
def test_gpu_sign_golden_and_kat(gpu_ctx, golden):
    sg, keys, msgs = _sign_rows(golden)
    r, s, ok = gpu_ctx.sign(keys, msgs)
    assert [bool(v) for v in ok] == [c["ok"] for c in sg]
    assert (r.reshape(-1) == pack([tuple(c["r_b8"]) for c in sg])).all()
    assert (s.reshape(-1) == pack([c["s"] for c in sg])).all()
    assert (gpu_ctx.scalar_keys(keys).reshape(-1) == pack([c["scalar_key"] for c in sg])).all()
    assert (gpu_ctx.public_keys(keys).reshape(-1) == pack([tuple(c["pk"]) for c in sg])).all()
    v = golden["reference_kats"]["circomlib_testvector"]
    import babyjubjub_rs_amd as bjj
    bjj.api._DEFAULT = gpu_ctx
    sk = bjj.PrivateKey.import_(bytes.fromhex(v["key"]))                       # lib.rs:699-703
    assert sk.scalar_key() == v["scalar_key"]
    pk = sk.public()
    assert (pk.x, pk.y) == tuple(v["pk"])
    sig = sk.sign(v["msg"])
    assert (sig.r_b8.x, sig.r_b8.y) == tuple(v["r_b8"]) and sig.s == v["s"]
    assert bjj.verify(pk, sig, v["msg"]) is True                               # lib.rs:736-737
    with pytest.raises(ValueError):
        sk.sign(Q + 1)                                                         # lib.rs:309-311


@pytest.mark.gpu
def test_gpu_sign_random_vs_oracle_and_roundtrip(gpu_ctx, oracle):
    from babyjubjub_rs_amd import workload as w
    n = 2048
    keys = w.random_u256(w.SEED_KEYS ^ 0x99, n)
    msgs = w.random_u256(w.SEED_MSGS ^ 0x99, n, 0, top_bits_cleared=3)
    msgs[5] = 0
    msgs[6] = np.frombuffer(le32(Q), np.uint8)
    msgs[7] = np.frombuffer(le32(Q + 1), np.uint8)
    r, s, ok = gpu_ctx.sign(keys, msgs)
    ro, so, oko = oracle.sign(keys, msgs)
    assert (ok == oko).all() and (r == ro).all() and (s == so).all() and ok[7] == 0 and ok.sum() == n - 1
    pk = gpu_ctx.public_keys(keys)
    assert (pk == oracle.public_keys(keys)).all()
    v = gpu_ctx.eddsa_verify(pk, r, s, msgs)
    assert (v == ok).all()                                                      # every produced signature verifies
    # full size: sign 2^19 on the GPU, verify all, sample against the oracle
    m = 1 << 19
    keys = w.random_u256(w.SEED_KEYS, m)
    msgs = w.random_u256(w.SEED_MSGS, m, 0, top_bits_cleared=3)
    r, s, ok = gpu_ctx.sign(keys, msgs)
    assert ok.all()
    assert gpu_ctx.eddsa_verify(gpu_ctx.public_keys(keys), r, s, msgs).all()
    idx = np.arange(0, m, 2053)
    ro, so, _ = oracle.sign(keys[idx], msgs[idx])
    assert (r[idx] == ro).all() and (s[idx] == so).all()


@pytest.mark.gpu
def test_gpu_constant_time_signer_is_bit_identical(golden, oracle):
    """bjj_set_signer_constant_time: public_keys / sign / sign_schnorr through the scanning policy over the small 4-bit table
    (no secret-dependent address or branch) give exactly the bytes of the indexed form and of the oracle -- golden vectors, the
    circomlib KAT (src/lib.rs:689-738), random batches with partially filled waves, msg edge cases; the flag is reported by
    bjj_get_info and can be switched off again."""
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    ctx = bjj.Context(0, 16)
    try:
        assert ctx.info().signer_constant_time == 0
        sg, keys, msgs = _sign_rows(golden)
        plain = (ctx.sign(keys, msgs), ctx.public_keys(keys))
        ctx.set_signer_constant_time(True)
        assert ctx.info().signer_constant_time == 1
        r, s, ok = ctx.sign(keys, msgs)
        assert (r == plain[0][0]).all() and (s == plain[0][1]).all() and (ok == plain[0][2]).all()
        assert [bool(v) for v in ok] == [c["ok"] for c in sg]
        assert (r.reshape(-1) == pack([tuple(c["r_b8"]) for c in sg])).all() and (s.reshape(-1) == pack([c["s"] for c in sg])).all()
        assert (ctx.public_keys(keys) == plain[1]).all()
        assert (ctx.public_keys(keys).reshape(-1) == pack([tuple(c["pk"]) for c in sg])).all()
        v = golden["reference_kats"]["circomlib_testvector"]
        key = np.frombuffer(bytes.fromhex(v["key"]), np.uint8).reshape(1, 32)
        r1, s1, ok1 = ctx.sign(key, pack([v["msg"]]).reshape(1, 32))
        assert ok1[0] == 1 and unpack(r1[0], 2)[0] == tuple(v["r_b8"]) and unpack(s1[0])[0] == v["s"]
        assert unpack(ctx.public_keys(key)[0], 2)[0] == tuple(v["pk"])
        for n in (1, 63, 65, 1000, 5000):
            keys = w.random_u256(w.SEED_KEYS ^ 0x5151, n, offset=n)
            msgs = w.random_u256(w.SEED_MSGS ^ 0x5151, n, offset=n, top_bits_cleared=3)
            if n > 8:
                msgs[5] = 0
                msgs[6] = np.frombuffer(le32(Q), np.uint8)
                msgs[7] = np.frombuffer(le32(Q + 1), np.uint8)
            r, s, ok = ctx.sign(keys, msgs)
            ro, so, oko = oracle.sign(keys, msgs)
            assert (ok == oko).all() and (r == ro).all() and (s == so).all(), n
            pk = ctx.public_keys(keys)
            assert (pk == oracle.public_keys(keys)).all(), n
            assert (ctx.eddsa_verify(pk, r, s, msgs) == ok).all()
        # Schnorr: unreduced s and R against the indexed form
        rng = np.random.default_rng(99)
        n = 777
        keys = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        msgs = rng.integers(0, 256, (n, 32), dtype=np.uint8); msgs[:, 31] &= 0x1f
        nonces = rng.integers(0, 256, (n, 128), dtype=np.uint8)
        ct = ctx.sign_schnorr(keys, msgs, nonces)
        ctx.set_signer_constant_time(False)
        assert ctx.info().signer_constant_time == 0
        ix = ctx.sign_schnorr(keys, msgs, nonces)
        assert all((a == b).all() for a, b in zip(ct, ix)) and ct[2].all()
    finally:
        ctx.close()
