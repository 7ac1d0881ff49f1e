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
