"""Codec row: Point::compress / decompress_point / decompress_signature and
verify on compressed inputs.  CPU part: the oracle against the reference's KATs
(src/lib.rs:575-632) and the product's bodies in the CPU debug harness; GPU part: the C ABI."""
import ctypes

import numpy as np
import pytest

from conftest import pack, unpack

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = (17777552123799933955779906779655732241715742912184938656739573121738514868268,
     2626589144620713026669568689430873010625803728049924121243784502389097019475)


def hexint(x):
    return int(x, 16) if isinstance(x, str) else int(x)


def _comp_rows(cases):
    return np.frombuffer(b"".join(bytes.fromhex(c["in"]) for c in cases), np.uint8).reshape(-1, 32)


# ---------------------------------------------------------------- oracle vs reference KATs (CPU)
def test_oracle_codec_reference_kats(oracle, pyoracle, golden):
    k = golden["reference_kats"]
    c = k["point_compress"]
    assert pyoracle.compress(tuple(c["p"])).hex() == c["hex"]                     # lib.rs:586-590
    assert bytes(oracle.compress(pack([tuple(c["p"])]))[0]).hex() == c["hex"]
    pts, ok = oracle.decompress(np.frombuffer(bytes.fromhex(c["hex"]), np.uint8))  # lib.rs:591-593
    assert ok[0] == 1 and unpack(pts, 2)[0] == tuple(c["p"])
    for case in k["point_decompress"]["cases"]:                                   # lib.rs:597-632
        yb = bytes.fromhex(case["y_bytes"])
        want_x = int.from_bytes(bytes.fromhex(case["x_le_bytes"]), "little")
        assert pyoracle.decompress_point(yb)[0] == want_x
        pts, ok = oracle.decompress(np.frombuffer(yb, np.uint8))
        assert ok[0] == 1 and unpack(pts, 2)[0][0] == want_x


def test_oracle_codec_golden(oracle, golden):
    cases = golden["oracle_vectors"]["decompress"]
    pts, ok = oracle.decompress(_comp_rows(cases))
    assert [bool(v) for v in ok] == [c["ok"] for c in cases]
    assert (pts.reshape(-1) == pack([tuple(c["out"]) for c in cases])).all()
    good = [i for i, c in enumerate(cases) if c["ok"]]
    assert (oracle.compress(pts[good]) == _comp_rows(cases)[good]).all()           # round trip, lib.rs:635-654
    vc = golden["oracle_vectors"]["verify_compressed"]
    got = oracle.verify_compressed(np.frombuffer(b"".join(bytes.fromhex(c["pk"]) for c in vc), np.uint8),
                                   np.frombuffer(b"".join(bytes.fromhex(c["sig"]) for c in vc), np.uint8),
                                   pack([c["msg"] for c in vc]))
    assert list(got) == [c["ok"] for c in vc]


def test_emul_codec_golden(emul, golden):
    out = ctypes.create_string_buffer(64)
    comp = ctypes.create_string_buffer(32)
    for c in golden["oracle_vectors"]["decompress"]:
        raw = bytes.fromhex(c["in"])
        ok = emul.emul_decompress(raw, out)
        assert bool(ok) == c["ok"], c["note"]
        assert unpack(out.raw, 2)[0] == tuple(hexint(v) for v in c["out"]), c["note"]
        if c["ok"]:
            emul.emul_compress(out.raw, comp)
            assert comp.raw == raw


# ---------------------------------------------------------------- GPU through the C ABI
@pytest.mark.gpu
def test_gpu_codec_golden_and_kats(gpu_ctx, golden):
    cases = golden["oracle_vectors"]["decompress"]
    pts, ok = gpu_ctx.decompress_points(_comp_rows(cases))
    assert [bool(v) for v in ok] == [c["ok"] for c in cases]
    assert (pts.reshape(-1) == pack([tuple(c["out"]) for c in cases])).all()
    good = [i for i, c in enumerate(cases) if c["ok"]]
    assert (gpu_ctx.compress_points(pts[good]) == _comp_rows(cases)[good]).all()
    k = golden["reference_kats"]
    assert bytes(gpu_ctx.compress_points(pack([tuple(k["point_compress"]["p"])]))[0]).hex() == k["point_compress"]["hex"]
    for case in k["point_decompress"]["cases"]:
        p, o = gpu_ctx.decompress_points(np.frombuffer(bytes.fromhex(case["y_bytes"]), np.uint8))
        assert o[0] == 1 and unpack(p, 2)[0][0] == int.from_bytes(bytes.fromhex(case["x_le_bytes"]), "little")
    vc = golden["oracle_vectors"]["verify_compressed"]
    got = gpu_ctx.eddsa_verify_compressed(np.frombuffer(b"".join(bytes.fromhex(c["pk"]) for c in vc), np.uint8),
                                          np.frombuffer(b"".join(bytes.fromhex(c["sig"]) for c in vc), np.uint8),
                                          pack([c["msg"] for c in vc]))
    assert list(got) == [c["ok"] for c in vc]


@pytest.mark.gpu
def test_gpu_codec_random_vs_oracle(gpu_ctx, oracle):
    from babyjubjub_rs_amd import workload as w
    n = 3000
    raw = w.random_u256(w.SEED_POINTS ^ 0x77, n)          # arbitrary 32-byte strings: ~45 % decompress
    raw[::3, 31] &= 0x3f                                   # make more of them < Q
    gp, gok = gpu_ctx.decompress_points(raw)
    op, ook = oracle.decompress(raw)
    assert (gok == ook).all() and (gp == op).all()
    assert 0.2 < gok.mean() < 0.8
    pts = oracle.mul_fixed_base(w.random_u256(w.SEED_KEYS ^ 5, n))
    assert (gpu_ctx.compress_points(pts) == oracle.compress(pts)).all()


@pytest.mark.gpu
def test_gpu_verify_compressed_and_roundtrip_1m(gpu_ctx, oracle):
    """signature compress -> decompress -> verify as src/lib.rs:657-675, at scale: 2^18 signatures made
    by the GPU kernels, 1/64 corrupted; and compress(decompress(.)) == id over 2^20 points."""
    from babyjubjub_rs_amd.workload import make_signatures, corrupt
    n = 1 << 18
    A, R, S, msg = make_signatures(gpu_ctx.mul_fixed_base, gpu_ctx.poseidon5, n)
    pk_c = gpu_ctx.compress_points(A)
    sig_c = np.concatenate([gpu_ctx.compress_points(R), S], axis=1)
    bad = corrupt(A, R, S, msg, n)   # corrupts the affine copies; recompress what changed in S / msg only
    sig_c[:, 32:] = S
    want = gpu_ctx.eddsa_verify(gpu_ctx.decompress_points(pk_c)[0], gpu_ctx.decompress_points(sig_c[:, :32].copy())[0], S, msg)
    got = gpu_ctx.eddsa_verify_compressed(pk_c, sig_c, msg)
    assert (got == want).all()
    idx = np.arange(0, n, 1009)
    assert (got[idx] == oracle.verify_compressed(pk_c[idx], sig_c[idx], msg[idx])).all()
    assert got.sum() > n * 0.95 and (got == 0).sum() > 100
    m = 1 << 20
    from babyjubjub_rs_amd import workload as w
    pts = gpu_ctx.mul_fixed_base(w.scalars_254(m))
    comp = gpu_ctx.compress_points(pts)
    back, ok = gpu_ctx.decompress_points(comp)
    assert ok.all() and (back == pts).all()


@pytest.mark.gpu
def test_gpu_reference_api_compress_decompress(gpu_ctx):
    """src/lib.rs:575-594 re-stated against the mirror"""
    import babyjubjub_rs_amd as bjj
    bjj.api._DEFAULT = gpu_ctx
    p = bjj.Point(*P)
    p_comp = p.compress()
    assert p_comp.hex() == "53b81ed5bffe9545b54016234682e7b2f699bd42a5e9eae27ff4051bc698ce85"
    p2 = bjj.decompress_point(p_comp)
    assert p.x == p2.x and p.y == p2.y
    with pytest.raises(ValueError):
        bjj.decompress_point((Q).to_bytes(32, "little"))   # "y outside the Finite Field over R", lib.rs:201-203
