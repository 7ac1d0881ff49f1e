"""The oracle (Python-int and C) against every known-answer vector the reference's own
tests hold for the hot path (tests/golden/reference_kats.json <- src/lib.rs:421-552, 689-738),
and the C oracle against the Python oracle on seeded inputs.  CPU only."""
import random

import numpy as np

from conftest import pack, unpack, le32


def test_python_oracle_reference_kats(pyoracle, golden):
    o, k = pyoracle, golden["reference_kats"]
    for name in ("add_same_point", "add_different_points"):
        c = k[name]
        p, q = tuple(c["p"]), tuple(c["q"])
        assert o.proj_affine(o.proj_add(p + (1,), q + (1,))) == tuple(c["sum"]), name
    m = k["mul_scalar"]
    P = tuple(m["p"])
    for c in m["cases"]:
        assert o.mul_scalar(P, c["n"]) == tuple(c["out"])
    # x3 cross-check exactly as src/lib.rs:513-516
    pp = P + (1,)
    assert o.proj_affine(o.proj_add(o.proj_add(pp, pp), pp)) == o.mul_scalar(P, 3)
    assert o.compress(tuple(k["point_compress"]["p"])).hex() == k["point_compress"]["hex"]
    v = k["circomlib_testvector"]
    assert o.mul_scalar(o.B8, v["scalar_key"]) == tuple(v["pk"])  # PrivateKey::public, lib.rs:304-306
    assert o.verify(tuple(v["pk"]), tuple(v["r_b8"]), v["s"], v["msg"]) is True
    assert o.verify(tuple(v["pk"]), tuple(v["r_b8"]), v["s"] + 1, v["msg"]) is False
    for c in k["poseidon_public"]["cases"]:
        assert o.poseidon(c["in"]) == c["out"]


def test_c_oracle_reference_kats(oracle, golden):
    k = golden["reference_kats"]
    for name in ("add_same_point", "add_different_points"):
        c = k[name]
        got = unpack(oracle.point_add(pack([tuple(c["p"])]), pack([tuple(c["q"])])), 2)[0]
        assert got == tuple(c["sum"]), name
    m = k["mul_scalar"]
    for c in m["cases"]:
        got = unpack(oracle.mul_var_base(pack([tuple(m["p"])]), pack([c["n"]])), 2)[0]
        assert got == tuple(c["out"])
    v = k["circomlib_testvector"]
    assert unpack(oracle.mul_fixed_base(pack([v["scalar_key"]])), 2)[0] == tuple(v["pk"])
    ok = oracle.verify(pack([tuple(v["pk"])]), pack([tuple(v["r_b8"])]), pack([v["s"]]), pack([v["msg"]]))
    assert ok[0] == 1
    bad = oracle.verify(pack([tuple(v["pk"])]), pack([tuple(v["r_b8"])]), pack([v["s"] ^ 2]), pack([v["msg"]]))
    assert bad[0] == 0
    for c in k["poseidon_public"]["cases"]:
        assert unpack(oracle.poseidon5(pack([tuple(c["in"])])))[0] == c["out"]
    b = k["bench_inputs"]
    for n in b["scalars"]:  # benches/bench_babyjubjub.rs:31-38
        got = unpack(oracle.mul_var_base(pack([tuple(b["p"])]), pack([n])), 2)[0]
        assert got == tuple(golden_mul(b["p"], n))


def golden_mul(p, n):
    import bjj_oracle as o
    return o.mul_scalar(tuple(p), n)


def test_c_oracle_matches_golden_vectors(oracle, golden):
    v = golden["oracle_vectors"]
    fb = v["fixed_base"]
    got = oracle.mul_fixed_base(pack([c["n"] for c in fb]))
    assert (got.reshape(-1) == pack([tuple(c["out"]) for c in fb])).all()
    vb = v["var_base"]
    got = oracle.mul_var_base(pack([tuple(c["p"]) for c in vb]), pack([c["n"] for c in vb]))
    assert (got.reshape(-1) == pack([tuple(c["out"]) for c in vb])).all()
    ps = v["poseidon5"]
    got = oracle.poseidon5(pack([tuple(c["in"]) for c in ps]))
    assert (got.reshape(-1) == pack([c["out"] for c in ps])).all()
    ad = v["point_add"]
    got = oracle.point_add(pack([tuple(c["p"]) for c in ad]), pack([tuple(c["q"]) for c in ad]))
    assert (got.reshape(-1) == pack([tuple(c["out"]) for c in ad])).all()
    ve = v["verify"]
    got = oracle.verify(pack([tuple(c["pk"]) for c in ve]), pack([tuple(c["r_b8"]) for c in ve]),
                        pack([c["s"] for c in ve]), pack([c["msg"] for c in ve]))
    assert [bool(x) for x in got] == [c["ok"] for c in ve]


def test_c_oracle_field_ops_vs_python(oracle, pyoracle):
    import ctypes
    o, L = pyoracle, oracle.lib
    rnd = random.Random(5)
    out = ctypes.create_string_buffer(32)
    vals = [0, 1, o.Q - 1, o.Q - 2, 2, (1 << 253)] + [rnd.randrange(o.Q) for _ in range(200)]
    for _ in range(500):
        a, b = rnd.choice(vals), rnd.choice(vals)
        L.bjjref_fr_mul(le32(a), le32(b), out)
        assert int.from_bytes(out.raw, "little") == a * b % o.Q
        L.bjjref_fr_add(le32(a), le32(b), out)
        assert int.from_bytes(out.raw, "little") == (a + b) % o.Q
        L.bjjref_fr_sub(le32(a), le32(b), out)
        assert int.from_bytes(out.raw, "little") == (a - b) % o.Q
        ok = L.bjjref_fr_inverse(le32(a), out)
        assert ok == (1 if a else 0)
        if a:
            assert int.from_bytes(out.raw, "little") == pow(a, o.Q - 2, o.Q)


def test_c_oracle_threads_agree(oracle):
    rng = np.random.default_rng(3)
    sc = rng.integers(0, 256, 64 * 32, dtype=np.uint8)
    t = oracle.threads
    oracle.threads = 1
    a = oracle.mul_fixed_base(sc)
    oracle.threads = max(2, t)
    b = oracle.mul_fixed_base(sc)
    oracle.threads = t
    assert (a == b).all()
