"""The product's per-item kernel bodies (babyjubjub-rs_amd/csrc/bjj_device.hpp) executed on
the CPU by the debug harness tests/emul, with limb- and value-bound assertions enabled,
against the oracle.  This is NOT the product path (that is tests/test_gpu_*.py through the
C ABI); it exists so arithmetic-contract violations are caught where there is no GPU."""
import ctypes

from conftest import le32, pack, unpack


def hexint(x):
    return int(x, 16) if isinstance(x, str) else int(x)


def test_emul_fixed_base_golden(emul, golden):
    out = ctypes.create_string_buffer(64)
    for W in (4, 6):
        for c in golden["oracle_vectors"]["fixed_base"][::3]:
            emul.emul_fixed_base(le32(hexint(c["n"])), W, out)
            assert unpack(out.raw, 2)[0] == tuple(hexint(v) for v in c["out"]), (W, c["n"])


def test_emul_var_base_golden(emul, golden):
    out = ctypes.create_string_buffer(64)
    for c in golden["oracle_vectors"]["var_base"]:
        p = pack([tuple(c["p"])]).tobytes()
        emul.emul_var_base(p, le32(hexint(c["n"])), out)
        assert unpack(out.raw, 2)[0] == tuple(hexint(v) for v in c["out"]), c


def test_emul_poseidon_golden(emul, golden):
    out = ctypes.create_string_buffer(32)
    for c in golden["oracle_vectors"]["poseidon5"]:
        emul.emul_poseidon5(pack([tuple(c["in"])]).tobytes(), out)
        assert unpack(out.raw)[0] == hexint(c["out"])
    for c in golden["reference_kats"]["poseidon_public"]["cases"]:
        emul.emul_poseidon5(pack([tuple(c["in"])]).tobytes(), out)
        assert unpack(out.raw)[0] == c["out"]


def test_emul_verify_golden(emul, golden):
    for c in golden["oracle_vectors"]["verify"]:
        got = emul.emul_verify(pack([tuple(c["pk"])]).tobytes(), pack([tuple(c["r_b8"])]).tobytes(),
                               le32(hexint(c["s"])), le32(hexint(c["msg"])), 6)
        assert bool(got) == c["ok"], c["note"]
    v = golden["reference_kats"]["circomlib_testvector"]
    assert emul.emul_verify(pack([tuple(v["pk"])]).tobytes(), pack([tuple(v["r_b8"])]).tobytes(), le32(v["s"]),
                            le32(v["msg"]), 6) == 1
