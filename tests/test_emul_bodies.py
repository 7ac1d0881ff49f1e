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


def test_emul_fixed_table_chain_builder(emul):
    """The chain builder (what bjj_k_build_fixed_table runs per thread) reproduces the independent per-entry
    ladder for every entry, for chain lengths that do and do not divide the window, and the induction check the
    GPU runs (bjj_check_table) accepts the table and fires on a single flipped bit."""
    emul.emul_table_selfcheck.restype = ctypes.c_ulonglong
    emul.emul_table_selfcheck.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_longlong]
    for W, chain in ((4, 4), (5, 7), (6, 64)):
        r = emul.emul_table_selfcheck(W, chain, -1)
        assert r == 0, (W, chain, r >> 32, r & 0xFFFFFFFF)
    r = emul.emul_table_selfcheck(5, 7, 40)
    assert (r >> 32) == 0 and (r & 0xFFFFFFFF) > 0


def test_emul_fixed_base_signed_digit_edges(emul, pyoracle):
    """Scalars that stress the signed recoding: digits exactly 2^(W-1), 2^(W-1)+1, all-ones runs (carry chains),
    l-1, l, l+1, 2^256-1 (reduced mod l first)."""
    o = pyoracle
    out = ctypes.create_string_buffer(64)
    l = o.SUBORDER
    for W in (4, 7):
        half = 1 << (W - 1)
        cases = [0, 1, half, half + 1, (1 << W) - 1, 1 << W, l - 1, l, l + 1, (1 << 256) - 1, (1 << 251) - 1,
                 sum(half << (W * j) for j in range(252 // W)) % l, sum((half + 1) << (W * j) for j in range(252 // W)),
                 sum(((1 << W) - 1) << (W * j) for j in range(0, 252 // W, 2))]
        for n in cases:
            emul.emul_fixed_base(le32(n % (1 << 256)), W, out)
            assert unpack(out.raw, 2)[0] == o.mul_scalar(o.B8, n % (1 << 256)), (W, hex(n))


def test_emul_joint_loop_at_the_minimal_window_count(emul, oracle, golden):
    """verify's joint double-and-add, called directly with exactly the window count an item needs -- ceil((b + 2) / 4) for a
    b-bit scalar (signed 4-bit digits: top nibble + carry must stay below 8) -- on the patterns that stress the recoding:
    all-ones scalars (a carry through every window), 0x80..0, 0x88..8 runs, both scalars at the edge and only one of them"""
    import numpy as np
    from babyjubjub_rs_amd import workload as w
    pts = oracle.mul_fixed_base(w.random_u256(0x70F, 2))             # two points of the group
    p1, p2 = pts[0].tobytes(), pts[1].tobytes()
    out = ctypes.create_string_buffer(64)
    cases = []
    for b in (4, 5, 8, 122, 123, 124, 125, 126, 127, 128, 130, 131, 132, 134, 250, 251):
        top = 1 << (b - 1)
        for val in (top, (1 << b) - 1, top | ((1 << (b - 1)) - 1) // 15 * 8, top + 0x88888888, (1 << b) - 0x77777777):
            if val.bit_length() == b:                                    # b-bit values only (small widths drop some patterns)
                cases.append((val, b))
    for (u, b) in cases:
        nwin = (b + 5) >> 2
        for v in (u, 1, (1 << (4 * nwin - 2)) - 1, 0):
            emul.emul_joint_mul(p1, p2, le32(u), le32(v), nwin, out)
            want = oracle.point_add(oracle.mul_var_base(pts[0:1], pack([u]).reshape(1, 32)),
                                    oracle.mul_var_base(pts[1:2], pack([v]).reshape(1, 32)))[0]
            assert out.raw == want.tobytes(), (hex(u), hex(v), nwin)


def test_emul_digit_stream_equals_the_indexed_digits(emul):
    """round 4: the kernels take the signed W-bit digits of the (reduced) scalar from a shift register instead of indexing the
    words by a run-time window number; both definitions agree on every window for every width 4..28"""
    import numpy as np
    L = 2736030358979909402780800718157159386076813972158567259200215660948447373041
    rng = np.random.default_rng(0xD161)
    edge = [0, 1, L - 1, (1 << 251) - 1, (1 << 250) + 1, int("8" * 62, 16) % L, int("7" * 62, 16) % L, (1 << 200) - 1]
    for W in range(4, 29):
        half = 1 << (W - 1)
        cases = edge + [sum(half << (W * j) for j in range(252 // W)) % L, sum((half + 1) << (W * j) for j in range(252 // W)) % L]
        cases += [int.from_bytes(rng.bytes(32), "little") % L for _ in range(40)]
        for v in cases:
            assert emul.emul_digit_stream_mismatches(le32(v), W) == 0, (W, hex(v))


def test_emul_fixed_base_scanning_policy(emul, pyoracle, golden):
    """GatherScan (the signer's constant-time option: all 9 entries of a 4-bit window are read, the digit selects
    arithmetically) gives the same points as the indexed gather: golden vectors and the signed-digit edge scalars"""
    o = pyoracle
    out = ctypes.create_string_buffer(64)
    l = o.SUBORDER
    for c in golden["oracle_vectors"]["fixed_base"][::2]:
        emul.emul_fixed_base_scan(le32(hexint(c["n"])), 4, out)
        assert unpack(out.raw, 2)[0] == tuple(hexint(v) for v in c["out"]), c["n"]
    for W in (4, 5):
        half = 1 << (W - 1)
        for n in (0, 1, half, half + 1, (1 << W) - 1, l - 1, l, l + 1, (1 << 256) - 1,
                  sum(half << (W * j) for j in range(252 // W)) % l, sum((half + 1) << (W * j) for j in range(252 // W))):
            emul.emul_fixed_base_scan(le32(n % (1 << 256)), W, out)
            assert unpack(out.raw, 2)[0] == o.mul_scalar(o.B8, n % (1 << 256)), (W, hex(n))


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


def test_emul_verify_half_size_scalar_boundaries(emul, oracle, pyoracle):
    """The EdDSA fast path multiplies the check by a short odd v (bjj_device.hpp: lattice_short_pair).
    Properties of the pair on edge / random kappa, and verdicts for signatures whose pair sits at the
    window-count boundary (>= 131 bits), found by scanning seeded candidates with the oracle's Poseidon."""
    import numpy as np
    from babyjubjub_rs_amd import workload as w
    o = pyoracle
    L = o.SUBORDER
    uo, vo = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    ks = [0, 1, 2, 3, L - 1, L - 2, (L + 1) // 2, (L - 1) // 2, 1 << 126, (1 << 126) - 1, (1 << 126) + 1, 1 << 250,
          L // 3, L - (1 << 126), (1 << 200) + 1] + w.to_ints(w.random_u256(0xABCDEF, 3000))
    for k in ks:
        k %= L
        neg = emul.emul_short_pair(le32(k), uo, vo)
        u, v = unpack(uo.raw)[0], unpack(vo.raw)[0]
        v = -v if neg else v
        assert (u - v * k) % L == 0 and v % 2 == 1 and v % L != 0, k

    def pair_bits(kappa):  # independent restatement of the selection rule
        r0, t0, r1, t1 = L, 0, kappa, 1
        while r1 >= (1 << 126):
            q = r0 // r1
            r0, r1, t0, t1 = r1, r0 - q * r1, t1, t0 - q * t1
        if t1 % 2:
            return max(r1.bit_length(), abs(t1).bit_length())
        best = max(r0.bit_length(), abs(t0).bit_length())
        q = r0 // r1
        r2, t2 = r0 - q * r1, t0 - q * t1
        if r2:
            best = min(best, max(r2.bit_length(), abs(t2).bit_length()))
        return best

    n = 6000
    kk = [v % L for v in w.to_ints(w.random_u256(w.SEED_KEYS ^ 0xB0, n))]
    rho = [v % L for v in w.to_ints(w.random_u256(w.SEED_NONCES ^ 0xB0, n))]
    msg = w.random_u256(w.SEED_MSGS ^ 0xB0, n, 0, 3)
    A, R = oracle.mul_fixed_base(w.from_ints(kk)), oracle.mul_fixed_base(w.from_ints(rho))
    hm = w.to_ints(oracle.poseidon5(np.concatenate([R, A, msg], axis=1)))
    bits = [pair_bits(hm[i] % L) for i in range(n)]
    picks = [i for i in range(n) if bits[i] >= 131][:6] + [0, 1]
    assert len(picks) >= 3
    # since round 4 an item runs ceil((bits + 2) / 4) windows (the host harness: exactly its own minimum), so EVERY multiple of four
    # is a window-count boundary: a few items of each bit length that occurs, both sides of 122 | 123, 126 | 127, 130 | 131
    for b in sorted(set(bits)):
        picks += [i for i in range(n) if bits[i] == b][:4]
    assert {126, 127, 130, 131} <= set(bits)
    for i in picks:
        S = (rho[i] + 8 * hm[i] * kk[i]) % L
        for s in (S, S ^ 2):
            got = emul.emul_verify(A[i].tobytes(), R[i].tobytes(), le32(s), msg[i].tobytes(), 6)
            want = oracle.verify(A[i], R[i], np.frombuffer(le32(s), np.uint8), msg[i])[0]
            assert got == want == (1 if s == S else 0), (i, pair_bits(hm[i] % L))
