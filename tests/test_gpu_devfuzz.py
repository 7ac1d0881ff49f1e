"""Device-side unit fuzz of what actually ships (tests/devfuzz): the inline-asm column multiplier / squarer, the asm
dot products, the binary-GCD inversion and the extended-coordinate formulas, run on the GPU on adversarial raw limb
vectors -- all-ones 29- and 30-bit limbs, values at 2r / 4r / 13r, the largest top limbs the contract of
babyjubjub-rs_amd/csrc/fr.hpp:14-20 allows -- and checked (i) against Python integers on samples and (ii) bit for bit
against the compiler-scheduled portable columns (-DBJJ_NO_ASM_COLUMNS) on 10^7 random pairs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
RADIX = 1 << 261
RINV = pow(RADIX, -1, R_MOD)
TOP_R = R_MOD >> 232            # 3171406: top limb of r
A_REF, D_REF = 168700, 168696
OPS = dict(mul=0, sqr=1, inv=2, dot6=3, dot15=4, dot151=5, dot2add=6, madd=7, dbl=8, addpn=9, dbl_not=10, madd_not=11, consts=20)


def val(limbs):
    return sum(int(x) << (29 * i) for i, x in enumerate(limbs))


def nform(v):
    """canonical limb decomposition of an integer < 2^258: limbs 0..7 < 2^29, the rest in limb 8"""
    return [(v >> (29 * i)) & 0x1fffffff for i in range(8)] + [v >> 232]


def lazy(v, rnd):
    """the same value with some carries pushed DOWN: limb i += 2^29, limb i+1 -= 1 (limbs stay < 2^30)"""
    l = nform(v)
    for i in range(8):
        if l[i + 1] > 0 and rnd.random() < 0.5:
            l[i] += 1 << 29
            l[i + 1] -= 1
    return l


class Fz:
    def __init__(self, variant):
        d = os.path.join(ROOT, "tests", "devfuzz")
        so = os.path.join(d, "libbjj_devfuzz_%s.so" % variant)
        src = os.path.join(d, "devfuzz.hip")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            r = subprocess.run(["make", "-s"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            assert r.returncode == 0, r.stdout
        import torch  # noqa: F401  (loads the HIP runtime first, see babyjubjub-rs_amd/_lib.py)
        self.lib = ctypes.CDLL(so)
        self.lib.fz_variant.restype = ctypes.c_char_p
        assert self.lib.fz_variant().decode() == variant
        vp = ctypes.c_void_p
        self.lib.fz_run.argtypes = [ctypes.c_int, vp, vp, vp, vp, ctypes.c_size_t, ctypes.c_int, vp]

    def run(self, op, a, b, c, n, wo, row=0):
        """a, b, c: int32 cuda tensors (or None); returns an (n, wo) int32 cuda tensor of raw limbs"""
        import torch
        out = torch.zeros(n * wo, dtype=torch.int32, device="cuda")
        p = lambda t: (t.data_ptr() if t is not None else 0)  # noqa: E731
        rc = self.lib.fz_run(OPS[op], p(a), p(b), p(c), out.data_ptr(), n, row, 0)
        assert rc == 0
        torch.cuda.synchronize()
        return out.view(n, wo)


@pytest.fixture(scope="module")
def fz():
    return Fz("asm"), Fz("portable")


def dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint32).view(np.int32).reshape(-1)).cuda()


def host(t):
    return t.cpu().numpy().view(np.uint32)


def adversarial_operands(rnd, extra=300):
    """limb vectors that satisfy fr_mul's contract (limbs < 2^30, value < 13 r) and sit on its edges"""
    ones29, ones30 = (1 << 29) - 1, (1 << 30) - 1
    ops = []
    for top in (0, 1, TOP_R - 1, TOP_R, 2 * TOP_R, 4 * TOP_R, 13 * TOP_R - 3):
        ops.append([ones29] * 8 + [top])                    # all-ones 29-bit limbs under every top limb
        if top <= 13 * TOP_R - 6:
            ops.append([ones30] * 8 + [top])                # all-ones 30-bit (lazy) limbs: value = (top + 2) * 2^232 - ...
        ops.append([0] * 8 + [top])
        ops.append([1] + [0] * 7 + [top])
    for k in (1, 2, 4, 8, 13):
        for d in (-1, 0, 1):
            v = k * R_MOD + d
            if 0 <= v < 13 * R_MOD:
                ops.append(nform(v))
                ops.append(lazy(v, rnd))
    ops.append(nform(13 * R_MOD - 1))
    for _ in range(extra):
        v = rnd.randrange(13 * R_MOD)
        ops.append(nform(v))
        ops.append(lazy(v, rnd))
    return ops


def check_nform_below(limbs, bound):
    assert all(int(x) < (1 << 29) for x in limbs[:8]), limbs
    assert val(limbs) < bound, (val(limbs), bound)


def test_shipped_multiplier_on_adversarial_limbs(fz):
    import random
    rnd = random.Random(2024)
    ops = adversarial_operands(rnd)
    pairs = [(x, y) for x in ops[:60] for y in ops[:60]] + [(rnd.choice(ops), rnd.choice(ops)) for _ in range(20000)]
    a = np.array([p[0] for p in pairs], dtype=np.uint64).astype(np.uint32)
    b = np.array([p[1] for p in pairs], dtype=np.uint64).astype(np.uint32)
    n = len(pairs)
    for f in fz:
        m = host(f.run("mul", dev(a), dev(b), None, n, 9))
        s = host(f.run("sqr", dev(a), None, None, n, 9))
        for i in range(n):
            va, vb = val(a[i]), val(b[i])
            check_nform_below(m[i], 2 * R_MOD)
            assert val(m[i]) % R_MOD == va * vb * RINV % R_MOD, (i, a[i], b[i])
            check_nform_below(s[i], 2 * R_MOD)
            assert val(s[i]) % R_MOD == va * va * RINV % R_MOD, (i, a[i])
    # both builds agree bit for bit on the adversarial set too
    assert (host(fz[0].run("mul", dev(a), dev(b), None, n, 9)) == host(fz[1].run("mul", dev(a), dev(b), None, n, 9))).all()


def random_limbs(rng, n, words, top_max, low_bits=29):
    x = rng.integers(0, 1 << low_bits, size=(n, words, 9), dtype=np.uint64)
    x[:, :, 8] = rng.integers(0, top_max, size=(n, words), dtype=np.uint64)
    return x.astype(np.uint32).reshape(n, words * 9)


def test_asm_columns_equal_portable_columns_1e7(fz):
    """10^7 random operand pairs in the widest form the contract allows (30-bit lazy limbs, values up to 13 r):
    the inline-asm columns and the compiler-scheduled columns must produce identical limbs."""
    import torch
    rng = np.random.default_rng(7)
    n = 10_000_000
    top = 13 * TOP_R - 4
    a = dev(random_limbs(rng, n, 1, top, 30))
    b = dev(random_limbs(rng, n, 1, top, 30))
    for op in ("mul", "sqr"):
        x = fz[0].run(op, a, b, None, n, 9)
        y = fz[1].run(op, a, b, None, n, 9)
        assert torch.equal(x, y), op
        assert int((x[:, :8] >> 29).abs().max()) == 0          # N-form out
    # sample against Python integers
    idx = np.arange(0, n, 9973)
    ha, hb, hm = host(a.view(n, 9)[idx]), host(b.view(n, 9)[idx]), host(fz[0].run("mul", a, b, None, n, 9)[idx])
    for i in range(idx.size):
        assert val(hm[i]) % R_MOD == val(ha[i]) * val(hb[i]) * RINV % R_MOD
        assert val(hm[i]) < 2 * R_MOD


def test_inversion_on_device(fz):
    import random
    rnd = random.Random(5)
    vals = [0, 1, 2, R_MOD - 1, R_MOD, R_MOD + 1, 2 * R_MOD - 1, 15 * R_MOD + 12345, 16 * R_MOD - 1] + \
           [rnd.randrange(16 * R_MOD) for _ in range(3000)] + [rnd.randrange(1 << k) for k in range(1, 254)]
    a = np.array([nform(v) for v in vals], dtype=np.uint64).astype(np.uint32)
    n = len(vals)
    x, y = host(fz[0].run("inv", dev(a), None, None, n, 9)), host(fz[1].run("inv", dev(a), None, None, n, 9))
    assert (x == y).all()
    for i, v in enumerate(vals):
        # Montgomery inverse: x = R^2 / v  (0 -> 0), canonical
        want = 0 if v % R_MOD == 0 else RADIX * RADIX * pow(v, -1, R_MOD) % R_MOD
        assert val(x[i]) == want, hex(v)


def test_asm_dot_products(fz):
    """the Poseidon dot products with scalar-register matrix operands (gen_fr_asm.py) against Python integers and the
    portable form; state operands up to the bounds poseidon.hpp states (N-form, values < 4 r; addend < 14 r)"""
    import torch
    rng = np.random.default_rng(11)
    consts = host(fz[0].run("consts", None, None, None, 1, 9 * (36 + 660 + 30 + 3)).view(-1, 9))
    cv = [val(c) for c in consts]
    PM, PSP, PCAB = cv[:36], cv[36:696], cv[696:726]
    n = 200_000
    b6 = random_limbs(rng, n, 6, 4 * TOP_R)
    b7 = random_limbs(rng, n, 7, 4 * TOP_R)
    a1 = random_limbs(rng, n, 1, 4 * TOP_R)
    c1 = random_limbs(rng, n, 1, 14 * TOP_R - 2)
    # edge rows: all-ones limbs under the largest top limb
    b6[0] = np.array(([(1 << 29) - 1] * 8 + [4 * TOP_R - 1]) * 6, dtype=np.uint32)
    b7[0] = np.array(([(1 << 29) - 1] * 8 + [4 * TOP_R - 1]) * 7, dtype=np.uint32)
    c1[0] = np.array([(1 << 29) - 1] * 8 + [14 * TOP_R - 3], dtype=np.uint32)
    d6, d7, da, dc = dev(b6), dev(b7), dev(a1), dev(c1)
    idx = np.concatenate([[0], np.arange(1, n, 499)])
    for row in (0, 1, 5, 17, 59):
        outs = {}
        for op, args, wo in (("dot6", (None, d6, None), 9), ("dot15", (None, d6, None), 9), ("dot151", (None, d7, None), 9),
                             ("dot2add", (da, dev(b6[:, :9].copy()), dc), 9)):
            x = fz[0].run(op, *args, n, wo, row)
            y = fz[1].run(op, *args, n, wo, row)
            assert torch.equal(x, y), (op, row)
            outs[op] = host(x)
        for i in idx:
            s6 = [val(b6[i, 9 * j:9 * j + 9]) for j in range(6)]
            s7 = [val(b7[i, 9 * j:9 * j + 9]) for j in range(7)]
            m = PM[6 * (row % 6):6 * (row % 6) + 6]
            sp = PSP[11 * row:11 * row + 11]
            assert val(outs["dot6"][i]) % R_MOD == sum(x * y for x, y in zip(m, s6)) * RINV % R_MOD
            assert val(outs["dot15"][i]) % R_MOD == sum(x * y for x, y in zip(sp[:6], s6)) * RINV % R_MOD
            assert val(outs["dot151"][i]) % R_MOD == (sum(x * y for x, y in zip(sp[:6], s7[:6])) + PCAB[row % 30] * s7[6]) * RINV % R_MOD
            av, bv, cvv = val(a1[i]), val(b6[i, :9]), val(c1[i])
            assert val(outs["dot2add"][i]) % R_MOD == ((sp[6] * av + sp[7] * bv) * RINV + cvv) % R_MOD
            for op in outs:
                assert all(int(t) < (1 << 29) for t in outs[op][i][:8]) and int(outs[op][i][8]) < (1 << 26)


def test_extended_coordinate_formulas(fz):
    """ext_madd / ext_add_pn / ext_dbl (a' = -1 curve) on the device: random points of the group in random projective
    scalings and lazy (< 2r) coordinates, against the affine group law in Python integers; asm == portable bit for bit."""
    import random
    import torch
    rnd = random.Random(99)
    consts = host(fz[0].run("consts", None, None, None, 1, 9 * (36 + 660 + 30 + 3)).view(-1, 9))
    DP = val(consts[726]) * RINV % R_MOD          # D' = -D/A
    F = val(consts[728]) * RINV % R_MOD           # sqrt(-A)
    assert DP == (-D_REF * pow(A_REF, -1, R_MOD)) % R_MOD and F * F % R_MOD == (-A_REF) % R_MOD
    inv = lambda v: pow(v, -1, R_MOD)  # noqa: E731

    def add(p, q):
        (x1, y1), (x2, y2) = p, q
        t = DP * x1 * x2 * y1 * y2 % R_MOD
        return ((x1 * y2 + y1 * x2) * inv(1 + t) % R_MOD, (y1 * y2 + x1 * x2) * inv(1 - t) % R_MOD)

    g = (F * 5299619240641551281634865583518297030282874472190772894086521144482721001553 % R_MOD,
         16950150798460657717958625567821834550301663161624707787222815936182638968203)
    assert (-g[0] * g[0] + g[1] * g[1] - 1 - DP * g[0] * g[0] * g[1] * g[1]) % R_MOD == 0
    pts = [(0, 1), g]
    for _ in range(400):
        pts.append(add(pts[-1], rnd.choice(pts[1:])))
    mont = lambda v: v * RADIX % R_MOD  # noqa: E731
    loose = lambda v: nform(mont(v) + (R_MOD if rnd.random() < 0.5 else 0))  # noqa: E731  (N-form, < 2r)
    n = 4000
    P, Qn, Qp, exp_add, exp_dbl = [], [], [], [], []
    for _ in range(n):
        p, q = rnd.choice(pts), rnd.choice(pts)
        z = rnd.randrange(1, R_MOD)
        P.append(loose(p[0] * z) + loose(p[1] * z) + loose(z) + loose(p[0] * p[1] * z))
        Qn.append(loose(q[1] - q[0]) + loose(q[1] + q[0]) + loose(2 * DP * q[0] * q[1]))
        z2 = rnd.randrange(1, R_MOD)
        Qp.append(loose((q[1] - q[0]) * z2) + loose((q[1] + q[0]) * z2) + loose(2 * DP * q[0] * q[1] * z2) + loose(2 * z2))
        exp_add.append(add(p, q))
        exp_dbl.append(add(p, p))
    dP, dQn, dQp = (dev(np.array(t, dtype=np.uint64).astype(np.uint32)) for t in (P, Qn, Qp))

    def affine(row, need_t):
        X, Y, Z, T = (val(row[9 * k:9 * k + 9]) * RINV % R_MOD for k in range(4))
        zi = inv(Z)
        if need_t:
            assert T * Z % R_MOD == X * Y % R_MOD
        return (X * zi % R_MOD, Y * zi % R_MOD)

    for op, b, exp, need_t in (("madd", dQn, exp_add, True), ("madd_not", dQn, exp_add, False), ("addpn", dQp, exp_add, True),
                               ("dbl", None, exp_dbl, True), ("dbl_not", None, exp_dbl, False)):
        x, y = fz[0].run(op, dP, b, None, n, 36), fz[1].run(op, dP, b, None, n, 36)
        assert torch.equal(x, y), op
        hx = host(x)
        for i in range(n):
            assert affine(hx[i], need_t) == exp[i], (op, i)
            for k in range(3):
                check_nform_below(hx[i][9 * k:9 * k + 9], 2 * R_MOD)     # outputs of fr_mul: < 2r, as the next formula needs


# ---- slot queues (csrc/slot_queue.hpp): hand-over and the bounded wait ------------------------------------------------
def _slotq_lib():
    d = os.path.join(ROOT, "tests", "devfuzz")
    so = os.path.join(d, "libbjj_slotq_test.so")
    if not os.path.exists(so) or os.path.getmtime(os.path.join(d, "slotq.hip")) > os.path.getmtime(so):
        r = subprocess.run(["make", "-s"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    import torch  # noqa: F401
    lib = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    lib.sq_run.argtypes = [vp, ctypes.c_uint32, vp, vp, vp, ctypes.c_int, ctypes.c_int, vp]
    lib.sq_spin_limit.restype = ctypes.c_uint
    return lib


def _queue(torch, nx, cap, filled=True):
    """host image of the rings: per XCD [head, tail, err, ..13 unused.., cap entries = slot id + 1]"""
    q = np.zeros((nx, 16 + cap), np.uint32)
    if filled:
        for x in range(nx):
            q[x, 16:] = x * cap + np.arange(cap, dtype=np.uint32) + 1
    return torch.from_numpy(q.view(np.int32).reshape(-1).copy()).cuda()


def test_slot_queue_hands_every_slot_to_one_holder_at_a_time():
    import torch
    lib = _slotq_lib()
    nx, cap, blocks = 8, 24, 4096                      # far more workgroups than slots: every slot changes hands many times
    q = _queue(torch, nx, cap)
    owner = torch.zeros(nx * cap + nx, dtype=torch.int32, device="cuda")
    got = torch.full((blocks,), -1, dtype=torch.int32, device="cuda")
    clash = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.sq_run(q.data_ptr(), cap | (nx << 16), owner.data_ptr(), got.data_ptr(), clash.data_ptr(), blocks, 40, None) == 0
    torch.cuda.synchronize()
    h = q.cpu().numpy().view(np.uint32).reshape(nx, 16 + cap)
    g = got.cpu().numpy()
    assert int(clash.item()) == 0 and (owner.cpu().numpy() == 0).all()
    assert (h[:, 2] == 0).all()                                            # nobody gave up
    assert ((g >= 0) & (g < nx * cap)).all()                               # regular slots only
    for x in range(nx):                                                    # every ring holds exactly its own slots again
        assert sorted(h[x, 16:].tolist()) == list(range(x * cap + 1, x * cap + cap + 1))
    used_x = np.unique(g // cap)
    assert len(used_x) >= 1 and (h[used_x, 0] == h[used_x, 1]).all()       # as many pops as pushes (tickets wrap together)


def test_slot_queue_starved_pop_gives_up_flags_it_and_takes_the_overflow_slot():
    """rings without a single free slot (what a kernel that died holding its slots leaves behind): every pop would wait for
    ever.  With the bound it gives up after BJJ_SLOT_SPIN_LIMIT polls, counts itself in the ring's error word and continues on
    the XCD's overflow slot -- the launch ENDS, and the host can see why its results are invalid (bjj_sync reports it)."""
    import torch
    lib = _slotq_lib()
    assert lib.sq_spin_limit() == 2000
    nx, cap, blocks = 8, 24, 512
    q = _queue(torch, nx, cap, filled=False)
    owner = torch.zeros(nx * cap + nx, dtype=torch.int32, device="cuda")
    got = torch.full((blocks,), -1, dtype=torch.int32, device="cuda")
    clash = torch.zeros(1, dtype=torch.int32, device="cuda")
    assert lib.sq_run(q.data_ptr(), cap | (nx << 16), owner.data_ptr(), got.data_ptr(), clash.data_ptr(), blocks, 1, None) == 0
    torch.cuda.synchronize()                                                # returns: no hang
    h = q.cpu().numpy().view(np.uint32).reshape(nx, 16 + cap)
    g = got.cpu().numpy()
    assert int(h[:, 2].sum()) == blocks                                     # every workgroup flagged itself
    assert ((g >= nx * cap) & (g < nx * cap + nx)).all()                    # ... and worked on an overflow slot
    assert (h[:, 16:] == 0).all()                                           # overflow slots are never queued


# ---------------------------------------------------------------- the wave-wide window count of verify's joint loop
L_SUB = 2736030358979909402780800718157159386076813972158567259200215660948447373041


def _joint_lib():
    d = os.path.join(ROOT, "tests", "devfuzz")
    so = os.path.join(d, "libbjj_joint_test.so")
    if not os.path.exists(so) or os.path.getmtime(os.path.join(d, "joint.hip")) > os.path.getmtime(so):
        r = subprocess.run(["make", "-s"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    import torch  # noqa: F401
    lib = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    lib.jt_run.argtypes = [vp, vp, vp, vp, ctypes.c_size_t, vp, vp, vp, ctypes.c_int, vp]
    return lib


def _windows_needed(bits):
    return 1 if bits <= 2 else (bits + 5) >> 2      # scalars < 2^(4 w - 2): ceil((bits + 2) / 4)


def test_joint_loop_runs_the_window_count_of_the_widest_lane_of_each_wave(oracle):
    """VERDICT r04 item 3.  verify's joint loop (joint_short_pair -> joint_mul_windowed, csrc/bjj_device.hpp) runs
    jw = wave_max(windows the item needs) windows: a cross-lane, data-dependent trip count.  Random signatures reach 34 windows
    for 0.1 % of the items and 35..64 never, so the cases are DIRECTED here, through the shipped function under the shipped
    gather policy (tests/devfuzz/joint.hip): one lane at 250 bits -- what the odd kappa = (l + 1) / 2 produces, 63 windows; 251 bits
    give all 64 -- between lanes at 1, 2, 3, 126, 127, 130 and 131 bits; waves whose widest lane needs exactly 32 / 33 / 34 windows; the wide
    value in u or in |v|; a partly filled last wave (its idle lanes repeat the last item).  Checked: the window count every
    wave ran, and u*P1 + |v|*P2 of EVERY item against the oracle's mul_var_base + point_add (src/lib.rs:405-411 through
    oracle/bjj_ref.c) -- also with the identity wave_max (each lane exactly its own minimum), bit for bit the same points."""
    import random
    import sys
    import torch
    from babyjubjub_rs_amd import workload as w
    sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd", "csrc"))
    import gen_tables
    f = gen_tables.sqrt_mod(-A_REF)
    f = min(f, R_MOD - f)                       # the root gen_tables.py ships as BJJ_K_F
    finv = pow(f, -1, R_MOD)
    rnd = random.Random(0x6a6f696e74)

    def with_bits(b):
        return 0 if b == 0 else (1 << (b - 1)) | rnd.getrandbits(b - 1)

    # (bit lengths of u, bit lengths of |v|) per lane, wave by wave
    small = [1, 2, 3, 64, 126]
    waves = [
        ([small[i % 5] for i in range(64)], [small[(i + 2) % 5] for i in range(64)]),                   # 32 windows
        ([126] * 64, [127 if i == 9 else 126 for i in range(64)]),                                       # one lane at 127: 33
        ([130 if i == 63 else 1 for i in range(64)], [1] * 64),                                         # 130 bits in the last lane: 33
        ([5] * 64, [131 if i == 0 else 100 for i in range(64)]),                                        # 131 bits in |v| of lane 0: 34
        ([250 if i == 7 else (1, 126, 127, 130, 131)[i % 5] for i in range(64)], [126] * 64),           # kappa = (l+1)/2: u of 250 bits, 63 windows
        ([126] * 64, [251 if i == 40 else 3 for i in range(64)]),                                       # 251 bits (the width of l) in |v|: all 64 windows
        ([0 if i % 2 else 126 for i in range(64)], [1] * 64),                                           # u == 0 lanes (identity entries only)
        ([(126, 2)[i % 2] for i in range(37)], [130 if i == 36 else 90 for i in range(37)]),            # partly filled: the LAST item is the wide one
    ]
    ub = [b for wv in waves for b in wv[0]]
    vb = [b for wv in waves for b in wv[1]]
    n = len(ub)
    assert n == 7 * 64 + 37
    u = [with_bits(b) for b in ub]
    v = [with_bits(b) | 1 for b in vb]                     # |v| is odd in verify
    u[4 * 64 + 7] = ((L_SUB + 1) // 2) % (1 << 250) | (1 << 249)   # a 250-bit value with kappa's low half
    assert all(x.bit_length() == b for x, b in zip(u, ub)) and all(x.bit_length() == b for x, b in zip(v, vb))
    # points of the whole group: A = k B8 + c T8 (cofactor components included), P2 = k' B8
    T8 = (4342719913949491028786768530115087822524712248835451589697801404893164183326,
          4826523245007015323400664741523384119579596407052839571721035538011798951543)
    kb = oracle.mul_fixed_base(w.random_u256(0x6a74, n, 0))
    tors = oracle.mul_var_base(np.tile(w.from_ints([T8[0], T8[1]]).reshape(1, 64), (8, 1)), w.from_ints(list(range(8))))
    A = oracle.point_add(kb, tors[np.arange(n) % 8])
    P2 = oracle.mul_fixed_base(w.random_u256(0x6a75, n, 0))
    ub_, vb_ = w.from_ints(u), w.from_ints(v)
    # expected: u * (8 A) + v * P2
    P1 = oracle.mul_var_base(A, w.from_ints([8] * n))
    want = oracle.point_add(oracle.mul_var_base(P1, ub_), oracle.mul_var_base(P2, vb_))
    lib = _joint_lib()
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).cuda()  # noqa: E731
    d_u, d_v, d_a, d_p = up(ub_), up(vb_), up(A), up(P2)
    tables = torch.zeros(((n + 63) // 64) * 64 * lib.jt_table_words(), dtype=torch.int32, device="cuda")
    results = []
    for per_lane in (0, 1):
        out = torch.zeros(n * 96, dtype=torch.uint8, device="cuda")
        win = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        assert lib.jt_run(d_u.data_ptr(), d_v.data_ptr(), d_a.data_ptr(), d_p.data_ptr(), n, tables.data_ptr(), out.data_ptr(),
                          win.data_ptr(), per_lane, 0) == 0
        torch.cuda.synchronize()
        win = win.cpu().numpy()
        xyz = w.to_ints(out.cpu().numpy().reshape(n * 3, 32))
        got = []
        for i in range(n):
            X, Y, Z = xyz[3 * i:3 * i + 3]
            assert Z % R_MOD != 0, i
            zi = pow(Z, -1, R_MOD)
            got.append((X * zi * finv % R_MOD, Y * zi % R_MOD))
        got = w.from_ints([c for p in got for c in p]).reshape(n, 64)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, (per_lane, bad[:10], [(ub[i], vb[i]) for i in bad[:10]])
        need = [_windows_needed(max(a, b)) for a, b in zip(ub, vb)]
        if per_lane:
            assert (win == np.array(need)).all()
        else:
            lo = 0
            for wv in waves:                      # every lane of a wave ran the maximum over the wave's lanes
                m = len(wv[0])
                assert (win[lo:lo + m] == max(need[lo:lo + m])).all(), (lo, win[lo:lo + m], max(need[lo:lo + m]))
                lo += m
            assert [int(win[64 * k]) for k in range(8)] == [32, 33, 33, 34, 63, 64, 32, 33]
        results.append(got)
    assert (results[0] == results[1]).all()
