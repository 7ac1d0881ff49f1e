#!/usr/bin/env python3
"""
Generates bjj_constants.inc for the HIP library (part of the product build;
standalone -- does NOT import oracle/).

Everything the kernels need as compile-time data, in the device's own
representation (9 x 29-bit little-endian limbs, Montgomery form a*2^261 mod r
unless the name says PLAIN):

  * curve constants of the reference crate (src/lib.rs:28-60): A, D, B8
  * the isomorphic a' = -1 curve used internally: F = sqrt(-A), FINV, D' = -D/A
  * Poseidon t=6 round constants / MDS matrix of poseidon-rs 0.0.8
    (Cargo.toml:20), regenerated with the Poseidon reference Grain-LFSR
    procedure
  * the "optimised Poseidon" equivalent constants (sparse partial rounds), derived
    algebraically from the same C / M (identical outputs)
  * group-order multiples for scalar reduction

tests/test_constants.py cross-checks the output against the oracle's
independently written generator.
"""
import os
import sys

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
A = 168700
D = 168696
B8 = (
    5299619240641551281634865583518297030282874472190772894086521144482721001553,
    16950150798460657717958625567821834550301663161624707787222815936182638968203,
)
ORDER = 21888242871839275222246405745257275088614511777268538073601725287587578984328
SUBORDER = ORDER >> 3
R = 1 << 261  # Montgomery radix of the 9 x 29-bit limb field (fr.hpp)
T = 6
RF = 8
RP = 60


def inv(a):
    return pow(a % Q, Q - 2, Q)


def sqrt_mod(n):
    """Tonelli-Shanks in F_r (r - 1 = 2^28 * odd)."""
    n %= Q
    assert pow(n, (Q - 1) // 2, Q) == 1, "not a square"
    q, s = Q - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (Q - 1) // 2, Q) != Q - 1:
        z += 1
    m, c, t, r = s, pow(z, q, Q), pow(n, q, Q), pow(n, (q + 1) // 2, Q)
    while t != 1:
        i, tt = 0, t
        while tt != 1:
            tt = tt * tt % Q
            i += 1
        b = pow(c, 1 << (m - i - 1), Q)
        m, c = i, b * b % Q
        t, r = t * c % Q, r * b % Q
    return r


# ---------------------------------------------------------------- Grain LFSR
class Grain:
    def __init__(self, field, sbox, n, t, rf, rp):
        bits = []
        for v, w in ((field, 2), (sbox, 4), (n, 12), (t, 12), (rf, 10), (rp, 10)):
            bits += [(v >> (w - 1 - i)) & 1 for i in range(w)]
        bits += [1] * 30
        self.s = bits
        for _ in range(160):
            self.clock()

    def clock(self):
        s = self.s
        b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        del s[0]
        s.append(b)
        return b

    def shrunk_bit(self):
        while True:
            keep, val = self.clock(), self.clock()
            if keep:
                return val

    def sample(self, n=254):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.shrunk_bit()
        return v


def poseidon_constants():
    g = Grain(1, 0, 254, T, RF, RP)
    C = []
    while len(C) < (RF + RP) * T:
        v = g.sample()
        if v < Q:
            C.append(v)
    xs = [g.sample() % Q for _ in range(T)]
    ys = [g.sample() % Q for _ in range(T)]
    M = [[inv(xs[i] + ys[j]) for j in range(T)] for i in range(T)]
    return C, M


def poseidon_plain(C, M, inputs):
    st = [0] + list(inputs)
    for r in range(RF + RP):
        st = [(st[j] + C[r * T + j]) % Q for j in range(T)]
        if r < RF // 2 or r >= RF // 2 + RP:
            st = [pow(v, 5, Q) for v in st]
        else:
            st[0] = pow(st[0], 5, Q)
        st = [sum(M[i][j] * st[j] for j in range(T)) % Q for i in range(T)]
    return st[0]


# ---------------------------------------------------------------- sparse partial rounds
def mat_vec(M, v):
    return [sum(M[i][j] * v[j] for j in range(len(v))) % Q for i in range(len(M))]


def mat_mul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][x] * B[x][j] for x in range(k)) % Q for j in range(m)] for i in range(n)]


def mat_inv(A):
    n = len(A)
    a = [list(r) + [1 if i == j else 0 for j in range(n)] for i, r in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c] % Q)
        a[c], a[piv] = a[piv], a[c]
        iv = inv(a[c][c])
        a[c] = [x * iv % Q for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % Q for x, y in zip(a[r], a[c])]
    return [r[n:] for r in a]


def poseidon_sparse_constants(C, M):
    """Equivalent form of the 60 partial rounds (same outputs, fewer multiplications).

    state' = M . sbox0(state + c)  with sbox0 touching element 0 only.
    (1) the constants of elements 1..5 commute with sbox0, so they are pushed through M into
        the next round; every partial round keeps only a scalar constant k_p for element 0,
        the last tail lands in the constants of full round 64.
    (2) M^(p) = A_p . B_p with A_p = diag(1, Mh) (commutes with the next sbox0) and the sparse
        B_p = [[m00, v], [Mh^-1 w, I]];  M^(p+1) = M . A_p.  A_59 is applied once at the end.
    Returns (CF[48], K[60], S[60][11] = m00, v[5], what[5], AL[25])."""
    full_c = [list(C[r * T:(r + 1) * T]) for r in range(RF + RP)]
    kp = [list(full_c[RF // 2 + p]) for p in range(RP)]
    for p in range(RP):
        tail = [0] + kp[p][1:]
        pushed = mat_vec(M, tail)
        if p + 1 < RP:
            kp[p + 1] = [(a + b) % Q for a, b in zip(kp[p + 1], pushed)]
        else:
            full_c[RF // 2 + RP] = [(a + b) % Q for a, b in zip(full_c[RF // 2 + RP], pushed)]
    K = [kp[p][0] for p in range(RP)]
    S = []
    Mp = [list(r) for r in M]
    A_last = None
    for p in range(RP):
        m00 = Mp[0][0]
        v = Mp[0][1:]
        w = [[Mp[i][0]] for i in range(1, T)]
        Mh = [Mp[i][1:] for i in range(1, T)]
        what = [x[0] for x in mat_mul(mat_inv(Mh), w)]
        S.append([m00] + v + what)
        A = [[1] + [0] * (T - 1)] + [[0] + Mh[i] for i in range(T - 1)]
        A_last = Mh
        Mp = mat_mul(M, A)
    CF = [c for r in range(RF // 2) for c in full_c[r]] + [c for r in range(RF // 2 + RP, RF + RP) for c in full_c[r]]
    AL = [A_last[i][j] for i in range(T - 1) for j in range(T - 1)]
    return CF, K, S, AL


def poseidon_sparse(CF, K, S, AL, M, inputs):
    st = [0] + list(inputs)
    for r in range(RF // 2):
        st = [pow((st[j] + CF[r * T + j]) % Q, 5, Q) for j in range(T)]
        st = mat_vec(M, st)
    for p in range(RP):
        x0 = pow((st[0] + K[p]) % Q, 5, Q)
        s = S[p]
        u0 = (s[0] * x0 + sum(s[1 + j] * st[1 + j] for j in range(T - 1))) % Q
        st = [u0] + [(s[T + j] * x0 + st[1 + j]) % Q for j in range(T - 1)]
    st = [st[0]] + [sum(AL[i * (T - 1) + j] * st[1 + j] for j in range(T - 1)) % Q for i in range(T - 1)]
    for r in range(RF // 2):
        st = [pow((st[j] + CF[(RF // 2 + r) * T + j]) % Q, 5, Q) for j in range(T)]
        st = mat_vec(M, st)
    return st[0]


def poseidon_pair_constants(S):
    """Partial rounds taken in PAIRS (a, b) = (2p, 2p+1): round b's dot product sees the elements 1..5 as they were
    BEFORE round a's update, plus one extra term  cab * x0a  with  cab = sum_j v_b[j] * what_a[j]; the elements are then
    updated once per pair by  st_j += what_a[j] x0a + what_b[j] x0b  (one reduction instead of two)."""
    return [sum(S[2 * p + 1][1 + j] * S[2 * p][T + j] for j in range(T - 1)) % Q for p in range(RP // 2)]


def poseidon_sparse_paired(CF, K, S, AL, CAB, M, inputs):
    st = [0] + list(inputs)
    for r in range(RF // 2):
        st = [pow((st[j] + CF[r * T + j]) % Q, 5, Q) for j in range(T)]
        st = mat_vec(M, st)
    for p in range(RP // 2):
        sa, sb = S[2 * p], S[2 * p + 1]
        x0a = pow((st[0] + K[2 * p]) % Q, 5, Q)
        u0a = (sa[0] * x0a + sum(sa[1 + j] * st[1 + j] for j in range(T - 1))) % Q
        x0b = pow((u0a + K[2 * p + 1]) % Q, 5, Q)
        u0b = (sb[0] * x0b + CAB[p] * x0a + sum(sb[1 + j] * st[1 + j] for j in range(T - 1))) % Q
        st = [u0b] + [(sa[T + j] * x0a + sb[T + j] * x0b + st[1 + j]) % Q for j in range(T - 1)]
    st = [st[0]] + [sum(AL[i * (T - 1) + j] * st[1 + j] for j in range(T - 1)) % Q for i in range(T - 1)]
    for r in range(RF // 2):
        st = [pow((st[j] + CF[(RF // 2 + r) * T + j]) % Q, 5, Q) for j in range(T)]
        st = mat_vec(M, st)
    return st[0]


# ---------------------------------------------------------------- emit helpers
def limbs32(v):
    """9 x 29-bit limbs (fr.hpp N-form); the top limb takes whatever is left (< 2^26)."""
    assert 0 <= v < (1 << 258)
    return "{{" + ",".join("0x%08xu" % ((v >> (29 * i)) & 0x1FFFFFFF if i < 8 else v >> 232) for i in range(9)) + "}}"


def mont(v):
    return (v % Q) * R % Q


def main():
    C, M = poseidon_constants()
    # sanity anchors: public circomlib value poseidon([1,2,3,4,5])
    assert poseidon_plain(C, M, [1, 2, 3, 4, 5]) == \
        6183221330272524995739186171720101788151706631170188140075976616310159254464

    f = sqrt_mod(-A)
    f = min(f, Q - f)
    dprime = (-D * inv(A)) % Q
    assert pow(dprime, (Q - 1) // 2, Q) == Q - 1  # non-square => complete addition law
    assert (f * f + A) % Q == 0

    o = []
    o.append("// GENERATED by babyjubjub-rs_amd/csrc/gen_tables.py -- do not edit.")
    o.append("// All Fr values: 9 x 29-bit LE limbs, Montgomery form (x * 2^261 mod r) unless marked PLAIN.")
    o.append("#define BJJ_K_A        %s" % limbs32(mont(A)))
    o.append("#define BJJ_K_D        %s" % limbs32(mont(D)))
    o.append("#define BJJ_K_F        %s  // sqrt(-A)" % limbs32(mont(f)))
    o.append("#define BJJ_K_FINV_PLAIN %s  // 1/sqrt(-A), canonical" % limbs32(inv(f)))
    o.append("#define BJJ_K_DP       %s  // d' = -D/A" % limbs32(mont(dprime)))
    o.append("#define BJJ_K_D2P      %s  // 2 d'" % limbs32(mont(2 * dprime)))
    o.append("#define BJJ_K_DPINV    %s  // 1/d'" % limbs32(mont(inv(dprime))))
    o.append("#define BJJ_K_B8X      %s" % limbs32(mont(B8[0])))
    o.append("#define BJJ_K_B8Y      %s" % limbs32(mont(B8[1])))
    # Tonelli-Shanks: r - 1 = 2^28 * s; generator of the 2^28 subgroup = n^s, n the least non-residue
    ts_s = (Q - 1) >> 28
    n = 2
    while pow(n, (Q - 1) // 2, Q) != Q - 1:
        n += 1
    o.append("#define BJJ_K_TS_G     %s  // %d^((r-1)/2^28): order 2^28" % (limbs32(mont(pow(n, ts_s, Q))), n))
    # a^((s-1)/2): sliding 3-bit windows over the constant exponent, odd powers a, a^3, a^5, a^7.
    # Program = (squarings, odd value) steps, most significant first; the first step just loads its power.
    e_bits = bin((ts_s - 1) // 2)[2:]
    prog, nsq, i = [], 0, 0
    while i < len(e_bits):
        if e_bits[i] == "0":
            nsq += 1; i += 1
            continue
        j = min(i + 3, len(e_bits))
        while e_bits[j - 1] == "0":
            j -= 1
        prog.append((0 if not prog else nsq + (j - i), int(e_bits[i:j], 2)))
        nsq, i = 0, j
    x = None
    for sq, val in prog:
        x = pow(7, val, Q) if x is None else pow(x, 1 << sq, Q) * pow(7, val, Q) % Q
    assert pow(x, 1 << nsq, Q) == pow(7, (ts_s - 1) // 2, Q)
    o.append("#define BJJ_TS_POW_STEPS %d" % len(prog))
    o.append("#define BJJ_TS_POW_TAIL %d  // squarings after the last step" % nsq)
    o.append("#define BJJ_TS_POW_PROG { " + ",".join("%d,%d" % (sq, (val - 1) // 2) for sq, val in prog) + " }  // (squarings, (odd power - 1) / 2)")
    o.append("#define BJJ_K_HALFQ    %s  // PLAIN (r-1)/2" % limbs32((Q - 1) // 2))
    # Pohlig-Hellman tables for the discrete log in <G> (order 2^28), four 7-bit digits e = sum e_k 2^(7k):
    #   TSN[k][j] = G^(-j 2^(7k))            (k = 0..2)  strips digit k from b
    #   TSH[k][j] = G^(-(j 2^(7k)) / 2)      (k = 0: index j/2, j even)  the matching factor of the root
    #   hash: canonical Montgomery limb 0 of H^j (H = G^(2^21), order 128) -> j
    G = pow(n, ts_s, Q)
    Ginv = inv(G)
    o.append("#define BJJ_K_TS_NEG { \\")
    for k in range(3):
        for j in range(128):
            o.append("  %s, \\" % limbs32(mont(pow(Ginv, j << (7 * k), Q))))
    o.append("}")
    o.append("#define BJJ_K_TS_HALF { \\")
    for j in range(64):
        o.append("  %s, \\" % limbs32(mont(pow(Ginv, j, Q))))               # e0 = 2j  ->  G^(-j)
    for k in range(1, 4):
        for j in range(128):
            o.append("  %s, \\" % limbs32(mont(pow(Ginv, j << (7 * k - 1), Q))))
    o.append("}")
    H = pow(G, 1 << 21, Q)
    keys = [mont(pow(H, j, Q)) & 0x1FFFFFFF for j in range(128)]
    assert len(set(keys)) == 128
    import random as _r
    rr = _r.Random(7)
    while True:
        magic = rr.getrandbits(32) | 1
        slots = [((kk * magic) & 0xFFFFFFFF) >> 21 for kk in keys]             # 11-bit slot
        if len(set(slots)) == 128:
            break
    table = [0] * 2048
    for j, sl in enumerate(slots):
        table[sl] = j
    o.append("#define BJJ_TS_HASH_MAGIC 0x%08xu" % magic)
    o.append("#define BJJ_K_TS_HASH { " + ",".join(str(v) for v in table) + " }")
    o.append("#define BJJ_K_FINV     %s  // 1/sqrt(-A), Montgomery" % limbs32(mont(inv(f))))
    # arithmetic mod l (scalar side of sign, lib.rs:328, 335-339): Montgomery radix 2^261 as well
    Lm = SUBORDER
    o.append("#define BJJ_L_NINV29   0x%08xu  // -l^-1 mod 2^29" % ((-pow(Lm, -1, 1 << 29)) % (1 << 29)))
    o.append("#define BJJ_K_L_R1     %s  // PLAIN 2^261 mod l" % limbs32(R % Lm))
    o.append("#define BJJ_K_L_R2     %s  // PLAIN 2^522 mod l" % limbs32(R * R % Lm))
    o.append("#define BJJ_K_ORDER    %s  // PLAIN integer 8*l" % limbs32(ORDER))
    o.append("#define BJJ_K_ORDER2   %s  // PLAIN 2*8*l" % limbs32(2 * ORDER))
    o.append("#define BJJ_K_ORDER4   %s  // PLAIN 4*8*l" % limbs32(4 * ORDER))
    o.append("#define BJJ_K_L        %s  // PLAIN l" % limbs32(SUBORDER))
    o.append("#define BJJ_K_L2       %s  // PLAIN 2*l" % limbs32(2 * SUBORDER))
    o.append("#define BJJ_K_L4       %s  // PLAIN 4*l" % limbs32(4 * SUBORDER))
    CF, KP, SP, AL = poseidon_sparse_constants(C, M)
    import random
    rnd = random.Random(2024)
    for _ in range(6):
        ins = [rnd.randrange(Q) for _ in range(5)]
        assert poseidon_sparse(CF, KP, SP, AL, M, ins) == poseidon_plain(C, M, ins), "sparse Poseidon is not equivalent"
    CAB = poseidon_pair_constants(SP)
    for _ in range(6):
        ins = [rnd.randrange(Q) for _ in range(5)]
        assert poseidon_sparse_paired(CF, KP, SP, AL, CAB, M, ins) == poseidon_plain(C, M, ins), "paired form is not equivalent"
    o.append("#define BJJ_K_POSEIDON_C { /* plain form, 68 x 6: kept for tests / documentation */ \\")
    for v in C:
        o.append("  %s, \\" % limbs32(mont(v)))
    o.append("}")
    o.append("#define BJJ_K_POSEIDON_CF { /* full-round constants: rounds 0-3, then 64-67 (64 adjusted) */ \\")
    for v in CF:
        o.append("  %s, \\" % limbs32(mont(v)))
    o.append("}")
    o.append("#define BJJ_K_POSEIDON_KP { /* scalar constant of each partial round */ \\")
    for v in KP:
        o.append("  %s, \\" % limbs32(mont(v)))
    o.append("}")
    o.append("#define BJJ_K_POSEIDON_CAB { /* per PAIR of partial rounds (2p, 2p+1): sum_j v_b[j] what_a[j] */ \\")
    for v in CAB:
        o.append("  %s, \\" % limbs32(mont(v)))
    o.append("}")
    o.append("#define BJJ_K_POSEIDON_SP { /* per partial round: m00, v[5], what[5] */ \\")
    for row in SP:
        for v in row:
            o.append("  %s, \\" % limbs32(mont(v)))
    o.append("}")
    o.append("#define BJJ_K_POSEIDON_AL { /* 5x5 block applied after the last partial round, row-major */ \\")
    for v in AL:
        o.append("  %s, \\" % limbs32(mont(v)))
    o.append("}")
    o.append("#define BJJ_K_POSEIDON_M { /* row-major M[i][j] */ \\")
    for i in range(T):
        for j in range(T):
            o.append("  %s, \\" % limbs32(mont(M[i][j])))
    o.append("}")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bjj_constants.inc")
    with open(path, "w") as fh:
        fh.write("\n".join(o) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    sys.exit(main())
