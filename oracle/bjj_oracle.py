"""
TEST INFRASTRUCTURE ONLY -- pure-Python-int oracle for the BabyJubJub hot path.

This file restates, with arbitrary-precision Python ints, the algorithm of the
reference crate arnaucube/babyjubjub-rs v0.0.11 for the one path this repo
accelerates (Fr arithmetic -> PointProjective::add -> Point::mul_scalar ->
Poseidon t=6 -> verify).  It is the *slow* oracle: used for small cases, for
pinning the C oracle (oracle/bjj_ref.c) and for generating tests/golden/.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything from oracle/.  The product (libbjj_hip.so) never does.

Parity status: PINNED.  Every known-answer test the reference holds for this
path is reproduced by tests/test_oracle_kat.py (reference src/lib.rs:421-552,
689-738).  The Poseidon constants live in the third-party crate poseidon-rs
0.0.8 (Cargo.toml:20), which is not vendored under /root/reference; they are
regenerated here with the Poseidon paper's Grain-LFSR procedure and pinned by
the reference's own verify() vector (src/lib.rs:689-738) plus the public
circomlib Poseidon values.

Reference citations use `lib.rs:N` for /root/reference/src/lib.rs line N.
"""

# ---------------------------------------------------------------------------
# Constants (lib.rs:28-60)
# ---------------------------------------------------------------------------
Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # lib.rs:33-36
A = 168700  # lib.rs:30
D = 168696  # lib.rs:28
B8 = (
    5299619240641551281634865583518297030282874472190772894086521144482721001553,
    16950150798460657717958625567821834550301663161624707787222815936182638968203,
)  # lib.rs:37-46
ORDER = 21888242871839275222246405745257275088614511777268538073601725287587578984328  # lib.rs:47-50
SUBORDER = ORDER >> 3  # lib.rs:53-58


def finv(a):
    return pow(a % Q, Q - 2, Q)


# ---------------------------------------------------------------------------
# PointProjective (lib.rs:62-132)
# ---------------------------------------------------------------------------
def proj_add(p, q):
    """PointProjective::add, add-2008-bbjlp, exact op order of lib.rs:88-131."""
    x1, y1, z1 = p
    x2, y2, z2 = q
    a = z1 * z2 % Q            # lib.rs:91-92
    b = a * a % Q              # lib.rs:93-94
    c = x1 * x2 % Q            # lib.rs:95-96
    d = y1 * y2 % Q            # lib.rs:97-98
    e = D * c % Q              # lib.rs:99-100
    e = e * d % Q              # lib.rs:101
    f = (b - e) % Q            # lib.rs:102-103
    g = (b + e) % Q            # lib.rs:104-105
    x1y1 = (x1 + y1) % Q       # lib.rs:106-107
    x2y2 = (x2 + y2) % Q       # lib.rs:108-109
    aux = x1y1 * x2y2 % Q      # lib.rs:110-111
    aux = (aux - c) % Q        # lib.rs:112
    aux = (aux - d) % Q        # lib.rs:113
    x3 = a * f % Q             # lib.rs:114-115
    x3 = x3 * aux % Q          # lib.rs:116
    ac = A * c % Q             # lib.rs:117-118
    dac = (d - ac) % Q         # lib.rs:119-120
    y3 = a * g % Q             # lib.rs:121-122
    y3 = y3 * dac % Q          # lib.rs:123
    z3 = f * g % Q             # lib.rs:124-125
    return (x3, y3, z3)


def proj_affine(p):
    """PointProjective::affine, lib.rs:70-85 (z == 0 -> (0, 0))."""
    x, y, z = p
    if z % Q == 0:
        return (0, 0)
    zi = finv(z)
    return (x * zi % Q, y * zi % Q)


# ---------------------------------------------------------------------------
# Point (lib.rs:134-190)
# ---------------------------------------------------------------------------
def mul_scalar(pt, n):
    """Point::mul_scalar, lib.rs:149-164: LSB-first double-and-add over
    n.bits() bits, sign of n dropped (to_bytes_le discards it), doubling done
    with the unified add, one affine() at the end."""
    n = abs(n)
    r = (0, 1, 1)
    exp = (pt[0] % Q, pt[1] % Q, 1)
    for i in range(n.bit_length()):
        if (n >> i) & 1:
            r = proj_add(r, exp)
        exp = proj_add(exp, exp)
    return proj_affine(r)


def on_curve(pt):
    x, y = pt
    return (A * x * x + y * y - 1 - D * x * x * y * y) % Q == 0


def compress(pt):
    """Point::compress, lib.rs:166-178 (used only for the G4 golden vector)."""
    x, y = pt
    b = bytearray(y.to_bytes(32, "little"))
    if x > (Q >> 1):
        b[31] |= 0x80
    return bytes(b)


# ---------------------------------------------------------------------------
# Point / signature codec (lib.rs:166-178, 192-224, 245-268; utils.rs:7-29, 109-160, 215-223)
# ---------------------------------------------------------------------------
def legendre_symbol(a):
    """utils.rs:215-223: -1 for a non-residue, 1 otherwise (also for 0)."""
    return -1 if pow(a % Q, (Q - 1) >> 1, Q) == Q - 1 else 1


def modsqrt(a):
    """utils.rs:109-160 (Tonelli-Shanks, Go style).  None where the reference returns Err:
    non-residue or a == 0.  (Which of the two roots comes out does not matter to
    decompress_point: the sign rule of lib.rs:217-219 picks by the sign bit.)"""
    a %= Q
    if legendre_symbol(a) != 1 or a == 0:
        return None
    s, e = Q - 1, 0
    while s % 2 == 0:
        s >>= 1
        e += 1
    n = 2
    while legendre_symbol(n) != -1:
        n += 1
    y, b, g, r = pow(a, (s + 1) >> 1, Q), pow(a, s, Q), pow(n, s, Q), e
    while True:
        t, m = b, 0
        while t != 1:
            t = t * t % Q
            m += 1
        if m == 0:
            return y
        t = pow(g, 1 << (r - m - 1), Q)
        g = pow(g, 1 << (r - m), Q)
        y = y * t % Q
        b = b * g % Q
        r = m


def decompress_point(bb):
    """lib.rs:192-224.  Returns (x, y) or None where the reference returns Err."""
    b = bytearray(bb)
    sign = bool(b[31] & 0x80)
    b[31] &= 0x7F
    y = int.from_bytes(bytes(b), "little")
    if y >= Q:                                     # lib.rs:201-203
        return None
    den_in = (A - (D * (y * y)) % Q) % Q           # lib.rs:207-213
    if den_in == 0:
        return None                                # modinv(0) -> Err (utils.rs:13-15); unreachable: A/D is a non-residue
    x2 = ((1 - (y * y) % Q) * finv(den_in)) % Q    # lib.rs:214
    x = modsqrt(x2)                                # lib.rs:215
    if x is None:
        return None
    if (sign and x <= (Q >> 1)) or ((not sign) and x > (Q >> 1)):   # lib.rs:217-219
        x = (-x) % Q
    return (x % Q, y)


def compress_signature(r_b8, s):
    """Signature::compress, lib.rs:245-258 (s must fit 32 bytes there as well)."""
    return compress(r_b8) + int(s).to_bytes(32, "little")


def decompress_signature(b):
    """lib.rs:260-268: (R, s) or None if R does not decompress; s is taken as-is."""
    r = decompress_point(bytes(b[:32]))
    if r is None:
        return None
    return r, int.from_bytes(bytes(b[32:64]), "little")


# ---------------------------------------------------------------------------
# Poseidon (third-party poseidon-rs 0.0.8)
# ---------------------------------------------------------------------------
_RP_TABLE = [56, 57, 56, 60, 60, 63, 64, 63]  # t = 2..9
_RF = 8


class _Grain:
    """Grain LFSR of the Poseidon reference parameter generator."""

    def __init__(self, t, rp, n=254, field=1, sbox=0, rf=_RF):
        bits = []

        def put(v, w):
            bits.extend(((v >> (w - 1 - i)) & 1) for i in range(w))

        put(field, 2)
        put(sbox, 4)
        put(n, 12)
        put(t, 12)
        put(rf, 10)
        put(rp, 10)
        bits.extend([1] * 30)
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        while True:
            b1 = self._step()
            b2 = self._step()
            if b1:
                return b2

    def word(self, n=254):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


_POSEIDON_CACHE = {}


def poseidon_params(t):
    """(C, M): round constants C[68*t] (rejection-sampled < Q) and the Cauchy
    MDS matrix M[i][j] = 1/(x_i + y_j) (samples reduced mod Q)."""
    if t in _POSEIDON_CACHE:
        return _POSEIDON_CACHE[t]
    rp = _RP_TABLE[t - 2]
    g = _Grain(t, rp)
    C = []
    while len(C) < (_RF + rp) * t:
        v = g.word()
        if v < Q:
            C.append(v)
    xs = [g.word() % Q for _ in range(t)]
    ys = [g.word() % Q for _ in range(t)]
    M = [[finv(xs[i] + ys[j]) for j in range(t)] for i in range(t)]
    _POSEIDON_CACHE[t] = (C, M, rp)
    return _POSEIDON_CACHE[t]


def poseidon(inputs):
    """Poseidon::hash (poseidon-rs 0.0.8) as called at lib.rs:400-404:
    state = [0, in...]; (RF + RP) rounds of ark, x^5 s-box (all / first
    element), state <- M . state; returns state[0]."""
    t = len(inputs) + 1
    C, M, rp = poseidon_params(t)
    st = [0] + [v % Q for v in inputs]
    nr = _RF + rp
    for r in range(nr):
        st = [(st[j] + C[r * t + j]) % Q for j in range(t)]
        if r < _RF // 2 or r >= _RF // 2 + rp:
            st = [pow(v, 5, Q) for v in st]
        else:
            st[0] = pow(st[0], 5, Q)
        st = [sum(M[i][j] * st[j] for j in range(t)) % Q for i in range(t)]
    return st[0]


# ---------------------------------------------------------------------------
# Blake-512 (the original BLAKE, SHA-3 finalist -- NOT BLAKE2), third-party blake-hash 0.4.0 /
# blake 2.0.1 (Cargo.toml:17-18), called through blh() at lib.rs:226-237.  Restated from the
# published specification; pinned by the digest KAT of lib.rs:695-696.
# ---------------------------------------------------------------------------
_BLAKE_IV = [0x6A09E667F3BCC908, 0xBB67AE8584CAA73B, 0x3C6EF372FE94F82B, 0xA54FF53A5F1D36F1,
             0x510E527FADE682D1, 0x9B05688C2B3E6C1F, 0x1F83D9ABFB41BD6B, 0x5BE0CD19137E2179]
_BLAKE_C = [0x243F6A8885A308D3, 0x13198A2E03707344, 0xA4093822299F31D0, 0x082EFA98EC4E6C89,
            0x452821E638D01377, 0xBE5466CF34E90C6C, 0xC0AC29B7C97C50DD, 0x3F84D5B5B5470917,
            0x9216D5D98979FB1B, 0xD1310BA698DFB5AC, 0x2FFD72DBD01ADFB7, 0xB8E1AFED6A267E96,
            0xBA7C9045F12C7F99, 0x24A19947B3916CF7, 0x0801F2E2858EFC16, 0x636920D871574E69]
_BLAKE_SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]


def _ror64(x, n):
    return ((x >> n) | (x << (64 - n))) & MASK64_


MASK64_ = (1 << 64) - 1


def _blake_compress(h, block, t):
    m = [int.from_bytes(block[8 * i:8 * i + 8], "big") for i in range(16)]
    c = _BLAKE_C
    v = h[:] + [c[0], c[1], c[2], c[3], c[4] ^ (t & MASK64_), c[5] ^ (t & MASK64_), c[6] ^ (t >> 64), c[7] ^ (t >> 64)]

    def g(a, b, cc, d, r, i):
        j, k = _BLAKE_SIGMA[r % 10][2 * i], _BLAKE_SIGMA[r % 10][2 * i + 1]
        v[a] = (v[a] + v[b] + (m[j] ^ c[k])) & MASK64_
        v[d] = _ror64(v[d] ^ v[a], 32)
        v[cc] = (v[cc] + v[d]) & MASK64_
        v[b] = _ror64(v[b] ^ v[cc], 25)
        v[a] = (v[a] + v[b] + (m[k] ^ c[j])) & MASK64_
        v[d] = _ror64(v[d] ^ v[a], 16)
        v[cc] = (v[cc] + v[d]) & MASK64_
        v[b] = _ror64(v[b] ^ v[cc], 11)

    for r in range(16):
        g(0, 4, 8, 12, r, 0); g(1, 5, 9, 13, r, 1); g(2, 6, 10, 14, r, 2); g(3, 7, 11, 15, r, 3)
        g(0, 5, 10, 15, r, 4); g(1, 6, 11, 12, r, 5); g(2, 7, 8, 13, r, 6); g(3, 4, 9, 14, r, 7)
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def blake512(msg):
    """64-byte digest.  Padding: 1-bit, zeros, 1-bit, 128-bit big-endian bit length; the counter t is the
    number of message bits hashed up to and including the block (0 for a block of pure padding)."""
    msg = bytes(msg)
    h = _BLAKE_IV[:]
    nbits = len(msg) * 8
    rem = len(msg) % 128
    padlen = (111 - rem) if rem <= 111 else (239 - rem)
    p = bytearray(msg) + (b"\x81" if padlen == 0 else b"\x80" + b"\x00" * (padlen - 1) + b"\x01") + nbits.to_bytes(16, "big")
    for blk in range(len(p) // 128):
        t = min(nbits, (blk + 1) * 1024) if blk * 1024 < nbits else 0
        h = _blake_compress(h, bytes(p[128 * blk:128 * blk + 128]), t)
    return b"".join(x.to_bytes(8, "big") for x in h)


# ---------------------------------------------------------------------------
# PrivateKey (lib.rs:270-342)
# ---------------------------------------------------------------------------
def scalar_key(key):
    """PrivateKey::scalar_key, lib.rs:284-302: Blake-512, RFC 8032 pruning of the low half, >> 3."""
    h = bytearray(blake512(bytes(key))[:32])
    h[0] &= 0xF8
    h[31] &= 0x7F
    h[31] |= 0x40
    return int.from_bytes(bytes(h), "little") >> 3


def public(key):
    """PrivateKey::public, lib.rs:304-306."""
    return mul_scalar(B8, scalar_key(key))


def sign(key, msg):
    """PrivateKey::sign, lib.rs:308-342 -> (R, s) or None where the reference returns Err (msg > Q)."""
    if msg > Q:                                              # lib.rs:309-311
        return None
    h = blake512(bytes(key))                                 # lib.rs:316
    msg32 = int(msg).to_bytes(32, "little")                  # lib.rs:318-320
    r = int.from_bytes(blake512(h[32:64] + msg32), "little") % SUBORDER   # lib.rs:323-328
    r_b8 = mul_scalar(B8, r)                                 # lib.rs:329
    a = public(key)                                          # lib.rs:330
    hm = poseidon([r_b8[0], r_b8[1], a[0], a[1], msg % Q])   # lib.rs:332-333
    s = (r + hm * (scalar_key(key) << 3)) % SUBORDER         # lib.rs:335-339
    return r_b8, s


# ---------------------------------------------------------------------------
# verify (lib.rs:395-412)
# ---------------------------------------------------------------------------
def verify(pk, sig_r, sig_s, msg):
    """verify(pk, Signature{r_b8, s}, msg) -> bool, lib.rs:395-412."""
    if msg > Q:                                   # lib.rs:396-398
        return False
    msg_fr = msg % Q                              # lib.rs:399 (from_str wraps msg == Q to 0)
    hm = poseidon([sig_r[0], sig_r[1], pk[0], pk[1], msg_fr])  # lib.rs:400-404
    l = mul_scalar(B8, sig_s)                     # lib.rs:405
    t = mul_scalar(pk, 8 * hm)                    # lib.rs:410
    r = proj_add((sig_r[0] % Q, sig_r[1] % Q, 1), (t[0], t[1], 1))  # lib.rs:407-410
    ra = proj_affine(r)                           # lib.rs:411
    return l[0] == ra[0] and l[1] == ra[1]


def schnorr_hash(pk, msg, c):
    """schnorr_hash(pk, msg, c), lib.rs:364-373: Poseidon(pk.x, pk.y, c.x, c.y, msg); None for Err (msg > Q)."""
    if msg > Q:
        return None
    return poseidon([pk[0], pk[1], c[0], c[1], msg % Q])


def verify_schnorr(pk, m, r, s):
    """verify_schnorr(pk, m, r, s) -> Result<bool, String>, lib.rs:375-385; None for Err."""
    sg = mul_scalar(B8, s)                                   # lib.rs:377
    h = schnorr_hash(pk, m, r)                               # lib.rs:380
    if h is None:
        return None
    pk_h = mul_scalar(pk, h)                                 # lib.rs:381
    right = proj_add((r[0] % Q, r[1] % Q, 1), (pk_h[0], pk_h[1], 1))   # lib.rs:382
    ra = proj_affine(right)
    return sg[0] == ra[0] and sg[1] == ra[1]                 # lib.rs:384


def sign_schnorr_with_nonce(key, m, k):
    """PrivateKey::sign_schnorr (lib.rs:345-361) with the random nonce k supplied by the caller
    (the reference draws 1024 random bits, lib.rs:347-348).  Returns (r, s) with s = k + scalar_key*h
    UNREDUCED as in the reference, or None for Err."""
    r = mul_scalar(B8, k)
    h = schnorr_hash(public(key), m, r)
    if h is None:
        return None
    return r, k + scalar_key(key) * h


def sign_with_scalars(k, rho, msg):
    """Algebraic equivalent of PrivateKey::sign (lib.rs:308-342) given the
    already-derived secret scalar k (= scalar_key() << 3 >> 3 ... i.e. the
    value whose 8x is used at lib.rs:335) and nonce rho: A = k*B8 ... NOTE the
    reference computes public() = B8 * scalar_key() (lib.rs:304-306) and
    S = r + hm * (scalar_key() << 3) (lib.rs:335-339).  Here `k` plays the
    role of scalar_key().  Used only to synthesise valid signatures for tests
    and the benchmark (SURVEY.md 8d cfg 4)."""
    Apt = mul_scalar(B8, k)
    R = mul_scalar(B8, rho)
    hm = poseidon([R[0], R[1], Apt[0], Apt[1], msg % Q])
    S = (rho + hm * (k << 3)) % SUBORDER
    return Apt, R, S


# ---------------------------------------------------------------------------
# SplitMix64: the one synthetic-input generator
# ---------------------------------------------------------------------------
MASK64 = (1 << 64) - 1
SEED_SCALARS = 0x424A4A5F5343414C
SEED_POINTS = 0x424A4A5F504F494E
SEED_MSGS = 0x424A4A5F4D534753
SEED_KEYS = 0x424A4A5F4B455953
SEED_NONCES = 0x424A4A5F4E4F4E43
SEED_BAD = 0x424A4A5F42414421


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & MASK64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def u256(self):
        """4 x u64, little-endian limbs -> one 256-bit integer."""
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v


# order-8 torsion point (SURVEY.md 8d cfg 3)
T8 = (
    4342719913949491028786768530115087822524712248835451589697801404893164183326,
    4826523245007015323400664741523384119579596407052839571721035538011798951543,
)


def to_le32(v):
    return int(v).to_bytes(32, "little")


def from_le32(b):
    return int.from_bytes(bytes(b), "little")
