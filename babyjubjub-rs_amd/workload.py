"""Synthetic workload + sharding helpers shared by bench.py and the tests.

Inputs follow SURVEY.md 8(d): one SplitMix64 generator, 4 x u64 little-endian per
32-byte value, fixed seeds per stream.  Generation is vectorised with numpy (u64
wrap-around arithmetic), so 1M-item batches take milliseconds on the host.
"""
import numpy as np

SEED_SCALARS = 0x424A4A5F5343414C
SEED_POINTS = 0x424A4A5F504F494E
SEED_MSGS = 0x424A4A5F4D534753
SEED_KEYS = 0x424A4A5F4B455953
SEED_NONCES = 0x424A4A5F4E4F4E43
SEED_BAD = 0x424A4A5F42414421

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, count, offset=0):
    """`count` outputs of SplitMix64(seed), starting at output index `offset`."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_u256(seed, n, offset=0, top_bits_cleared=0):
    """n values of 32 bytes (uint8 array of shape (n, 32)); item i uses outputs 4i..4i+3."""
    w = splitmix64(seed, 4 * n, 4 * offset).reshape(n, 4)
    if top_bits_cleared:
        w[:, 3] &= np.uint64((1 << (64 - top_bits_cleared)) - 1)
    return w.astype("<u8").view(np.uint8).reshape(n, 32)


def scalars_254(n, offset=0):
    """cfg 1-3 scalars: uniform in [0, 2^254)."""
    return random_u256(SEED_SCALARS, n, offset, top_bits_cleared=2)


def shard_bounds(n, world_size, rank):
    """Contiguous block partition of n items: [lo, hi) for `rank`."""
    per = (n + world_size - 1) // world_size
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


# ---- arithmetic mod l on arrays of little-endian 256-bit integers (host-side signer math) ----
L_ORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041


def to_ints(a):
    b = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
    return [int.from_bytes(r.tobytes(), "little") for r in b]


def from_ints(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), np.uint8).reshape(-1, 32).copy()


# ---- cfg-4 workload: valid EdDSA-Poseidon signatures, 1 in 64 corrupted ----------------------
def make_signatures(fixed_base, poseidon5, n, offset=0):
    """A = k*B8, R = rho*B8, S = rho + 8*hm*k mod l -- algebraically what PrivateKey::sign produces (src/lib.rs:335-339).
    `fixed_base` / `poseidon5` are callables on (n, 32) / (n, 160) byte arrays (the GPU library or the oracle)."""
    k = [v % L_ORDER for v in to_ints(random_u256(SEED_KEYS, n, offset))]
    rho = [v % L_ORDER for v in to_ints(random_u256(SEED_NONCES, n, offset))]
    msg = random_u256(SEED_MSGS, n, offset, top_bits_cleared=3)  # < 2^253 < Q
    A = fixed_base(from_ints(k))
    R = fixed_base(from_ints(rho))
    hm = to_ints(poseidon5(np.concatenate([R, A, msg], axis=1)))
    S = from_ints([(rho[i] + 8 * hm[i] * k[i]) % L_ORDER for i in range(n)])
    return A, R, S, msg


def corrupt(A, R, S, msg, n, offset=0):
    """1 item in 64 gets one seeded bit flipped in S, msg, R.y or A.x (in place); returns the bad mask (numpy bool).
    A, R: (n, 64), S, msg: (n, 32) uint8 -- numpy arrays or torch tensors (on any device): vectorised, so that the
    2^24-item batch of BASELINE cfg 5 is corrupted in place in HBM."""
    r = splitmix64(SEED_BAD, n, offset)
    bad = (r & np.uint64(63)) == 0
    idx = np.nonzero(bad)[0]
    which = ((r[idx] >> np.uint64(6)) & np.uint64(3)).astype(np.int64)
    bit = ((r[idx] >> np.uint64(8)) % np.uint64(250)).astype(np.int64)
    for t, (arr, col0) in enumerate(((S, 0), (msg, 0), (R, 32), (A, 0))):
        sel = which == t
        rows, cols = idx[sel], bit[sel] // 8 + col0
        mask = (1 << (bit[sel] % 8)).astype(np.uint8)
        if isinstance(arr, np.ndarray):
            arr[rows, cols] ^= mask
        else:  # torch tensor
            import torch
            ri = torch.from_numpy(rows).to(arr.device)
            ci = torch.from_numpy(cols).to(arr.device)
            arr[ri, ci] = arr[ri, ci] ^ torch.from_numpy(mask).to(arr.device)
    return bad


def piece_bounds(cnt, pieces=4, min_piece=1 << 15):
    """How a peer's block of `cnt` items is cut for the pipelined scatter -> kernels -> gather schedule: at most `pieces`
    pieces, none smaller than min_piece items (except that a block below 2 * min_piece stays whole), every piece a multiple
    of 64 items but the last.  The SAME geometry as bjj_multi_* inside libbjj_hip.so (csrc/bjj_multi.inc: chunk geometry).
    Returns [(lo, hi), ...] relative to the block."""
    if cnt <= 0:
        return []
    c = max(1, min(int(pieces), cnt // min_piece if min_piece else int(pieces)))
    csz = (-(-cnt // c) + 63) & ~63
    return [(lo, min(cnt, lo + csz)) for lo in range(0, cnt, csz)]

