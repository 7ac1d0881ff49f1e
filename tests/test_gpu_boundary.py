"""Round-2 boundary additions through the C ABI, against the oracle: raw PointProjective::add / affine for any z
(src/lib.rs:88-131, 70-85, chained as in the reference's own test :513-516), scalars wider than 256 bits
(`n: &BigInt`, src/lib.rs:149, 156-157), the default / automatic table width, ordering of calls on different
streams, and the multi-GPU handle (bjj_multi_*: partition, host-pointer form, RCCL scatter/gather form).
Needs a real MI355X: run with `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest

from conftest import pack, unpack

pytestmark = pytest.mark.gpu

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617
L = 2736030358979909402780800718157159386076813972158567259200215660948447373041
B8 = (5299619240641551281634865583518297030282874472190772894086521144482721001553,
      16950150798460657717958625567821834550301663161624707787222815936182638968203)


def _proj_add_oracle(oracle, p, q):
    out = np.empty_like(p)
    for i in range(p.shape[0]):
        oracle.lib.bjjref_proj_add(oracle._p(p[i]), oracle._p(q[i]), oracle._p(out[i]))
    return out


def _proj_affine_oracle(oracle, p):
    out = np.empty((p.shape[0], 64), np.uint8)
    for i in range(p.shape[0]):
        oracle.lib.bjjref_proj_affine(oracle._p(p[i]), oracle._p(out[i]))
    return out


def _mul_scalar_oracle(oracle, pts, scalars, nbytes):
    out = np.empty((pts.shape[0], 64), np.uint8)
    for i in range(pts.shape[0]):
        oracle.lib.bjjref_mul_scalar(oracle._p(pts[i]), oracle._p(scalars[i]), ctypes.c_size_t(nbytes), oracle._p(out[i]))
    return out


# ---------------------------------------------------------------- PointProjective::add / affine, any z
def test_proj_add_and_affine_general_z(gpu_ctx, oracle):
    from babyjubjub_rs_amd import workload as w
    n = 777
    p = w.random_u256(0x5052, 3 * n, 0, top_bits_cleared=3).reshape(n, 96).copy()   # arbitrary (x, y, z): off the curve too
    q = w.random_u256(0x5053, 3 * n, 0, top_bits_cleared=3).reshape(n, 96).copy()
    p[5, 64:] = 0                       # z == 0
    q[6, 64:] = 0
    p[7] = 0                            # all zero
    p[8, :32] = 0xFF                    # x >= r: reduced like Fr::from_repr would refuse -- the ABI reduces mod r
    # on-curve operands with z != 1: (x*z, y*z, z)
    sc = w.scalars_254(64)
    aff = gpu_ctx.mul_fixed_base(sc)
    for i in range(64):
        x, y = unpack(aff[i], 2)[0]
        z = int.from_bytes(p[100 + i, 64:].tobytes(), "little") % Q or 1
        p[100 + i] = pack([(x * z % Q, y * z % Q, z)])
    got = gpu_ctx.proj_add(p, q)
    want = _proj_add_oracle(oracle, p, q)
    assert (got == want).all()
    ga = gpu_ctx.proj_affine(np.concatenate([p, got]))
    assert (ga == _proj_affine_oracle(oracle, np.concatenate([p, got]))).all()
    assert (ga[5] == 0).all() and (ga[7] == 0).all()           # z == 0 -> (0, 0), src/lib.rs:71-76
    assert (ga[100:164] == aff).all()                          # (xz, yz, z) -> (x, y)


def test_reference_chained_add_equals_mul_scalar_3(gpu_ctx, golden):
    """src/lib.rs:513-516: p.mul_scalar(3) == p.projective().add(&p.projective()).add(&p.projective()).affine() -- the
    second add has a z != 1 operand, which is why the raw (x, y, z) entry point exists."""
    import babyjubjub_rs_amd as bjj
    b = golden["reference_kats"]["bench_inputs"]
    p = bjj.Point(int(b["p"][0], 16) if isinstance(b["p"][0], str) else b["p"][0],
                  int(b["p"][1], 16) if isinstance(b["p"][1], str) else b["p"][1])
    two = p.projective().add(p.projective(), ctx=gpu_ctx)
    assert two.z != 1
    three = two.add(p.projective(), ctx=gpu_ctx)
    r = p.mul_scalar(3, ctx=gpu_ctx)
    assert three.affine(ctx=gpu_ctx).equals(r)
    assert bjj.PointProjective(1, 2, 0).affine(ctx=gpu_ctx).equals(bjj.Point(0, 0))


# ---------------------------------------------------------------- scalars wider than 256 bits
@pytest.mark.parametrize("nbytes", [64, 128, 352])
def test_wide_scalars_on_and_off_curve(gpu_ctx, oracle, nbytes):
    from babyjubjub_rs_amd import workload as w
    n = 130
    k = nbytes // 32
    sc = w.random_u256(0x5749 + nbytes, k * n).reshape(n, nbytes).copy()
    sc[0] = 0                                   # zero scalar -> (0, 1)
    sc[1] = 0; sc[1, 0] = 1                     # one
    sc[2, 32:] = 0                              # fits in 256 bits
    sc[3] = 0xFF                                # all ones
    pts = gpu_ctx.mul_fixed_base(w.scalars_254(n, offset=77)).copy()
    pts[10::13, 0] ^= 1                         # off the curve: the device replays all n.bits() bits of src/lib.rs:157-162
    pts[4] = pack([(0, 1)])                     # identity
    pts[5] = pack([(0, Q - 1)])                 # order 2
    got = gpu_ctx.mul_var_base_wide(pts, sc, nbytes)
    want = _mul_scalar_oracle(oracle, pts, sc, nbytes)
    assert (got == want).all(), np.nonzero((got != want).any(axis=1))[0][:8]
    # the 32-byte record through the wide entry is the plain entry
    assert (gpu_ctx.mul_var_base_wide(pts, sc[:, :32].copy(), 32) == gpu_ctx.mul_var_base(pts, sc[:, :32].copy())).all()


def test_point_mul_scalar_wide_bigint(gpu_ctx, golden):
    import babyjubjub_rs_amd as bjj
    from conftest import ints
    n = (1 << 1023) + 0x1234567 * (1 << 300) + 99
    v = golden["gpu_expected"]["wide_bigint"]                  # tests/golden/make_gpu_expected.py
    assert ints(v["n"]) == n
    p = bjj.Point(*B8)
    r = p.mul_scalar(n, ctx=gpu_ctx)
    assert (r.x, r.y) == ints(v["result"])
    assert p.mul_scalar(n % (8 * L), ctx=gpu_ctx).equals(r)    # the group order is 8l
    assert p.mul_scalar(-n, ctx=gpu_ctx).equals(r)      # sign dropped, src/lib.rs:156
    with pytest.raises(bjj.BjjError):
        gpu_ctx.mul_var_base_wide(pack([B8]), np.zeros(48, np.uint8), 48)      # not a multiple of 32


# ---------------------------------------------------------------- table width: default is modest, wide is opt-in
def test_default_window_is_modest(gpu_ctx, oracle):
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    ctx = bjj.Context(0)
    try:
        info = ctx.info()
        assert info.window_bits == 23 and info.table_bytes == 11 * ((1 << 22) + 1) * 128 < 6 << 30
        assert info.init_ms > 0
        sc = w.scalars_254(500, offset=31)
        assert (ctx.mul_fixed_base(sc) == oracle.mul_fixed_base(sc)).all()
    finally:
        ctx.close()


# ---------------------------------------------------------------- calls on different streams are ordered by the context
def test_calls_on_two_streams_share_scratch_safely(gpu_ctx, oracle):
    """verify (stream A) and variable-base (stream B) both use the context's work list / per-lane tables; the second call is
    enqueued while the first is still running.  The context inserts the cross-stream dependency itself."""
    import torch
    from babyjubjub_rs_amd import workload as w
    dev = torch.device("cuda", 0)
    n = 1 << 16
    A, R, S, msg = w.make_signatures(gpu_ctx.mul_fixed_base, gpu_ctx.poseidon5, n)
    bad = w.corrupt(A, R, S, msg, n)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)  # noqa: E731
    d_pk, d_r, d_s, d_m = up(A), up(R), up(S), up(msg)
    sc = w.scalars_254(n, offset=5)
    d_sc = up(sc)
    d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    d_out2 = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        gpu_ctx.eddsa_verify_dev(d_pk.data_ptr(), d_r.data_ptr(), d_s.data_ptr(), d_m.data_ptr(), n, d_ok.data_ptr(), sa.cuda_stream)
        gpu_ctx.mul_var_base_dev(d_pk.data_ptr(), d_sc.data_ptr(), n, d_out.data_ptr(), sb.cuda_stream)
        gpu_ctx.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out2.data_ptr(), sa.cuda_stream)
    gpu_ctx.sync()      # also waits for work enqueued on a caller's stream
    torch.cuda.synchronize()
    assert (d_ok.cpu().numpy() == (~bad).astype(np.uint8)).all()
    idx = np.arange(0, n, 61)
    assert (d_out.cpu().numpy().reshape(n, 64)[idx] == oracle.mul_var_base(A[idx], sc[idx])).all()
    assert (d_out2.cpu().numpy().reshape(n, 64)[idx] == oracle.mul_fixed_base(sc[idx])).all()


def test_overlapping_launches_on_two_and_three_streams_use_separate_scratch_sets(oracle):
    """The context keeps two scratch sets (per-lane tables, work lists, epilogue scratch): launches that alternate over two
    streams get one set each and run CONCURRENTLY (the head of one fills the tail of the other) -- what bench.py's two-stream
    protocol times.  Different batches per stream, every verdict / point checked, several rounds so that each set is reused
    while the other stream's launch is still running; a third stream has to take over the least recently used set and wait
    for it.  Runs on its own context so that the scratch accounting starts from zero."""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    dev = torch.device("cuda", 0)
    ctx = bjj.Context(0, 16)
    try:
        n = (1 << 17) + 77                                  # ragged: partially filled last wave and last chunk
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)  # noqa: E731
        batches = []
        for b in range(3):
            keys = w.random_u256(w.SEED_KEYS, n, offset=b * n)
            msg = w.random_u256(w.SEED_MSGS, n, offset=b * n, top_bits_cleared=3)
            A = ctx.public_keys(keys)
            R, S, okf = ctx.sign(keys, msg)
            assert okf.all()
            bad = w.corrupt(A, R, S, msg, n, offset=b * n)
            sc = w.scalars_254(n, offset=17 + b * n)
            batches.append(dict(A=A, sc=sc, bad=bad, d=[up(A), up(R), up(S), up(msg)], d_sc=up(sc),
                                d_ok=torch.zeros(n, dtype=torch.uint8, device=dev),
                                d_out=torch.zeros(n * 64, dtype=torch.uint8, device=dev),
                                d_fb=torch.zeros(n * 64, dtype=torch.uint8, device=dev)))
        streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
        torch.cuda.synchronize()
        before = ctx.info().scratch_bytes
        for ns in (2, 3):
            for B in batches:
                B["d_ok"].zero_(); B["d_out"].zero_()
            torch.cuda.synchronize()
            for rnd in range(4):
                for b in range(ns):
                    B, st = batches[b], streams[b].cuda_stream
                    ctx.eddsa_verify_dev(*[t.data_ptr() for t in B["d"]], n, B["d_ok"].data_ptr(), st)
                    ctx.mul_var_base_dev(B["d"][0].data_ptr(), B["d_sc"].data_ptr(), n, B["d_out"].data_ptr(), st)
                    ctx.mul_fixed_base_dev(B["d_sc"].data_ptr(), n, B["d_fb"].data_ptr(), st)   # picks its shape per call
            ctx.sync()                                       # waits for every caller stream the context has work on
            for b in range(ns):
                B = batches[b]
                assert (B["d_ok"].cpu().numpy() == (~B["bad"]).astype(np.uint8)).all(), (ns, b)
                idx = np.arange(b, n, 257)
                got = B["d_out"].cpu().numpy().reshape(n, 64)[idx]
                assert (got == oracle.mul_var_base(B["A"][idx], B["sc"][idx])).all(), (ns, b)
                assert (B["d_fb"].cpu().numpy().reshape(n, 64)[idx] == oracle.mul_fixed_base(B["sc"][idx])).all(), (ns, b)
        assert ctx.info().scratch_bytes > before            # the second set came into being with the second stream
        # a single-stream caller never leaves set 0: a fresh context's footprint does not depend on this feature
        c1 = bjj.Context(0, 16)
        try:
            B = batches[0]
            for _ in range(3):
                c1.eddsa_verify_dev(*[t.data_ptr() for t in B["d"]], n, B["d_ok"].data_ptr(), streams[0].cuda_stream)
            c1.sync()
            one_set = c1.info().scratch_bytes
            c1.eddsa_verify_dev(*[t.data_ptr() for t in B["d"]], n, B["d_ok"].data_ptr(), streams[1].cuda_stream)
            c1.sync()
            assert c1.info().scratch_bytes >= 2 * one_set - 64 > one_set
        finally:
            c1.close()
    finally:
        ctx.close()


def test_large_launches_on_two_streams_queue_behind_each_other(oracle):
    """Launches of more than 2^21 items do not share the chip with another launch of the context: they queue behind the other
    scratch set's last call (device-side wait) and run in the form of a launch that runs alone (persistent verify waves, K2
    tiles).  Two streams alternating large and small launches: every verdict of every launch, sampled points, nothing stale."""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    dev = torch.device("cuda", 0)
    ctx = bjj.Context(0, 16)
    try:
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)  # noqa: E731
        sizes = [(1 << 21) + 4099, (1 << 21) + 64, 70001]      # two large batches, one small
        batches = []
        for b, n in enumerate(sizes):
            keys = w.random_u256(w.SEED_KEYS, n, offset=b << 22)
            msg = w.random_u256(w.SEED_MSGS, n, offset=b << 22, top_bits_cleared=3)
            A = ctx.public_keys(keys)
            R, S, okf = ctx.sign(keys, msg)
            assert okf.all()
            bad = w.corrupt(A, R, S, msg, n, offset=b << 22)
            sc = w.scalars_254(n, offset=5 + (b << 22))
            batches.append(dict(n=n, A=A, sc=sc, bad=bad, d=[up(A), up(R), up(S), up(msg)], d_sc=up(sc),
                                d_ok=torch.zeros(n, dtype=torch.uint8, device=dev), d_out=torch.zeros(n * 64, dtype=torch.uint8, device=dev)))
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        order = [(0, 0), (1, 1), (2, 0), (1, 1), (0, 0), (2, 1)]   # (batch, stream): large / large / small / large / large / small
        for rnd in range(2):
            for B in batches:
                B["d_ok"].fill_(7); B["d_out"].zero_()
            torch.cuda.synchronize()
            for b, si in order:
                B, st = batches[b], streams[si].cuda_stream
                ctx.eddsa_verify_dev(*[t.data_ptr() for t in B["d"]], B["n"], B["d_ok"].data_ptr(), st)
                ctx.mul_var_base_dev(B["d"][0].data_ptr(), B["d_sc"].data_ptr(), B["n"], B["d_out"].data_ptr(), st)
            ctx.sync()
            for b, B in enumerate(batches):
                assert (B["d_ok"].cpu().numpy() == (~B["bad"]).astype(np.uint8)).all(), (rnd, b)
                idx = np.arange(b, B["n"], 4099)
                got = B["d_out"].cpu().numpy().reshape(B["n"], 64)[idx]
                assert (got == oracle.mul_var_base(B["A"][idx], B["sc"][idx])).all(), (rnd, b)
    finally:
        ctx.close()


def test_entry_points_leave_the_callers_current_device_alone(oracle):
    """every entry point selects its context's device and restores the calling thread's current device on return"""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    if torch.cuda.device_count() < 2:
        # one GPU: the guard has nothing to restore; still run the calls with the device current
        torch.cuda.set_device(0)
    other = torch.cuda.device_count() - 1
    torch.cuda.set_device(other)
    ctx = bjj.Context(0, 16)
    try:
        sc = w.scalars_254(100, offset=1)
        assert (ctx.mul_fixed_base(sc) == oracle.mul_fixed_base(sc)).all()
        ctx.sync()
        assert torch.cuda.current_device() == other
    finally:
        ctx.close()
        torch.cuda.set_device(0)


# ---------------------------------------------------------------- multi-GPU handle
def _pieces(cnt, chunks, min_chunk):
    """sizes of the pieces bjj_multi_* cuts a peer's block of `cnt` items into (csrc/bjj_multi.inc: chunk geometry)"""
    if cnt == 0:
        return []
    c = max(1, min(chunks, cnt // min_chunk if min_chunk else chunks))
    csz = (-(-cnt // c) + 63) & ~63
    return [min(csz, cnt - lo) for lo in range(0, cnt, csz)]


def _multi(devs):
    import babyjubjub_rs_amd as bjj
    return bjj.MultiContext(devs, 16)       # a small table per device keeps the test light


@pytest.mark.parametrize("n", [1, 64, 4099])
def test_multi_host_form_one_device(oracle, n):
    from babyjubjub_rs_amd import workload as w
    m = _multi([0])
    try:
        assert m.size == 1 and m.device(0) == 0 and m.shard_bounds(n, 0) == (0, n)
        sc = w.scalars_254(n, offset=11)
        fb = m.mul_fixed_base(sc)
        assert (fb == oracle.mul_fixed_base(sc)).all()
        assert (m.mul_var_base(fb, sc) == oracle.mul_var_base(fb, sc)).all()
        A, R, S, msg = w.make_signatures(oracle.mul_fixed_base, oracle.poseidon5, n)
        bad = w.corrupt(A, R, S, msg, n)
        assert (m.eddsa_verify(A, R, S, msg) == (~bad).astype(np.uint8)).all()
    finally:
        m.close()


def test_multi_dev_form_rccl_one_device(oracle):
    """G = 1 through the device-resident entry points: RCCL is loaded (dlopen), ncclCommInitAll creates the clique, the
    grouped ncclScatter / ncclGather of the serial schedule run in place at the root (on the root's transfer stream), the
    kernels run on the caller's buffers."""
    import torch
    from babyjubjub_rs_amd import workload as w
    dev = torch.device("cuda", 0)
    m = _multi([0])
    try:
        n = 5000
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)  # noqa: E731
        sc = w.scalars_254(n, offset=3)
        d_sc = up(sc)
        d_out = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
        fb = d_out.cpu().numpy().reshape(n, 64)
        assert (fb == oracle.mul_fixed_base(sc)).all()
        t = m.last_timing()
        assert t["rccl_version"] > 0 and t["compute_ms"] > 0
        d_out2 = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        m.mul_var_base_dev(d_out.data_ptr(), d_sc.data_ptr(), n, d_out2.data_ptr())
        assert (d_out2.cpu().numpy().reshape(n, 64) == oracle.mul_var_base(fb, sc)).all()
        A, R, S, msg = w.make_signatures(oracle.mul_fixed_base, oracle.poseidon5, n)
        bad = w.corrupt(A, R, S, msg, n)
        d_ok = torch.empty(n, dtype=torch.uint8, device=dev)
        t_A, t_R, t_S, t_m = up(A), up(R), up(S), up(msg)       # keep the tensors alive across the call
        m.eddsa_verify_dev(t_A.data_ptr(), t_R.data_ptr(), t_S.data_ptr(), t_m.data_ptr(), n, d_ok.data_ptr())
        assert (d_ok.cpu().numpy() == (~bad).astype(np.uint8)).all()
    finally:
        m.close()


@pytest.mark.parametrize("g", [2, 3, 8])
def test_multi_block_arithmetic_with_several_contexts_on_one_gpu(oracle, g):
    """G > 1 on the one-GPU box: the handle names device 0 g times (g contexts, g streams, g block buffers), the
    device-resident form moves the blocks with the peer-copy transport (RCCL refuses duplicate devices and says so), the
    host-pointer form runs its g pipeline threads concurrently.  Even, ragged and tiny batches; every output vs the oracle."""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    dev = torch.device("cuda", 0)
    m = bjj.MultiContext([0] * g, 8)
    try:
        assert m.size == g
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)  # noqa: E731
        sc0 = w.scalars_254(16)
        d0, o0 = up(sc0), torch.empty(16 * 64, dtype=torch.uint8, device=dev)
        with pytest.raises(bjj.BjjError, match="distinct devices"):
            m.mul_fixed_base_dev(d0.data_ptr(), 16, o0.data_ptr())          # default transport = RCCL
        m.set_transport("peer")
        for n, chunks in ((g * 512, 1), (g * 512, 4), (g * 512 + 5, 3), (g * 1000 + 77, 16), (3, 4), (1, 4), (g, 2)):
            m.set_chunks(chunks, 64)        # pieces of >= 64 items, so that these small blocks really are cut up
            sc = w.scalars_254(n, offset=7 * n)
            want = oracle.mul_fixed_base(sc)
            assert (m.mul_fixed_base(sc) == want).all()                      # host-pointer form, g threads
            d_sc = up(sc)
            d_out = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
            fb = d_out.cpu().numpy().reshape(n, 64)
            assert (fb == want).all(), n
            d_out2 = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
            m.mul_var_base_dev(d_out.data_ptr(), d_sc.data_ptr(), n, d_out2.data_ptr())
            assert (d_out2.cpu().numpy().reshape(n, 64) == oracle.mul_var_base(fb, sc)).all(), n
            A, R, S, msg = w.make_signatures(oracle.mul_fixed_base, oracle.poseidon5, n, offset=n)
            bad = w.corrupt(A, R, S, msg, n, offset=n)
            if n > 2:
                A[2, 5] ^= 1; bad[2] = True                                  # an off-curve pk: the exact path inside a block
            assert (m.eddsa_verify(A, R, S, msg) == (~bad).astype(np.uint8)).all()
            t_A, t_R, t_S, t_m = up(A), up(R), up(S), up(msg)
            d_ok = torch.full((((n + 15) // 16) * 16,), 7, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            m.eddsa_verify_dev(t_A.data_ptr(), t_R.data_ptr(), t_S.data_ptr(), t_m.data_ptr(), n, d_ok.data_ptr())
            got = d_ok.cpu().numpy()
            assert (got[:n] == (~bad).astype(np.uint8)).all() and (got[n:] == 7).all(), n   # nothing written past n
            t = m.last_timing()
            per = -(-n // g)
            want_pieces = len(_pieces(min(per, max(0, n - per)), chunks, 64)) or 1      # rank 1 holds the largest peer block
            assert t["chunks"] == want_pieces and t["total_ms"] > 0 and t["wall_ms"] >= t["total_ms"] * 0.5, (n, chunks, t)
        t = m.last_timing()
        assert t["compute_ms"] > 0 and t["rccl_version"] == 0
    finally:
        m.close()


_FAKE_RCCL_SCRIPT = r"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["BJJ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BJJ_ROOT"], "tests"))
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
from conftest import Oracle
orc = Oracle()
fake = ctypes.CDLL(os.environ["BJJ_RCCL_LIBRARY"])
cnt = (ctypes.c_long * 5)()
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)
for g in (2, 3, 8):
    m = bjj.MultiContext([0] * g, 8)            # default transport: RCCL -- here the test double
    for n in (g * 640, g * 640 + 5, 3, g):
        even = n % g == 0
        sc = w.scalars_254(n, offset=3 * n + g)
        want = orc.mul_fixed_base(sc)
        d_sc, d_out = up(sc), torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        fake.fake_rccl_counters(cnt)
        m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
        fake.fake_rccl_counters(cnt)
        groups, scatter, gather, send, recv = list(cnt)
        assert groups == 2, (g, n, list(cnt))                       # one group out, one group back
        if even:
            assert (scatter, gather, send, recv) == (g, g, 0, 0), (g, n, list(cnt))
        else:
            peers = sum(1 for r in range(1, g) if m.shard_bounds(n, r)[1] > m.shard_bounds(n, r)[0])
            assert (scatter, gather, send, recv) == (0, 0, 2 * peers, 2 * peers), (g, n, list(cnt))
        fb = d_out.cpu().numpy().reshape(n, 64)
        assert (fb == want).all(), (g, n)
        A, R, S, msg = w.make_signatures(orc.mul_fixed_base, orc.poseidon5, n, offset=n)
        bad = w.corrupt(A, R, S, msg, n, offset=n)
        t = [up(x) for x in (A, R, S, msg)]
        d_ok = torch.full((((n + 15) // 16) * 16,), 9, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        fake.fake_rccl_counters(cnt)
        m.eddsa_verify_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), n, d_ok.data_ptr())
        fake.fake_rccl_counters(cnt)
        assert cnt[0] == 2 and (cnt[1] == 4 * g if even else cnt[3] > 0), (g, n, list(cnt))    # four input arrays in ONE group
        got = d_ok.cpu().numpy()
        assert (got[:n] == (~bad).astype(np.uint8)).all() and (got[n:] == 9).all(), (g, n)
    assert m.last_timing()["rccl_version"] == 99999
    # ---- the pipelined schedule: every peer block in up to `chunks` pieces, piece c of all peers and arrays in ONE group
    #      of exact-count send / receive pairs, results back piece by piece
    def pieces(cnt, chunks, min_chunk):
        if cnt == 0:
            return []
        c = max(1, min(chunks, cnt // min_chunk))
        csz = (-(-cnt // c) + 63) & ~63
        return [min(csz, cnt - lo) for lo in range(0, cnt, csz)]
    for chunks, n in ((4, g * 640), (3, g * 640 + 5), (16, g * 2000 + 1), (2, g + 1)):
        m.set_chunks(chunks, 64)
        per_peer = [pieces(m.shard_bounds(n, r)[1] - m.shard_bounds(n, r)[0], chunks, 64) for r in range(1, g)]
        rounds = max(len(p) for p in per_peer)
        A, R, S, msg = w.make_signatures(orc.mul_fixed_base, orc.poseidon5, n, offset=5 * n)
        bad = w.corrupt(A, R, S, msg, n, offset=5 * n)
        if n > 700:
            A[700, 3] ^= 2; bad[700] = True                      # an off-curve key inside a later piece
        t = [up(x) for x in (A, R, S, msg)]
        d_ok = torch.full((((n + 15) // 16) * 16,), 9, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        fake.fake_rccl_counters(cnt)
        m.eddsa_verify_dev(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), n, d_ok.data_ptr())
        fake.fake_rccl_counters(cnt)
        groups, scatter, gather, send, recv = list(cnt)
        total = sum(len(p) for p in per_peer)
        assert (groups, scatter, gather, send, recv) == (2 * rounds, 0, 0, 5 * total, 5 * total), (g, n, chunks, list(cnt), per_peer)
        got = d_ok.cpu().numpy()
        assert (got[:n] == (~bad).astype(np.uint8)).all() and (got[n:] == 9).all(), (g, n, chunks)
        lt = m.last_timing()
        assert lt["chunks"] == max(rounds, 1) and lt["total_ms"] > 0, lt
        sc = w.scalars_254(n, offset=n + chunks)
        d_sc, d_out = up(sc), torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
        fb = d_out.cpu().numpy().reshape(n, 64)
        assert (fb == orc.mul_fixed_base(sc)).all(), (g, n, chunks)
        d_out2 = torch.zeros(n * 64, dtype=torch.uint8, device=dev)
        m.mul_var_base_dev(d_out.data_ptr(), d_sc.data_ptr(), n, d_out2.data_ptr())
        idx = np.arange(0, n, 37)
        assert (d_out2.cpu().numpy().reshape(n, 64)[idx] == orc.mul_var_base(fb[idx], sc[idx])).all(), (g, n, chunks)
    m.close()
print("fake-rccl ok")
"""


def test_multi_rccl_call_sequence_against_a_test_double():
    """The RCCL branch of bjj_multi_* for G = 2, 3, 8 on the one-GPU box: BJJ_RCCL_LIBRARY points the library at an in-process
    test double (tests/fake_rccl) that executes ncclScatter / ncclGather / ncclSend / ncclRecv with device copies and rejects what
    real RCCL would reject or hang on (a rank missing from a collective, unmatched send / receive, calls outside a group).
    Checks the buffers, offsets, counts and grouping the library hands to RCCL; every output against the oracle."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    d = os.path.join(ROOT, "tests", "fake_rccl")
    so = os.path.join(d, "libfake_rccl.so")
    if not os.path.exists(so) or os.path.getmtime(os.path.join(d, "fake_rccl.cpp")) > os.path.getmtime(so):
        r = subprocess.run(["make", "-s"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    env = dict(os.environ, BJJ_RCCL_LIBRARY=so, BJJ_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", _FAKE_RCCL_SCRIPT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=900)
    assert r.returncode == 0 and "fake-rccl ok" in r.stdout, r.stdout[-4000:]


_FAKE_RCCL_FAILURE_SCRIPT = r"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["BJJ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BJJ_ROOT"], "tests"))
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
from conftest import Oracle
orc = Oracle()
fake = ctypes.CDLL(os.environ["BJJ_RCCL_LIBRARY"])
fake.fake_rccl_aborted.restype = ctypes.c_long
dev = torch.device("cuda", 0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)
g, n = 3, 3 * 640 + 5
m = bjj.MultiContext([0] * g, 8)
m.set_chunks(4, 64)
sc = w.scalars_254(n, offset=1)
d_sc, d_out = up(sc), torch.zeros(n * 64, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
try:
    m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())       # FAKE_RCCL_FAIL_AT: a send of the second piece fails
    raise SystemExit("the injected failure was not reported")
except bjj.BjjError as e:
    assert "injected failure" in str(e) and "unusable" in str(e), str(e)
assert fake.fake_rccl_aborted() == g, fake.fake_rccl_aborted()         # every communicator aborted, none destroyed twice
try:
    m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
    raise SystemExit("a retired handle accepted another call")
except bjj.BjjError as e:
    assert "unusable" in str(e), str(e)
assert (m.mul_fixed_base(sc) == orc.mul_fixed_base(sc)).all()          # the host-pointer form needs no communicator
m.close()                                                              # must return (no hang on half-posted transfers)
os.environ["FAKE_RCCL_FAIL_AT"] = "0"
m2 = bjj.MultiContext([0] * g, 8)                                      # a fresh handle works
m2.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
assert (d_out.cpu().numpy().reshape(n, 64) == orc.mul_fixed_base(sc)).all()
m2.close()
print("fake-rccl failure path ok")
"""


def test_multi_rccl_failure_retires_the_handle_without_hanging():
    """a transfer that fails after the call has started to enqueue must not leave a half-posted group behind.
    The test double rejects the 7th posted operation (a send of the second piece): the library stops posting, the double --
    like real RCCL -- discards that group, the handle aborts its communicators (ncclCommAbort), reports the first error and
    refuses further device-resident calls; freeing it returns."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    d = os.path.join(ROOT, "tests", "fake_rccl")
    so = os.path.join(d, "libfake_rccl.so")
    if not os.path.exists(so) or os.path.getmtime(os.path.join(d, "fake_rccl.cpp")) > os.path.getmtime(so):
        r = subprocess.run(["make", "-s"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    env = dict(os.environ, BJJ_RCCL_LIBRARY=so, BJJ_ROOT=ROOT, FAKE_RCCL_FAIL_AT="7")
    r = subprocess.run([sys.executable, "-c", _FAKE_RCCL_FAILURE_SCRIPT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert r.returncode == 0 and "fake-rccl failure path ok" in r.stdout, r.stdout[-4000:]


_ALL_DEVICES_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["BJJ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BJJ_ROOT"], "tests"))
import babyjubjub_rs_amd as bjj
from babyjubjub_rs_amd import workload as w
from conftest import Oracle
orc = Oracle()
g = torch.cuda.device_count()
m = bjj.MultiContext(None, 16)
dev = torch.device("cuda", m.device(0))
up = lambda a: torch.from_numpy(np.ascontiguousarray(a).reshape(-1)).to(dev)
assert m.size == g
for chunks in (4, 1):                       # the pipelined schedule, then the serial ncclScatter / ncclGather one
    m.set_chunks(chunks, 64)
    for n in (g * 1000, g * 1000 + 7, 3, g * 40000 + 11):
        sc = w.scalars_254(n, offset=n)
        want = orc.mul_fixed_base(sc)
        assert (m.mul_fixed_base(sc) == want).all()
        d_sc = up(sc)
        d_out = torch.zeros(n * 64 + 64, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        m.mul_fixed_base_dev(d_sc.data_ptr(), n, d_out.data_ptr())
        got = d_out.cpu().numpy()
        assert (got[:n * 64].reshape(n, 64) == want).all() and not got[n * 64:].any(), (chunks, n)
        print("fixed base ok: chunks", chunks, "n", n, flush=True)
    n = g * 3000 + 5
    A, R, S, msg = w.make_signatures(orc.mul_fixed_base, orc.poseidon5, n)
    bad = w.corrupt(A, R, S, msg, n)
    d_ok = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    dA, dR, dS, dM = up(A), up(R), up(S), up(msg)
    torch.cuda.synchronize()
    m.eddsa_verify_dev(dA.data_ptr(), dR.data_ptr(), dS.data_ptr(), dM.data_ptr(), n, d_ok.data_ptr())
    ok = d_ok.cpu().numpy()
    assert (ok[:n] == (~bad).astype(np.uint8)).all() and not ok[n:].any(), chunks
    print("verify ok: chunks", chunks, flush=True)
m.close()
print("all-devices rccl ok", g)
"""


def test_multi_all_devices_scatter_gather():
    """every visible device over REAL RCCL (the driver's multi-GPU box; skipped on a 1-GPU box): even, ragged, tiny and
    multi-piece batches through both forms and both schedules, nothing written past n.  Runs in a child process under a hard
    time limit: this is the one path that cannot be exercised before that box exists, and a hang in it must cost one test,
    not the session."""
    import os
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    # BJJ_TEST_ALL_DEVICES_ON_ONE=1 runs the script on a 1-GPU box as well (G = 1 through RCCL): checks the script itself
    if torch.cuda.device_count() < 2 and os.environ.get("BJJ_TEST_ALL_DEVICES_ON_ONE") != "1":
        pytest.skip("needs at least two GPUs")
    env = dict(os.environ, BJJ_ROOT=ROOT)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, "-c", _ALL_DEVICES_SCRIPT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
    except subprocess.TimeoutExpired as e:
        pytest.fail("bjj_multi_* over real RCCL did not finish within 600 s; output so far:\n%s" % (e.stdout or "")[-4000:])
    assert r.returncode == 0 and "all-devices rccl ok" in r.stdout, r.stdout[-4000:]


def test_two_contexts_on_two_devices_in_one_process(oracle):
    """every *_dev entry selects its context's device: a context on GPU 1 next to one on GPU 0 (skipped on a 1-GPU box)"""
    import torch
    import babyjubjub_rs_amd as bjj
    from babyjubjub_rs_amd import workload as w
    if torch.cuda.device_count() < 2:
        pytest.skip("needs at least two GPUs")
    c0, c1 = bjj.Context(0, 16), bjj.Context(1, 16)
    try:
        n = 3000
        inp = w.random_u256(w.SEED_MSGS, 5 * n, 0, top_bits_cleared=3).reshape(n, 160)
        for c, d in ((c1, 1), (c0, 0), (c1, 1)):
            dev = torch.device("cuda", d)
            d_in = torch.from_numpy(inp.reshape(-1)).to(dev)
            d_out = torch.empty(n * 32, dtype=torch.uint8, device=dev)
            torch.cuda.set_device(0)            # the caller's current device is NOT the context's
            c.poseidon5_dev(d_in.data_ptr(), n, d_out.data_ptr(), 0)
            c.sync()
            assert (d_out.cpu().numpy().reshape(n, 32) == oracle.poseidon5(inp)).all()
    finally:
        c0.close(); c1.close()
