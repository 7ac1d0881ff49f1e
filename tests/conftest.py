import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _sh(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s failed in %s:\n%s" % (" ".join(cmd), cwd, r.stdout))


# ---- the oracle (test infrastructure) ------------------------------------------
class Oracle:
    """ctypes view of oracle/libbjj_oracle.so (the C restatement of the reference algorithm)."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "libbjj_oracle.so")
        if os.environ.get("BJJ_ORACLE_SANITIZE") == "1":   # tests/test_emul_sanitizers.py: the checker itself under ASan + UBSan
            path = os.path.join(ROOT, "oracle", "libbjj_oracle_san.so")
            src = os.path.join(ROOT, "oracle", "bjj_ref.c")
            if not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path):
                _sh(["gcc", "-O1", "-g", "-fPIC", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                     "-fno-omit-frame-pointer", "-shared", "-o", path, src, "-lpthread"], os.path.join(ROOT, "oracle"))
        elif not os.path.exists(path):
            _sh(["make", "-s"], os.path.join(ROOT, "oracle"))
        self.lib = ctypes.CDLL(path)
        self.threads = min(os.cpu_count() or 1, 64)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def mul_fixed_base(self, scalars):
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
        n = s.size // 32
        out = np.empty(n * 64, np.uint8)
        self.lib.bjjref_mul_fixed_base_batch(self._p(s), ctypes.c_size_t(n), self._p(out), self.threads)
        return out.reshape(n, 64)

    def mul_var_base(self, pts, scalars):
        p = np.ascontiguousarray(pts, dtype=np.uint8).reshape(-1)
        s = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1)
        n = s.size // 32
        out = np.empty(n * 64, np.uint8)
        self.lib.bjjref_mul_var_base_batch(self._p(p), self._p(s), ctypes.c_size_t(n), self._p(out), self.threads)
        return out.reshape(n, 64)

    def poseidon5(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
        n = a.size // 160
        out = np.empty(n * 32, np.uint8)
        self.lib.bjjref_poseidon5_batch(self._p(a), ctypes.c_size_t(n), self._p(out), self.threads)
        return out.reshape(n, 32)

    def verify(self, pk, r, s, m):
        pk, r, s, m = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1) for x in (pk, r, s, m))
        n = s.size // 32
        out = np.empty(n, np.uint8)
        self.lib.bjjref_verify_batch(self._p(pk), self._p(r), self._p(s), self._p(m), ctypes.c_size_t(n), self._p(out),
                                     self.threads)
        return out

    def decompress(self, comp):
        a = np.ascontiguousarray(comp, dtype=np.uint8).reshape(-1)
        n = a.size // 32
        out = np.empty(n * 64, np.uint8)
        ok = np.empty(n, np.uint8)
        self.lib.bjjref_decompress_batch(self._p(a), ctypes.c_size_t(n), self._p(out), self._p(ok), self.threads)
        return out.reshape(n, 64), ok

    def compress(self, pts):
        a = np.ascontiguousarray(pts, dtype=np.uint8).reshape(-1)
        n = a.size // 64
        out = np.empty(n * 32, np.uint8)
        self.lib.bjjref_compress_batch(self._p(a), ctypes.c_size_t(n), self._p(out), self.threads)
        return out.reshape(n, 32)

    def verify_compressed(self, pk, sig, msg):
        pk, sig, msg = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1) for x in (pk, sig, msg))
        n = pk.size // 32
        out = np.empty(n, np.uint8)
        self.lib.bjjref_verify_compressed_batch(self._p(pk), self._p(sig), self._p(msg), ctypes.c_size_t(n),
                                                self._p(out), self.threads)
        return out

    def verify_schnorr(self, pk, r, s, m):
        pk, r, s, m = (np.ascontiguousarray(x, dtype=np.uint8).reshape(-1) for x in (pk, r, s, m))
        n = s.size // 32
        out = np.empty(n, np.uint8)
        self.lib.bjjref_verify_schnorr_batch(self._p(pk), self._p(r), self._p(s), self._p(m), ctypes.c_size_t(n),
                                             self._p(out), self.threads)
        return out

    def sign(self, keys, msgs):
        k = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1)
        m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(-1)
        n = k.size // 32
        r, s, ok = np.empty(n * 64, np.uint8), np.empty(n * 32, np.uint8), np.empty(n, np.uint8)
        self.lib.bjjref_sign_batch(self._p(k), self._p(m), ctypes.c_size_t(n), self._p(r), self._p(s), self._p(ok),
                                   self.threads)
        return r.reshape(n, 64), s.reshape(n, 32), ok

    def public_keys(self, keys):
        k = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1)
        n = k.size // 32
        out = np.empty(n * 64, np.uint8)
        self.lib.bjjref_public_batch(self._p(k), ctypes.c_size_t(n), self._p(out), self.threads)
        return out.reshape(n, 64)

    def point_add(self, p, q):
        """p.projective().add(&q.projective()).affine() per item (src/lib.rs:141-147, 88-131, 70-85)"""
        p = np.ascontiguousarray(p, dtype=np.uint8).reshape(-1)
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1)
        n = p.size // 64
        out = np.empty(n * 64, np.uint8)
        self.lib.bjjref_point_add_batch(self._p(p), self._p(q), ctypes.c_size_t(n), self._p(out), self.threads)
        return out.reshape(n, 64)


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def pyoracle():
    """The pure-Python oracle: CPU-side test infrastructure only.  No `-m gpu` test takes this fixture
    (tests/test_golden_gpu_expected.py enforces it): on the GPU box expected values come from tests/golden/*.json and from
    the C oracle.  oracle/ is put on sys.path here and nowhere else."""
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)
    import bjj_oracle
    return bjj_oracle


@pytest.fixture(scope="session")
def golden():
    g = {}
    for name in ("reference_kats", "oracle_vectors", "gpu_expected"):
        with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
            g[name] = json.load(f)
    return g


# ---- CPU emulation of the kernel bodies (debug harness, see tests/emul) ----------
@pytest.fixture(scope="session", params=[0, 1], ids=["raw_per_lane_tables", "packed_per_lane_tables"])
def emul(request):
    """the kernel bodies compiled for the CPU with bound assertions -- once per per-lane table layout that ships
    (BJJ_PNIELS_LAYOUT: 0 = raw entries, the verify unit; 1 = packed entries, the variable-base unit)"""
    d = os.path.join(ROOT, "tests", "emul")
    layout = request.param
    # BJJ_EMUL_SANITIZE=1 (set by tests/test_emul_sanitizers.py for a child pytest that runs under LD_PRELOAD=libasan):
    # the same bodies with AddressSanitizer + UndefinedBehaviorSanitizer, any report aborts the process
    san = os.environ.get("BJJ_EMUL_SANITIZE") == "1"
    so = os.path.join(d, "libbjj_emul_l%d%s.so" % (layout, "_san" if san else ""))
    srcs = [os.path.join(d, "emul_bodies.cpp")] + [
        os.path.join(ROOT, "babyjubjub-rs_amd", "csrc", f)
        for f in ("fr.hpp", "curve.hpp", "poseidon.hpp", "bjj_device.hpp", "bjj_constants.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        extra = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"] if san else []
        _sh(["g++", "-O1" if san else "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-DBJJ_PNIELS_LAYOUT=%d" % layout] + extra
            + ["-o", so, srcs[0]], d)
    return ctypes.CDLL(so)


# ---- the product on a GPU ---------------------------------------------------------
@pytest.fixture(scope="session")
def gpu_ctx():
    """The session's context: EXPLICITLY the 28-bit table (154.6 GB), the configuration bench.py measures.  Only when that
    table cannot be allocated (the GPU has other tenants) does the session fall back to the widest that fits -- and then
    `ctx_w28` skips, so no test silently claims W = 28.  Raises loudly if the library or the GPU is missing: no fallback."""
    import babyjubjub_rs_amd as bjj
    try:
        ctx = bjj.Context(0, 28)
    except bjj.BjjError as e:
        if "cannot allocate the fixed-base table" not in str(e):
            raise
        ctx = bjj.Context(0, bjj.WINDOW_AUTO)
    print("\n[gpu_ctx] fixed-base window_bits = %d (%.1f GB table)" % (ctx.info().window_bits, ctx.info().table_bytes / 1e9))
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def ctx_w28(gpu_ctx):
    """the headline configuration, asserted: 28-bit windows, 9 digits"""
    info = gpu_ctx.info()
    if info.window_bits != 28:
        pytest.skip("bjj_init(window_bits=28) reported NOMEM on this box; the session runs with %d bits" % info.window_bits)
    assert info.n_windows == 9 and info.table_bytes == 9 * ((1 << 27) + 1) * 128
    return gpu_ctx


@pytest.fixture(scope="session")
def ctx_w23():
    """the library's default configuration (bjj_init(.., 0, ..)): 23-bit windows, 11 digits, 5.9 GB"""
    import babyjubjub_rs_amd as bjj
    ctx = bjj.Context(0, 0)
    info = ctx.info()
    assert info.window_bits == 23 and info.n_windows == 11 and info.table_bytes == 11 * ((1 << 22) + 1) * 128
    yield ctx
    ctx.close()


@pytest.fixture
def ctx_for_window(request):
    """indirect parametrisation: the context whose table has exactly `request.param` window bits"""
    ctx = request.getfixturevalue({23: "ctx_w23", 28: "ctx_w28"}[request.param])
    assert ctx.info().window_bits == request.param
    print("[window_bits = %d]" % ctx.info().window_bits)
    return ctx


# ---- helpers shared by tests ---------------------------------------------------------
def le32(v):
    return int(v).to_bytes(32, "little")


def pack(vals):
    """list of ints (or tuples of ints) -> flat uint8 array of 32-byte LE records"""
    b = bytearray()
    for v in vals:
        for x in (v if isinstance(v, (tuple, list)) else (v,)):
            b += le32(int(x, 16) if isinstance(x, str) else x)
    return np.frombuffer(bytes(b), np.uint8).copy()


def ints(v):
    """fixture value -> int / tuple of ints (values are stored as hex strings or plain ints, points as 2-lists)"""
    if isinstance(v, (list, tuple)):
        return tuple(ints(x) for x in v)
    return int(v, 16) if isinstance(v, str) else int(v)


def unpack(arr, per_item=1):
    b = bytes(arr) if isinstance(arr, (bytes, bytearray)) else np.ascontiguousarray(arr, np.uint8).tobytes()
    vals = [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]
    if per_item == 1:
        return vals
    return [tuple(vals[i:i + per_item]) for i in range(0, len(vals), per_item)]
