import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


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
        if not os.path.exists(path):
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
        p = np.ascontiguousarray(p, dtype=np.uint8).reshape(-1, 64)
        q = np.ascontiguousarray(q, dtype=np.uint8).reshape(-1, 64)
        one = np.zeros(32, np.uint8)
        one[0] = 1
        out = np.empty((p.shape[0], 64), np.uint8)
        for i in range(p.shape[0]):
            a = np.concatenate([p[i], one])
            b = np.concatenate([q[i], one])
            s = np.empty(96, np.uint8)
            self.lib.bjjref_proj_add(self._p(a), self._p(b), self._p(s))
            self.lib.bjjref_proj_affine(self._p(s), self._p(out[i]))
        return out


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def pyoracle():
    import bjj_oracle
    return bjj_oracle


@pytest.fixture(scope="session")
def golden():
    g = {}
    for name in ("reference_kats", "oracle_vectors"):
        with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
            g[name] = json.load(f)
    return g


# ---- CPU emulation of the kernel bodies (debug harness, see tests/emul) ----------
@pytest.fixture(scope="session")
def emul():
    d = os.path.join(ROOT, "tests", "emul")
    so = os.path.join(d, "libbjj_emul.so")
    srcs = [os.path.join(d, "emul_bodies.cpp")] + [
        os.path.join(ROOT, "babyjubjub-rs_amd", "csrc", f)
        for f in ("fr.hpp", "curve.hpp", "poseidon.hpp", "bjj_device.hpp", "bjj_constants.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        _sh(["g++", "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-o", so, srcs[0]], d)
    return ctypes.CDLL(so)


# ---- the product on a GPU ---------------------------------------------------------
@pytest.fixture(scope="session")
def gpu_ctx():
    import babyjubjub_rs_amd as bjj
    # the widest fixed-base table that fits (28 bits = 154.6 GB on an empty MI355X: the configuration bench.py measures);
    # raises loudly if the library or the GPU is missing: no fallback
    ctx = bjj.Context(0, bjj.WINDOW_AUTO)
    yield ctx
    ctx.close()


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


def unpack(arr, per_item=1):
    b = bytes(arr) if isinstance(arr, (bytes, bytearray)) else np.ascontiguousarray(arr, np.uint8).tobytes()
    vals = [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]
    if per_item == 1:
        return vals
    return [tuple(vals[i:i + per_item]) for i in range(0, len(vals), per_item)]
