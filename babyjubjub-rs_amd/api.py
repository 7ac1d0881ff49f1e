"""Host-side mirror of the reference crate's API for the accelerated path.

Names and argument meaning follow /root/reference/src/lib.rs:
  Point{x,y}.mul_scalar(n)      lib.rs:149-164   -> Point.mul_scalar / mul_scalar_batch
  Point.projective()/.equals()  lib.rs:141-147, 180-185
  PointProjective.add/.affine   lib.rs:70-131    -> point_add_batch (add + affine)
  Signature{r_b8, s}            lib.rs:239-243
  verify(pk, sig, msg) -> bool  lib.rs:395-412   -> verify / verify_batch
  B8.mul_scalar(k) (PrivateKey::public, lib.rs:304-306) -> mul_fixed_base_batch

Everything numeric happens in libbjj_hip.so on the GPU.  This file only marshals
Python ints / numpy byte arrays to the 32-byte little-endian records of the C ABI.
Like the reference, mul_scalar and add are infallible and verify() folds every
failure into False; BjjError is raised only for API misuse / HIP errors.
"""
import ctypes

import numpy as np

from . import _lib

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # lib.rs:33-36
B8 = (
    5299619240641551281634865583518297030282874472190772894086521144482721001553,
    16950150798460657717958625567821834550301663161624707787222815936182638968203,
)  # lib.rs:37-46
SUBORDER = 21888242871839275222246405745257275088614511777268538073601725287587578984328 >> 3  # lib.rs:53-58


class BjjError(RuntimeError):
    pass


def _as_u8(a, width, name):
    """Accepts an (n, width) / (n*width,) uint8 array, or a list of ints / int tuples."""
    if isinstance(a, np.ndarray):
        if a.dtype != np.uint8:
            # a value cast would keep only the low byte of every element: reinterpret little-endian unsigned
            # limbs (e.g. the natural (n, 4) uint64 layout of Fr::into_repr().0) and refuse everything else
            if a.dtype.kind != "u" or 32 % a.dtype.itemsize or (a.dtype.itemsize > 1 and a.dtype.byteorder == ">"):
                raise BjjError("%s: array dtype %s is neither uint8 bytes nor little-endian unsigned limbs" % (name, a.dtype))
            if a.dtype.byteorder == "=" and a.dtype.itemsize > 1:
                import sys as _sys
                if _sys.byteorder != "little":
                    raise BjjError("%s: native-endian limbs on a big-endian host" % name)
            a = np.ascontiguousarray(a).view(np.uint8)
        arr = np.ascontiguousarray(a).reshape(-1)
        if arr.size % width:
            raise BjjError("%s: byte length %d is not a multiple of %d" % (name, arr.size, width))
        return arr
    if isinstance(a, (bytes, bytearray)):
        return _as_u8(np.frombuffer(bytes(a), dtype=np.uint8), width, name)
    out = bytearray()
    for item in a:
        vals = item if isinstance(item, (tuple, list)) else (item,)
        if len(vals) * 32 != width:
            raise BjjError("%s: expected %d integers per item" % (name, width // 32))
        for v in vals:
            v = int(v)
            if v < 0 or v >> 256:
                raise BjjError("%s: integer out of the unsigned 256-bit range" % name)
            out += v.to_bytes(32, "little")
    return np.frombuffer(bytes(out), dtype=np.uint8).copy()


def _ints(arr, per_item):
    b = arr.tobytes()
    vals = [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]
    if per_item == 1:
        return vals
    return [tuple(vals[i:i + per_item]) for i in range(0, len(vals), per_item)]


class Context:
    """One GPU + stream + fixed-base table (bjj_init / bjj_free).
    window_bits: 0 = the library default (23 bits, 5.9 GB), WINDOW_AUTO = widest table that fits in 60 % of the free
    HBM, or an explicit width 4..28 (28 = 154.6 GB, what bench.py measures)."""

    def __init__(self, device=0, window_bits=0, _borrowed=None):
        self.lib = _lib.load()
        self._owned = _borrowed is None
        if _borrowed is not None:  # a per-device context owned by a MultiContext
            self.handle = ctypes.c_void_p(_borrowed)
            return
        h = ctypes.c_void_p()
        rc = self.lib.bjj_init(int(device), int(window_bits), ctypes.byref(h))
        if rc != _lib.BJJ_OK:
            raise BjjError("bjj_init failed (%d): %s" % (rc, self.lib.bjj_last_error().decode()))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            for ptr in list(getattr(self, "_pinned", {}).values()):   # arrays from host_empty that were never released
                self.lib.bjj_host_free(self.handle, ptr)
            self._pinned = {}
            if self._owned:
                self.lib.bjj_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != _lib.BJJ_OK:
            raise BjjError("%s failed (%d): %s" % (what, rc, self.lib.bjj_last_error().decode()))

    def info(self):
        i = _lib.BjjInfo()
        i.struct_size = ctypes.sizeof(_lib.BjjInfo)
        self._ck(self.lib.bjj_get_info(self.handle, ctypes.byref(i)), "bjj_get_info")
        return i

    def check_table(self):
        """number of violated link conditions of the fixed-base table (0 = sound), checked on the device"""
        bad = ctypes.c_uint64(0)
        self._ck(self.lib.bjj_check_table(self.handle, ctypes.byref(bad)), "bjj_check_table")
        return bad.value

    def sync(self):
        self._ck(self.lib.bjj_sync(self.handle), "bjj_sync")

    def reserve(self, n):
        self._ck(self.lib.bjj_reserve(self.handle, n), "bjj_reserve")

    # ---- pinned host memory (bjj_host_alloc / bjj_host_register): host-pointer calls copy such arrays directly ----
    def host_empty(self, nbytes):
        """uint8 numpy array of `nbytes` bytes in page-locked memory owned by the library; release with host_free(array).
        Host-pointer calls whose arrays live in such memory skip the staging copy (include/bjj_hip.h)."""
        p = ctypes.c_void_p()
        self._ck(self.lib.bjj_host_alloc(self.handle, int(nbytes), ctypes.byref(p)), "bjj_host_alloc")
        buf = (ctypes.c_uint8 * int(nbytes)).from_address(p.value)
        a = np.frombuffer(buf, dtype=np.uint8)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p.value
        return a

    def host_free(self, a):
        """release an array from host_empty (the array must not be used afterwards)"""
        ptr = getattr(self, "_pinned", {}).pop(a.ctypes.data, None)
        if ptr is None:
            raise BjjError("host_free: not an array from host_empty")
        self._ck(self.lib.bjj_host_free(self.handle, ptr), "bjj_host_free")

    def host_register(self, a):
        """pin the memory of an existing contiguous numpy array in place (hipHostRegister)"""
        self._ck(self.lib.bjj_host_register(self.handle, a.ctypes.data, a.nbytes), "bjj_host_register")

    def host_unregister(self, a):
        self._ck(self.lib.bjj_host_unregister(self.handle, a.ctypes.data), "bjj_host_unregister")

    def host_is_pinned(self, a):
        rc = self.lib.bjj_host_is_pinned(self.handle, a.ctypes.data, a.nbytes)
        if rc < 0:
            self._ck(rc, "bjj_host_is_pinned")
        return bool(rc)

    # ---- host-buffer batch calls (numpy uint8 in / out) ----
    def mul_fixed_base(self, scalars):
        s = _as_u8(scalars, 32, "scalars")
        n = s.size // 32
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_mul_fixed_base(self.handle, s.ctypes.data, n, out.ctypes.data), "bjj_mul_fixed_base")
        return out.reshape(n, 64)

    def mul_fixed_base_compressed(self, scalars):
        """B8.mul_scalar(n).compress() in one pass (lib.rs:149-164 + 166-178) -> (n, 32)"""
        s = _as_u8(scalars, 32, "scalars")
        n = s.size // 32
        out = np.empty(n * 32, dtype=np.uint8)
        self._ck(self.lib.bjj_mul_fixed_base_compressed(self.handle, s.ctypes.data, n, out.ctypes.data), "bjj_mul_fixed_base_compressed")
        return out.reshape(n, 32)

    def mul_var_base(self, points, scalars):
        p = _as_u8(points, 64, "points")
        s = _as_u8(scalars, 32, "scalars")
        n = s.size // 32
        if p.size != n * 64:
            raise BjjError("mul_var_base: %d points vs %d scalars" % (p.size // 64, n))
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_mul_var_base(self.handle, p.ctypes.data, s.ctypes.data, n, out.ctypes.data),
                 "bjj_mul_var_base")
        return out.reshape(n, 64)

    def mul_var_base_wide(self, points, scalars, scalar_bytes):
        """Point::mul_scalar for scalars of scalar_bytes (a multiple of 32) little-endian bytes each (lib.rs:149, 156-157)."""
        p = _as_u8(points, 64, "points")
        s = _as_u8(scalars, scalar_bytes, "scalars")
        n = p.size // 64
        if s.size != n * scalar_bytes:
            raise BjjError("mul_var_base_wide: %d points vs %d scalars" % (n, s.size // scalar_bytes))
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_mul_var_base_wide(self.handle, p.ctypes.data, s.ctypes.data, scalar_bytes, n, out.ctypes.data),
                 "bjj_mul_var_base_wide")
        return out.reshape(n, 64)

    def proj_add(self, p, q):
        """raw PointProjective::add (lib.rs:88-131): (n, 96) x/y/z records in and out, any z"""
        a = _as_u8(p, 96, "p")
        b = _as_u8(q, 96, "q")
        n = a.size // 96
        if b.size != a.size:
            raise BjjError("proj_add: array lengths disagree")
        out = np.empty(n * 96, dtype=np.uint8)
        self._ck(self.lib.bjj_proj_add(self.handle, a.ctypes.data, b.ctypes.data, n, out.ctypes.data), "bjj_proj_add")
        return out.reshape(n, 96)

    def proj_affine(self, p):
        """PointProjective::affine (lib.rs:70-85): (n, 96) -> (n, 64); z == 0 -> (0, 0)"""
        a = _as_u8(p, 96, "p")
        n = a.size // 96
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_proj_affine(self.handle, a.ctypes.data, n, out.ctypes.data), "bjj_proj_affine")
        return out.reshape(n, 64)

    def poseidon5(self, inputs):
        a = _as_u8(inputs, 160, "inputs")
        n = a.size // 160
        out = np.empty(n * 32, dtype=np.uint8)
        self._ck(self.lib.bjj_poseidon5(self.handle, a.ctypes.data, n, out.ctypes.data), "bjj_poseidon5")
        return out.reshape(n, 32)

    def eddsa_verify(self, pk, r_b8, s, msg):
        a = _as_u8(pk, 64, "pk")
        r = _as_u8(r_b8, 64, "r_b8")
        sv = _as_u8(s, 32, "s")
        m = _as_u8(msg, 32, "msg")
        n = sv.size // 32
        if a.size != n * 64 or r.size != n * 64 or m.size != n * 32:
            raise BjjError("eddsa_verify: array lengths disagree")
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_eddsa_verify(self.handle, a.ctypes.data, r.ctypes.data, sv.ctypes.data, m.ctypes.data, n,
                                           ok.ctypes.data), "bjj_eddsa_verify")
        return ok

    def schnorr_verify(self, pk, r, s, msg):
        """verify_schnorr in bulk: 1 / 0 = Ok(true / false), 2 = Err (msg > Q).  s: 32-byte integers."""
        a = _as_u8(pk, 64, "pk")
        rr = _as_u8(r, 64, "r")
        sv = _as_u8(s, 32, "s")
        m = _as_u8(msg, 32, "msg")
        n = sv.size // 32
        if a.size != n * 64 or rr.size != n * 64 or m.size != n * 32:
            raise BjjError("schnorr_verify: array lengths disagree")
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_schnorr_verify(self.handle, a.ctypes.data, rr.ctypes.data, sv.ctypes.data, m.ctypes.data, n,
                                             ok.ctypes.data), "bjj_schnorr_verify")
        return ok

    def point_add(self, p, q):
        a = _as_u8(p, 64, "p")
        b = _as_u8(q, 64, "q")
        n = a.size // 64
        if b.size != a.size:
            raise BjjError("point_add: array lengths disagree")
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_point_add(self.handle, a.ctypes.data, b.ctypes.data, n, out.ctypes.data), "bjj_point_add")
        return out.reshape(n, 64)

    def set_signer_constant_time(self, on=True):
        """signer hardening: public_keys / sign / sign_schnorr scan a small 4-bit table instead of indexing the big one
        with secret digits -- no secret-dependent address or branch; bit-identical results, ~2x slower sign"""
        self._ck(self.lib.bjj_set_signer_constant_time(self.handle, 1 if on else 0), "bjj_set_signer_constant_time")

    def scalar_keys(self, keys):
        a = _as_u8(keys, 32, "keys")
        n = a.size // 32
        out = np.empty(n * 32, dtype=np.uint8)
        self._ck(self.lib.bjj_scalar_keys(self.handle, a.ctypes.data, n, out.ctypes.data), "bjj_scalar_keys")
        return out.reshape(n, 32)

    def public_keys(self, keys):
        a = _as_u8(keys, 32, "keys")
        n = a.size // 32
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_public_keys(self.handle, a.ctypes.data, n, out.ctypes.data), "bjj_public_keys")
        return out.reshape(n, 64)

    def public_keys_compressed(self, keys):
        """sk.public().compress() in one pass (lib.rs:304-306 + 166-178) -> (n, 32)"""
        a = _as_u8(keys, 32, "keys")
        n = a.size // 32
        out = np.empty(n * 32, dtype=np.uint8)
        self._ck(self.lib.bjj_public_keys_compressed(self.handle, a.ctypes.data, n, out.ctypes.data), "bjj_public_keys_compressed")
        return out.reshape(n, 32)

    def sign_compressed(self, keys, msgs):
        """sk.sign(msg)?.compress() in one pass (lib.rs:308-342 + 245-258) -> (sig (n, 64), ok (n,)); ok == 0 (sig all-zero) for Err"""
        a = _as_u8(keys, 32, "keys")
        m = _as_u8(msgs, 32, "msgs")
        n = a.size // 32
        if m.size != a.size:
            raise BjjError("sign_compressed: array lengths disagree")
        sig = np.empty(n * 64, dtype=np.uint8)
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_sign_compressed(self.handle, a.ctypes.data, m.ctypes.data, n, sig.ctypes.data, ok.ctypes.data), "bjj_sign_compressed")
        return sig.reshape(n, 64), ok

    def sign(self, keys, msgs):
        """-> (r_b8 (n, 64), s (n, 32), ok (n,)); ok == 0 where the reference returns Err (msg > Q)"""
        a = _as_u8(keys, 32, "keys")
        m = _as_u8(msgs, 32, "msgs")
        n = a.size // 32
        if m.size != a.size:
            raise BjjError("sign: array lengths disagree")
        r = np.empty(n * 64, dtype=np.uint8)
        s = np.empty(n * 32, dtype=np.uint8)
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_sign(self.handle, a.ctypes.data, m.ctypes.data, n, r.ctypes.data, s.ctypes.data,
                                   ok.ctypes.data), "bjj_sign")
        return r.reshape(n, 64), s.reshape(n, 32), ok

    def sign_schnorr(self, keys, msgs, nonces):
        """PrivateKey::sign_schnorr in bulk with caller-supplied 1024-bit nonces (n x 128 bytes, little-endian).
        -> (r (n, 64), s (n, 160): the reference's unreduced k + scalar_key*h, ok (n,): 0 = Err (msg > Q))"""
        a = _as_u8(keys, 32, "keys")
        m = _as_u8(msgs, 32, "msgs")
        k = np.ascontiguousarray(np.asarray(nonces, dtype=np.uint8)).reshape(-1)
        n = a.size // 32
        if m.size != a.size or k.size != n * 128:
            raise BjjError("sign_schnorr: array lengths disagree (nonces are 128 bytes each)")
        r = np.empty(n * 64, dtype=np.uint8)
        s = np.empty(n * 160, dtype=np.uint8)
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_sign_schnorr(self.handle, a.ctypes.data, m.ctypes.data, k.ctypes.data, n, r.ctypes.data,
                                           s.ctypes.data, ok.ctypes.data), "bjj_sign_schnorr")
        return r.reshape(n, 64), s.reshape(n, 160), ok

    def sign_schnorr_dev(self, d_keys, d_msgs, d_nonces, n, d_r, d_s, d_ok, stream=0):
        self._ck(self.lib.bjj_sign_schnorr_dev(self.handle, d_keys, d_msgs, d_nonces, n, d_r, d_s, d_ok, stream),
                 "bjj_sign_schnorr_dev")

    def sign_dev(self, d_keys, d_msgs, n, d_r, d_s, d_ok, stream=0):
        self._ck(self.lib.bjj_sign_dev(self.handle, d_keys, d_msgs, n, d_r, d_s, d_ok, stream), "bjj_sign_dev")

    def compress_points(self, points):
        a = _as_u8(points, 64, "points")
        n = a.size // 64
        out = np.empty(n * 32, dtype=np.uint8)
        self._ck(self.lib.bjj_compress_points(self.handle, a.ctypes.data, n, out.ctypes.data), "bjj_compress_points")
        return out.reshape(n, 32)

    def decompress_points(self, comp):
        """-> (points (n, 64), ok (n,)): ok == 0 where the reference's decompress_point returns Err"""
        a = _as_u8(comp, 32, "compressed points")
        n = a.size // 32
        out = np.empty(n * 64, dtype=np.uint8)
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_decompress_points(self.handle, a.ctypes.data, n, out.ctypes.data, ok.ctypes.data),
                 "bjj_decompress_points")
        return out.reshape(n, 64), ok

    def eddsa_verify_compressed(self, pk, sig, msg):
        """pk (n, 32), sig (n, 64) = compressed R then s, msg (n, 32) -> 1 / 0 / 2 (decompression Err)"""
        a = _as_u8(pk, 32, "pk")
        g = _as_u8(sig, 64, "sig")
        m = _as_u8(msg, 32, "msg")
        n = a.size // 32
        if g.size != n * 64 or m.size != n * 32:
            raise BjjError("eddsa_verify_compressed: array lengths disagree")
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_eddsa_verify_compressed(self.handle, a.ctypes.data, g.ctypes.data, m.ctypes.data, n,
                                                      ok.ctypes.data), "bjj_eddsa_verify_compressed")
        return ok

    # ---- device-pointer calls (integers: device addresses / hipStream_t) ----
    def mul_fixed_base_dev(self, d_scalars, n, d_out, stream=0):
        self._ck(self.lib.bjj_mul_fixed_base_dev(self.handle, d_scalars, n, d_out, stream), "bjj_mul_fixed_base_dev")

    def mul_var_base_dev(self, d_pts, d_scalars, n, d_out, stream=0):
        self._ck(self.lib.bjj_mul_var_base_dev(self.handle, d_pts, d_scalars, n, d_out, stream), "bjj_mul_var_base_dev")

    def poseidon5_dev(self, d_in, n, d_out, stream=0):
        self._ck(self.lib.bjj_poseidon5_dev(self.handle, d_in, n, d_out, stream), "bjj_poseidon5_dev")

    def eddsa_verify_dev(self, d_pk, d_r, d_s, d_msg, n, d_ok, stream=0):
        self._ck(self.lib.bjj_eddsa_verify_dev(self.handle, d_pk, d_r, d_s, d_msg, n, d_ok, stream),
                 "bjj_eddsa_verify_dev")

    def eddsa_verify_compressed_dev(self, d_pk, d_sig, d_msg, n, d_ok, stream=0):
        self._ck(self.lib.bjj_eddsa_verify_compressed_dev(self.handle, d_pk, d_sig, d_msg, n, d_ok, stream),
                 "bjj_eddsa_verify_compressed_dev")

    def decompress_points_dev(self, d_in, n, d_out, d_ok, stream=0):
        self._ck(self.lib.bjj_decompress_points_dev(self.handle, d_in, n, d_out, d_ok, stream),
                 "bjj_decompress_points_dev")

    def compress_points_dev(self, d_pts, n, d_out, stream=0):
        self._ck(self.lib.bjj_compress_points_dev(self.handle, d_pts, n, d_out, stream), "bjj_compress_points_dev")

    def point_add_dev(self, d_p, d_q, n, d_out, stream=0):
        self._ck(self.lib.bjj_point_add_dev(self.handle, d_p, d_q, n, d_out, stream), "bjj_point_add_dev")

    def proj_add_dev(self, d_p, d_q, n, d_out, stream=0):
        self._ck(self.lib.bjj_proj_add_dev(self.handle, d_p, d_q, n, d_out, stream), "bjj_proj_add_dev")

    def proj_affine_dev(self, d_p, n, d_out, stream=0):
        self._ck(self.lib.bjj_proj_affine_dev(self.handle, d_p, n, d_out, stream), "bjj_proj_affine_dev")

    def mul_var_base_wide_dev(self, d_pts, d_scalars, scalar_bytes, n, d_out, stream=0):
        self._ck(self.lib.bjj_mul_var_base_wide_dev(self.handle, d_pts, d_scalars, scalar_bytes, n, d_out, stream),
                 "bjj_mul_var_base_wide_dev")

    def public_keys_dev(self, d_keys, n, d_out, stream=0):
        self._ck(self.lib.bjj_public_keys_dev(self.handle, d_keys, n, d_out, stream), "bjj_public_keys_dev")

    def mul_fixed_base_compressed_dev(self, d_scalars, n, d_out32, stream=0):
        self._ck(self.lib.bjj_mul_fixed_base_compressed_dev(self.handle, d_scalars, n, d_out32, stream), "bjj_mul_fixed_base_compressed_dev")

    def public_keys_compressed_dev(self, d_keys, n, d_out32, stream=0):
        self._ck(self.lib.bjj_public_keys_compressed_dev(self.handle, d_keys, n, d_out32, stream), "bjj_public_keys_compressed_dev")

    def sign_compressed_dev(self, d_keys, d_msgs, n, d_sig64, d_ok, stream=0):
        self._ck(self.lib.bjj_sign_compressed_dev(self.handle, d_keys, d_msgs, n, d_sig64, d_ok, stream), "bjj_sign_compressed_dev")

    def scalar_keys_dev(self, d_keys, n, d_out, stream=0):
        self._ck(self.lib.bjj_scalar_keys_dev(self.handle, d_keys, n, d_out, stream), "bjj_scalar_keys_dev")

    def schnorr_verify_dev(self, d_pk, d_r, d_s, d_msg, n, d_ok, stream=0):
        self._ck(self.lib.bjj_schnorr_verify_dev(self.handle, d_pk, d_r, d_s, d_msg, n, d_ok, stream),
                 "bjj_schnorr_verify_dev")


class MultiContext:
    """All (or the listed) GPUs of this process behind one handle (bjj_multi_init): contiguous ceil(n/G) blocks per
    device, replicated tables.  Host arrays -> one pipeline thread per device; *_dev -> arrays resident on the first
    device, RCCL scatter / kernels / gather (BASELINE cfg 5)."""

    def __init__(self, devices=None, window_bits=0, transport="rccl"):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        if devices is None:
            arr, n = None, 0
        else:
            devices = [int(d) for d in devices]
            arr, n = (ctypes.c_int * len(devices))(*devices), len(devices)
        rc = self.lib.bjj_multi_init(arr, n, int(window_bits), ctypes.byref(h))
        if rc != _lib.BJJ_OK:
            raise BjjError("bjj_multi_init failed (%d): %s" % (rc, self.lib.bjj_last_error().decode()))
        self.handle = h
        if transport != "rccl":
            self.set_transport(transport)

    def set_transport(self, transport):
        """"rccl" (grouped ncclScatter / ncclGather) or "peer" (hipMemcpyPeerAsync of the same blocks) for the *_dev form"""
        t = {"rccl": _lib.BJJ_TRANSPORT_RCCL, "peer": _lib.BJJ_TRANSPORT_PEER_COPY}[transport]
        self._ck(self.lib.bjj_multi_set_transport(self.handle, t), "bjj_multi_set_transport")

    def set_chunks(self, chunks, min_chunk_items=1 << 15):
        """pipeline depth of the *_dev form: pieces per peer block (1 = serial scatter -> kernels -> gather)"""
        self._ck(self.lib.bjj_multi_set_chunks(self.handle, int(chunks), int(min_chunk_items)), "bjj_multi_set_chunks")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.bjj_multi_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != _lib.BJJ_OK:
            raise BjjError("%s failed (%d): %s" % (what, rc, self.lib.bjj_last_error().decode()))

    @property
    def size(self):
        return self.lib.bjj_multi_size(self.handle)

    def device(self, rank):
        return self.lib.bjj_multi_device(self.handle, rank)

    def ctx(self, rank):
        return Context(_borrowed=self.lib.bjj_multi_ctx(self.handle, rank))

    def shard_bounds(self, n, rank):
        lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
        self.lib.bjj_shard_bounds(n, self.size, rank, ctypes.byref(lo), ctypes.byref(hi))
        return lo.value, hi.value

    def last_timing(self):
        s, c, g, v = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._ck(self.lib.bjj_multi_last_timing(self.handle, ctypes.byref(s), ctypes.byref(c), ctypes.byref(g), ctypes.byref(v)),
                 "bjj_multi_last_timing")
        tot, wall, ch = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        self._ck(self.lib.bjj_multi_last_overlap(self.handle, ctypes.byref(tot), ctypes.byref(wall), ctypes.byref(ch)),
                 "bjj_multi_last_overlap")
        return {"scatter_ms": s.value, "compute_ms": c.value, "gather_ms": g.value, "rccl_version": v.value,
                "total_ms": tot.value, "wall_ms": wall.value, "chunks": ch.value}

    def mul_fixed_base(self, scalars):
        s = _as_u8(scalars, 32, "scalars")
        n = s.size // 32
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_mul_fixed_base_multi(self.handle, s.ctypes.data, n, out.ctypes.data), "bjj_mul_fixed_base_multi")
        return out.reshape(n, 64)

    def mul_var_base(self, points, scalars):
        p = _as_u8(points, 64, "points")
        s = _as_u8(scalars, 32, "scalars")
        n = s.size // 32
        if p.size != n * 64:
            raise BjjError("mul_var_base: %d points vs %d scalars" % (p.size // 64, n))
        out = np.empty(n * 64, dtype=np.uint8)
        self._ck(self.lib.bjj_mul_var_base_multi(self.handle, p.ctypes.data, s.ctypes.data, n, out.ctypes.data),
                 "bjj_mul_var_base_multi")
        return out.reshape(n, 64)

    def eddsa_verify(self, pk, r_b8, s, msg):
        a, r, sv, m = _as_u8(pk, 64, "pk"), _as_u8(r_b8, 64, "r_b8"), _as_u8(s, 32, "s"), _as_u8(msg, 32, "msg")
        n = sv.size // 32
        if a.size != n * 64 or r.size != n * 64 or m.size != n * 32:
            raise BjjError("eddsa_verify: array lengths disagree")
        ok = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.bjj_eddsa_verify_multi(self.handle, a.ctypes.data, r.ctypes.data, sv.ctypes.data, m.ctypes.data, n,
                                                 ok.ctypes.data), "bjj_eddsa_verify_multi")
        return ok

    # device-resident form: integer device addresses on the handle's first device; synchronous
    def mul_fixed_base_dev(self, d_scalars, n, d_out):
        self._ck(self.lib.bjj_mul_fixed_base_multi_dev(self.handle, d_scalars, n, d_out), "bjj_mul_fixed_base_multi_dev")

    def mul_var_base_dev(self, d_pts, d_scalars, n, d_out):
        self._ck(self.lib.bjj_mul_var_base_multi_dev(self.handle, d_pts, d_scalars, n, d_out), "bjj_mul_var_base_multi_dev")

    def eddsa_verify_dev(self, d_pk, d_r, d_s, d_msg, n, d_ok):
        self._ck(self.lib.bjj_eddsa_verify_multi_dev(self.handle, d_pk, d_r, d_s, d_msg, n, d_ok),
                 "bjj_eddsa_verify_multi_dev")


_DEFAULT = None


def default_context():
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = Context()
    return _DEFAULT


# ---- crate-shaped objects ----------------------------------------------------
class Point:
    """`pub struct Point { pub x: Fr, pub y: Fr }` (lib.rs:134-138); x, y are canonical ints."""

    __slots__ = ("x", "y")

    def __init__(self, x, y):
        self.x = int(x) % Q
        self.y = int(y) % Q

    def projective(self):  # lib.rs:141-147
        return PointProjective(self.x, self.y, 1)

    def mul_scalar(self, n, ctx=None):  # lib.rs:149-164 (sign of n dropped; n of any size)
        n = abs(int(n))
        ctx = ctx or default_context()
        if n >> 256:
            # `n: &BigInt` is unbounded (lib.rs:149): wide records of 32 k bytes; the device reduces mod 8l for an
            # on-curve point (exact) and replays all n.bits() bits of the reference's loop for an off-curve one
            nbytes = ((n.bit_length() + 255) // 256) * 32
            if nbytes > _lib.BJJ_MAX_SCALAR_BYTES:
                raise BjjError("mul_scalar: scalars wider than %d bits are outside the accelerated boundary"
                               % (8 * _lib.BJJ_MAX_SCALAR_BYTES))
            out = ctx.mul_var_base_wide([(self.x, self.y)], np.frombuffer(n.to_bytes(nbytes, "little"), np.uint8), nbytes)
        elif (self.x, self.y) == B8:
            out = ctx.mul_fixed_base([n])
        else:
            out = ctx.mul_var_base([(self.x, self.y)], [n])
        x, y = _ints(out, 2)[0]
        return Point(x, y)

    def compress(self, ctx=None):  # lib.rs:166-178 -> 32 bytes
        return bytes((ctx or default_context()).compress_points([(self.x, self.y)])[0])

    def equals(self, p):  # lib.rs:180-185
        return self.x == p.x and self.y == p.y

    def __eq__(self, o):
        return isinstance(o, Point) and self.equals(o)

    def __repr__(self):
        return "Point(x=%d, y=%d)" % (self.x, self.y)


class PointProjective:
    """`pub struct PointProjective { pub x, pub y, pub z }` (lib.rs:62-67): any z, like the reference."""

    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z=1):
        self.x, self.y, self.z = int(x) % Q, int(y) % Q, int(z) % Q

    def affine(self, ctx=None):  # lib.rs:70-85 (z == 0 -> (0, 0))
        x, y = _ints((ctx or default_context()).proj_affine([(self.x, self.y, self.z)]), 2)[0]
        return Point(x, y)

    def add(self, q, ctx=None):  # lib.rs:88-131: the raw (x, y, z) of the reference's formula sequence
        x, y, z = _ints((ctx or default_context()).proj_add([(self.x, self.y, self.z)], [(q.x, q.y, q.z)]), 3)[0]
        return PointProjective(x, y, z)


class Signature:
    """`pub struct Signature { pub r_b8: Point, pub s: BigInt }` (lib.rs:239-243)."""

    __slots__ = ("r_b8", "s")

    def __init__(self, r_b8, s):
        self.r_b8 = r_b8
        self.s = int(s)

    def compress(self, ctx=None):  # lib.rs:245-258 -> 64 bytes: compressed R, then the first 32 little-endian bytes of s
        s_le = self.s.to_bytes(max(32, (self.s.bit_length() + 7) // 8), "little")[:32]
        return self.r_b8.compress(ctx) + s_le


class PrivateKey:
    """`pub struct PrivateKey { pub key: [u8; 32] }` (lib.rs:270-342)."""

    __slots__ = ("key",)

    def __init__(self, key):
        self.key = bytes(key)

    @staticmethod
    def import_(b):  # PrivateKey::import, lib.rs:275-282
        if len(b) != 32:
            raise ValueError("imported key can not be bigger than 32 bytes")
        return PrivateKey(bytes(b))

    def scalar_key(self, ctx=None):  # lib.rs:284-302
        return _ints((ctx or default_context()).scalar_keys(np.frombuffer(self.key, np.uint8)), 1)[0]

    def public(self, ctx=None):  # lib.rs:304-306
        x, y = _ints((ctx or default_context()).public_keys(np.frombuffer(self.key, np.uint8)), 2)[0]
        return Point(x, y)

    def sign(self, msg, ctx=None):  # lib.rs:308-342 -> Signature; ValueError for the reference's Err
        msg = int(msg)
        if msg < 0:
            raise BjjError("sign: negative msg")
        if msg > Q:
            raise ValueError("msg outside the Finite Field")
        r, s, ok = (ctx or default_context()).sign(np.frombuffer(self.key, np.uint8), [msg])
        if not ok[0]:
            raise ValueError("msg outside the Finite Field")
        x, y = _ints(r, 2)[0]
        return Signature(Point(x, y), _ints(s, 1)[0])

    def sign_schnorr(self, m, k=None, ctx=None):  # lib.rs:344-361 -> (Point, BigInt); ValueError for Err
        """k: the 1024-bit nonce; None draws it from the OS generator as the reference does from
        rand::thread_rng (lib.rs:347-348).  s is the reference's unreduced integer k + scalar_key*h."""
        m = int(m)
        if m < 0:
            raise BjjError("sign_schnorr: negative msg")
        if m > Q:
            raise ValueError("msg outside the Finite Field")
        if k is None:
            import secrets
            k = secrets.randbits(1024)
        k = int(k)
        if k < 0 or k >> 1024:
            raise BjjError("sign_schnorr: nonce outside the 1024-bit record of the C ABI")
        r, s, ok = (ctx or default_context()).sign_schnorr(np.frombuffer(self.key, np.uint8), [m],
                                                           np.frombuffer(k.to_bytes(128, "little"), np.uint8))
        if not ok[0]:
            raise ValueError("msg outside the Finite Field")
        x, y = _ints(r, 2)[0]
        return Point(x, y), int.from_bytes(s[0].tobytes(), "little")


def decompress_point(bb, ctx=None):
    """decompress_point(bb: [u8; 32]) -> Result<Point, String>  (lib.rs:192-224); raises ValueError for Err"""
    if len(bb) != 32:
        raise BjjError("decompress_point: need exactly 32 bytes")
    pts, ok = (ctx or default_context()).decompress_points(np.frombuffer(bytes(bb), np.uint8))
    if not ok[0]:
        raise ValueError("decompress_point: y outside the field or x^2 not a (non-zero) square")
    x, y = _ints(pts, 2)[0]
    return Point(x, y)


def decompress_signature(b, ctx=None):
    """decompress_signature(b: &[u8; 64]) -> Result<Signature, String>  (lib.rs:260-268)"""
    if len(b) != 64:
        raise BjjError("decompress_signature: need exactly 64 bytes")
    return Signature(decompress_point(bytes(b[:32]), ctx), int.from_bytes(bytes(b[32:]), "little"))


def new_key():
    """new_key() -> PrivateKey (lib.rs:387-393): 1024 random bits, the first 32 big-endian bytes kept."""
    import secrets
    raw = secrets.randbits(1024).to_bytes(128, "big").lstrip(b"\x00")
    while len(raw) < 32:  # probability 2^-768; the reference would panic on the slice here
        raw = secrets.randbits(1024).to_bytes(128, "big").lstrip(b"\x00")
    return PrivateKey.import_(raw[:32])


def verify(pk, sig, msg, ctx=None):
    """verify(pk: Point, sig: Signature, msg: BigInt) -> bool  (lib.rs:395-412)."""
    msg = int(msg)
    if msg < 0:
        raise BjjError("verify: negative msg (the reference panics at lib.rs:399)")
    if msg > Q:  # lib.rs:396-398; also keeps msg inside the 32-byte record
        return False
    if sig.s < 0 or sig.s >> 256:
        raise BjjError("verify: s outside the 32-byte record of the C ABI")
    ctx = ctx or default_context()
    ok = ctx.eddsa_verify([(pk.x, pk.y)], [(sig.r_b8.x, sig.r_b8.y)], [sig.s], [msg])
    return bool(ok[0])


def verify_schnorr(pk, m, r, s, ctx=None):
    """verify_schnorr(pk: Point, m: BigInt, r: Point, s: BigInt) -> Result<bool, String>  (lib.rs:375-385).
    ValueError for the reference's Err.  s may be any non-negative integer: it is reduced mod 8l on the
    host, which is exact because it only multiplies B8 (lib.rs:377)."""
    m, s = int(m), int(s)
    if m < 0 or s < 0:
        raise BjjError("verify_schnorr: negative integer")
    if m > Q:
        raise ValueError("msg outside the Finite Field")
    ok = (ctx or default_context()).schnorr_verify([(pk.x, pk.y)], [(r.x, r.y)], [s % (8 * SUBORDER)], [m])
    if ok[0] == 2:
        raise ValueError("msg outside the Finite Field")
    return bool(ok[0])


# ---- batch forms ---------------------------------------------------------------
def mul_fixed_base_batch(scalars, ctx=None):
    return (ctx or default_context()).mul_fixed_base(scalars)


def mul_scalar_batch(points, scalars, ctx=None):
    return (ctx or default_context()).mul_var_base(points, scalars)


def poseidon5_batch(inputs, ctx=None):
    return (ctx or default_context()).poseidon5(inputs)


def verify_batch(pk, r_b8, s, msg, ctx=None):
    return (ctx or default_context()).eddsa_verify(pk, r_b8, s, msg)


def point_add_batch(p, q, ctx=None):
    return (ctx or default_context()).point_add(p, q)
