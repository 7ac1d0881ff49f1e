"""ctypes binding of libbjj_hip.so (the C ABI declared in include/bjj_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at
import time, and if no HIP device is usable `bjj_init` fails with BJJ_E_NO_DEVICE.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BJJ_LIB_PATH: developer override for interleaved A/B runs of several builds (tools/ab_*.sh) -- never a fallback
LIB_PATH = os.environ.get("BJJ_LIB_PATH") or os.path.join(_HERE, "csrc", "libbjj_hip.so")

BJJ_OK = 0
BJJ_E_INVALID = -1
BJJ_E_NO_DEVICE = -2
BJJ_E_HIP = -3
BJJ_E_NOMEM = -4
BJJ_E_RCCL = -5
BJJ_WINDOW_AUTO = -1
BJJ_MAX_SCALAR_BYTES = 4096
BJJ_TRANSPORT_RCCL = 0
BJJ_TRANSPORT_PEER_COPY = 1

# every symbol include/bjj_hip.h declares
EXPORTED_SYMBOLS = (
    "bjj_version", "bjj_last_error", "bjj_init", "bjj_free", "bjj_sync", "bjj_stream",
    "bjj_mul_fixed_base", "bjj_mul_var_base", "bjj_poseidon5", "bjj_eddsa_verify", "bjj_point_add",
    "bjj_mul_fixed_base_dev", "bjj_mul_var_base_dev", "bjj_poseidon5_dev", "bjj_eddsa_verify_dev",
    "bjj_point_add_dev", "bjj_reserve", "bjj_get_info", "bjj_check_table",
    "bjj_compress_points", "bjj_decompress_points", "bjj_eddsa_verify_compressed",
    "bjj_schnorr_verify", "bjj_schnorr_verify_dev",
    "bjj_set_signer_constant_time", "bjj_scalar_keys", "bjj_public_keys", "bjj_sign", "bjj_scalar_keys_dev", "bjj_public_keys_dev", "bjj_sign_dev",
    "bjj_sign_schnorr", "bjj_sign_schnorr_dev",
    "bjj_compress_points_dev", "bjj_decompress_points_dev", "bjj_eddsa_verify_compressed_dev",
    "bjj_mul_var_base_wide", "bjj_mul_var_base_wide_dev", "bjj_proj_add", "bjj_proj_add_dev",
    "bjj_proj_affine", "bjj_proj_affine_dev",
    "bjj_multi_init", "bjj_multi_free", "bjj_multi_size", "bjj_multi_ctx", "bjj_multi_device", "bjj_shard_bounds",
    "bjj_mul_fixed_base_multi", "bjj_mul_var_base_multi", "bjj_eddsa_verify_multi",
    "bjj_mul_fixed_base_multi_dev", "bjj_mul_var_base_multi_dev", "bjj_eddsa_verify_multi_dev",
    "bjj_multi_last_timing", "bjj_multi_set_transport", "bjj_multi_set_chunks", "bjj_multi_last_overlap",
    "bjj_host_alloc", "bjj_host_free", "bjj_host_register", "bjj_host_unregister", "bjj_host_is_pinned",
    "bjj_mul_fixed_base_compressed", "bjj_public_keys_compressed", "bjj_sign_compressed",
    "bjj_mul_fixed_base_compressed_dev", "bjj_public_keys_compressed_dev", "bjj_sign_compressed_dev",
)


class BjjInfo(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("device", ctypes.c_int),
        ("compute_units", ctypes.c_int),
        ("window_bits", ctypes.c_int),
        ("n_windows", ctypes.c_int),
        ("table_bytes", ctypes.c_uint64),
        ("scratch_bytes", ctypes.c_uint64),
        ("kernel_fixed_base", ctypes.c_char_p),
        ("kernel_var_base", ctypes.c_char_p),
        ("kernel_poseidon5", ctypes.c_char_p),
        ("kernel_verify", ctypes.c_char_p),
        ("init_ms", ctypes.c_double),
        ("signer_constant_time", ctypes.c_int),
        ("last_fixed_base_shape", ctypes.c_int),
        ("last_var_base_form", ctypes.c_int),
        ("last_verify_dispatch", ctypes.c_int),
        ("last_host_direct_arrays", ctypes.c_uint32),
        ("last_host_staged_arrays", ctypes.c_uint32),
        ("last_host_chunks", ctypes.c_uint32),
        ("host_copy_threads", ctypes.c_int),
        ("kernel_fixed_base_overlap", ctypes.c_char_p),
        ("kernel_var_base_overlap", ctypes.c_char_p),
        ("last_var_base_split", ctypes.c_int),
        ("last_host_zero_copy", ctypes.c_uint32),
        ("last_poseidon_form", ctypes.c_int),
        ("last_sign_form", ctypes.c_int),
    ]


class _Stub:
    """placeholder for a symbol an older A/B build does not export: calling it raises"""

    def __init__(self, name):
        self.name = name

    def __call__(self, *a):
        raise AttributeError("%s is not exported by %s" % (self.name, LIB_PATH))


class _Tolerant:
    """attribute access that survives missing symbols (only used with BJJ_LIB_PATH, for A/B runs against older builds)"""

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        try:
            return getattr(self._lib, name)
        except AttributeError:
            stub = _Stub(name)
            object.__setattr__(self, name, stub)
            return stub


def load():
    # torch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  If torch is going to
    # be used in this process (device memory / streams / torch.distributed), it must be the
    # first to load the HIP runtime, otherwise the process ends up with two runtimes and
    # torch reports "No HIP GPUs are available".  Pure C/C++ users are unaffected.
    if os.environ.get("BJJ_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libbjj_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C babyjubjub-rs_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = _Tolerant(ctypes.CDLL(LIB_PATH)) if os.environ.get("BJJ_LIB_PATH") else ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.bjj_version.restype = ctypes.c_char_p
    lib.bjj_last_error.restype = ctypes.c_char_p
    lib.bjj_init.argtypes = [ci, ci, ctypes.POINTER(vp)]
    lib.bjj_free.argtypes = [vp]
    lib.bjj_free.restype = None
    lib.bjj_sync.argtypes = [vp]
    lib.bjj_stream.argtypes = [vp]
    lib.bjj_stream.restype = vp
    lib.bjj_reserve.argtypes = [vp, sz]
    lib.bjj_get_info.argtypes = [vp, ctypes.POINTER(BjjInfo)]
    lib.bjj_check_table.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.bjj_host_alloc.argtypes = [vp, sz, ctypes.POINTER(vp)]
    lib.bjj_host_free.argtypes = [vp, vp]
    lib.bjj_host_register.argtypes = [vp, vp, sz]
    lib.bjj_host_unregister.argtypes = [vp, vp]
    lib.bjj_host_is_pinned.argtypes = [vp, vp, sz]
    lib.bjj_mul_fixed_base.argtypes = [vp, vp, sz, vp]
    lib.bjj_mul_var_base.argtypes = [vp, vp, vp, sz, vp]
    lib.bjj_poseidon5.argtypes = [vp, vp, sz, vp]
    lib.bjj_eddsa_verify.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.bjj_point_add.argtypes = [vp, vp, vp, sz, vp]
    lib.bjj_schnorr_verify.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.bjj_schnorr_verify_dev.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp]
    lib.bjj_mul_fixed_base_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_mul_var_base_dev.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.bjj_poseidon5_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_eddsa_verify_dev.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp]
    lib.bjj_point_add_dev.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.bjj_compress_points.argtypes = [vp, vp, sz, vp]
    lib.bjj_scalar_keys.argtypes = [vp, vp, sz, vp]
    lib.bjj_set_signer_constant_time.argtypes = [vp, ci]
    lib.bjj_public_keys.argtypes = [vp, vp, sz, vp]
    lib.bjj_sign.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    lib.bjj_scalar_keys_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_public_keys_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_sign_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp]
    lib.bjj_sign_schnorr.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp]
    lib.bjj_sign_schnorr_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp, vp, vp]
    lib.bjj_decompress_points.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_eddsa_verify_compressed.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.bjj_compress_points_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_decompress_points_dev.argtypes = [vp, vp, sz, vp, vp, vp]
    lib.bjj_eddsa_verify_compressed_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    lib.bjj_mul_var_base_wide.argtypes = [vp, vp, vp, sz, sz, vp]
    lib.bjj_mul_var_base_wide_dev.argtypes = [vp, vp, vp, sz, sz, vp, vp]
    lib.bjj_proj_add.argtypes = [vp, vp, vp, sz, vp]
    lib.bjj_proj_add_dev.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.bjj_proj_affine.argtypes = [vp, vp, sz, vp]
    lib.bjj_proj_affine_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_mul_fixed_base_compressed.argtypes = [vp, vp, sz, vp]
    lib.bjj_public_keys_compressed.argtypes = [vp, vp, sz, vp]
    lib.bjj_sign_compressed.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.bjj_mul_fixed_base_compressed_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_public_keys_compressed_dev.argtypes = [vp, vp, sz, vp, vp]
    lib.bjj_sign_compressed_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    # multi-GPU
    pd = ctypes.POINTER(ctypes.c_double)
    lib.bjj_multi_init.argtypes = [ctypes.POINTER(ci), ci, ci, ctypes.POINTER(vp)]
    lib.bjj_multi_free.argtypes = [vp]
    lib.bjj_multi_free.restype = None
    lib.bjj_multi_size.argtypes = [vp]
    lib.bjj_multi_ctx.argtypes = [vp, ci]
    lib.bjj_multi_ctx.restype = vp
    lib.bjj_multi_device.argtypes = [vp, ci]
    lib.bjj_shard_bounds.argtypes = [sz, ci, ci, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.bjj_shard_bounds.restype = None
    lib.bjj_mul_fixed_base_multi.argtypes = [vp, vp, sz, vp]
    lib.bjj_mul_var_base_multi.argtypes = [vp, vp, vp, sz, vp]
    lib.bjj_eddsa_verify_multi.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.bjj_mul_fixed_base_multi_dev.argtypes = [vp, vp, sz, vp]
    lib.bjj_mul_var_base_multi_dev.argtypes = [vp, vp, vp, sz, vp]
    lib.bjj_eddsa_verify_multi_dev.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.bjj_multi_last_timing.argtypes = [vp, pd, pd, pd, ctypes.POINTER(ci)]
    lib.bjj_multi_set_transport.argtypes = [vp, ci]
    lib.bjj_multi_set_chunks.argtypes = [vp, ci, sz]
    lib.bjj_multi_last_overlap.argtypes = [vp, pd, pd, ctypes.POINTER(ci)]
    return lib
