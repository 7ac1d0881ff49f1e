"""The C-ABI library: loads here (no GPU), exports every symbol include/bjj_hip.h declares,
refuses to run without a device (no CPU fallback), and the host-side mirror marshals
correctly.  CPU only -- no compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "bjj_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bjj_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from babyjubjub_rs_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == header_functions()


def test_library_exports_every_declared_symbol():
    from babyjubjub_rs_amd import _lib
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), "libbjj_hip.so does not export %s" % name
    assert lib.bjj_version().decode().startswith("bjj-hip")


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from babyjubjub_rs_amd import _lib
    import babyjubjub_rs_amd as bjj
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.bjj_init(0, 0, ctypes.byref(h))
    assert rc == _lib.BJJ_E_NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.bjj_last_error()
    with pytest.raises(bjj.BjjError):
        bjj.Context(0)
    # NULL context is rejected, not dereferenced
    assert lib.bjj_mul_fixed_base(None, None, 4, None) == _lib.BJJ_E_INVALID
    assert lib.bjj_eddsa_verify_dev(None, None, None, None, None, 4, None, None) == _lib.BJJ_E_INVALID


def test_init_argument_checks():
    from babyjubjub_rs_amd import _lib
    lib = _lib.load()
    assert lib.bjj_init(0, 0, None) == _lib.BJJ_E_INVALID
    h = ctypes.c_void_p()
    assert lib.bjj_init(0, 3, ctypes.byref(h)) == _lib.BJJ_E_INVALID  # window_bits out of range
    assert lib.bjj_init(0, 29, ctypes.byref(h)) == _lib.BJJ_E_INVALID
    assert lib.bjj_init(0, -2, ctypes.byref(h)) == _lib.BJJ_E_INVALID


def test_multi_argument_checks_and_partition():
    """bjj_shard_bounds is pure host arithmetic: contiguous ceil(n/G) blocks, ragged tails, empty ranks"""
    import torch
    from babyjubjub_rs_amd import _lib
    from babyjubjub_rs_amd import workload as w
    lib = _lib.load()
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    for g in range(1, 9):
        for n in (0, 1, 7, 8, 9, 37, 1000, (1 << 20) + 333, 1 << 24):
            covered = 0
            for r in range(g):
                lib.bjj_shard_bounds(n, g, r, ctypes.byref(lo), ctypes.byref(hi))
                assert (lo.value, hi.value) == w.shard_bounds(n, g, r)
                assert lo.value == covered or lo.value == hi.value == n
                assert hi.value - lo.value <= -(-n // g)
                covered = hi.value
            assert covered == n
    h = ctypes.c_void_p()
    assert lib.bjj_multi_init(None, 1, 0, None) == _lib.BJJ_E_INVALID
    if not torch.cuda.is_available():
        assert lib.bjj_multi_init(None, 1, 0, ctypes.byref(h)) == _lib.BJJ_E_NO_DEVICE and not h.value
    assert lib.bjj_mul_fixed_base_multi(None, None, 4, None) == _lib.BJJ_E_INVALID
    assert lib.bjj_eddsa_verify_multi_dev(None, None, None, None, None, 4, None) == _lib.BJJ_E_INVALID
    assert lib.bjj_multi_size(None) == 0 and lib.bjj_multi_device(None, 0) == -1


def test_marshalling_helpers():
    from babyjubjub_rs_amd import api
    a = api._as_u8([(1, 2), (3, 4)], 64, "pts")
    assert a.dtype == np.uint8 and a.size == 128 and a[0] == 1 and a[32] == 2 and a[64] == 3
    assert api._ints(a, 2) == [(1, 2), (3, 4)]
    with pytest.raises(api.BjjError):
        api._as_u8([1 << 256], 32, "s")
    with pytest.raises(api.BjjError):
        api._as_u8([-1], 32, "s")
    with pytest.raises(api.BjjError):
        api._as_u8(np.zeros(33, np.uint8), 32, "s")
    # reference-shaped objects
    p = api.Point(api.Q + 5, 7)
    assert p.x == 5 and p.projective().z == 1 and p.equals(api.Point(5, 7))
    assert api.PointProjective(api.Q + 1, 2, 3).x == 1
    # little-endian unsigned limbs are reinterpreted, everything else that is not bytes is refused (no value cast)
    limbs = np.array([[1, 0, 0, 0], [2, 0, 0, 1 << 63]], dtype=np.uint64)
    b = api._as_u8(limbs, 32, "s")
    assert b.size == 64 and b[0] == 1 and b[32] == 2 and b[63] == 0x80
    with pytest.raises(api.BjjError):
        api._as_u8(np.zeros((2, 4), np.int64), 32, "s")
    with pytest.raises(api.BjjError):
        api._as_u8(np.zeros((2, 8), np.float32), 32, "s")
    with pytest.raises(api.BjjError):
        api._as_u8(np.zeros((2, 4), ">u8"), 32, "s")
    # verify(): msg > Q is false before anything touches the device (lib.rs:396-398)
    sig = api.Signature(api.Point(0, 1), 0)
    assert api.verify(api.Point(0, 1), sig, api.Q + 1, ctx=object()) is False


def test_workload_generator_matches_oracle_splitmix(pyoracle):
    from babyjubjub_rs_amd import workload as w
    g = pyoracle.SplitMix64(pyoracle.SEED_SCALARS)
    want = [g.u256() & ((1 << 254) - 1) for _ in range(10)]
    assert w.to_ints(w.scalars_254(10)) == want
    assert w.to_ints(w.scalars_254(4, offset=6)) == want[6:]
    assert [w.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert w.shard_bounds(2, 4, 3) == (2, 2)


def test_library_is_not_older_than_its_sources():
    """a stale libbjj_hip.so (sources edited, library not rebuilt) invalidates every GPU measurement"""
    d = os.path.join(ROOT, "babyjubjub-rs_amd", "csrc")
    so = os.path.join(d, "libbjj_hip.so")
    srcs = ["bjj_hip.hip", "bjj_multi.inc", "bjj_launch.hpp", "k_common.hpp", "k_fixed.hip", "k_var.hip", "k_hash_codec.hip",
            "k_verify.hip", "k_sign.hip", "k_small.hip", "fr.hpp", "fr_mul_columns.inc", "curve.hpp", "poseidon.hpp", "bjj_device.hpp",
            "sign.hpp", "bjj_constants.inc", "slot_queue.hpp", os.path.join("..", "..", "include", "bjj_hip.h")]
    newest = max(os.path.getmtime(os.path.join(d, s)) for s in srcs)
    assert os.path.getmtime(so) >= newest, "rebuild: python -c 'import __graft_entry__ as g; g.build()'"


def test_failure_injection_is_not_in_the_shipped_library():
    """BJJ_MULTI_INJECT_FAIL_GROUP (bjj_multi.inc) walks the RCCL failure path for tools/scale_session.sh: it is compiled in only with
    -DBJJ_TEST_HOOKS (`make hooks` -> tests/hooks/libbjj_hip_hooks.so).  The library a caller links does not know the variable."""
    import os
    from conftest import ROOT
    from babyjubjub_rs_amd import _lib
    shipped = open(os.path.join(ROOT, "babyjubjub-rs_amd", "csrc", "libbjj_hip.so"), "rb").read()
    assert b"BJJ_MULTI_INJECT_FAIL_GROUP" not in shipped and b"+test-hooks" not in shipped
    hooks = os.path.join(ROOT, "tests", "hooks", "libbjj_hip_hooks.so")
    if os.path.exists(hooks):
        h = open(hooks, "rb").read()
        assert b"BJJ_MULTI_INJECT_FAIL_GROUP" in h and b"+test-hooks" in h
    assert _lib.LIB_PATH.endswith(os.path.join("csrc", "libbjj_hip.so")) or os.environ.get("BJJ_LIB_PATH")
