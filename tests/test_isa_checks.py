"""Checks on the gfx950 machine code that actually ships (the code objects inside babyjubjub-rs_amd/csrc/libbjj_hip.so, taken
apart with llvm-objcopy / llvm-objdump; no GPU needed).

Slot hand-over (csrc/slot_queue.hpp): a workgroup returns its slot of per-lane table scratch with an atomic push,
and its table stores must have been acknowledged by the L2 -- `s_waitcnt vmcnt(0)` -- before that atomic is issued.  A
workgroup-scope release fence does not emit the wait on gfx950, so the source carries it explicitly (slot_release_wave);
this test keeps it from getting lost again: in every kernel that uses the rings, an `s_waitcnt vmcnt(0)` precedes the push
atomic with no vector-memory instruction in between, and in the multi-wave tile kernels every wave waits before the barrier
in front of the push."""
import os
import re
import struct
import subprocess

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "babyjubjub-rs_amd", "csrc", "libbjj_hip.so")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{mangled kernel name: [instruction text, ...]} of every gfx950 code object in the shipped library"""
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump")
    td = tmp_path_factory.mktemp("isa")
    fb = str(td / "fatbin.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, LIB], check=True)
    d = open(fb, "rb").read()
    magic, out, i = b"__CLANG_OFFLOAD_BUNDLE__", {}, 0
    while True:                                       # one bundle per translation unit, concatenated
        p = d.find(magic, i)
        if p < 0:
            break
        i = p + 1
        n = struct.unpack_from("<Q", d, p + 24)[0]
        off = p + 32
        for _ in range(n):
            o, s, tl = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + tl].decode()
            off += tl
            if "gfx950" not in triple or not s:
                continue
            co = str(td / ("co_%d.o" % p))
            open(co, "wb").write(d[p + o:p + o + s])
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], stdout=subprocess.PIPE, text=True, check=True).stdout
            for m in re.finditer(r"^[0-9a-f]+ <(\w+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, flags=re.M | re.S):
                out[m.group(1)] = [l.split("//")[0].strip() for l in m.group(2).split("\n") if l.strip()]
    assert out, "no gfx950 code object found in " + LIB
    return out


VMEM = re.compile(r"^(global|buffer|flat)_(load|store|atomic)")


def _push_atomic(ins):
    """index of the push: the LAST global_atomic_inc of the kernel (the pop's ticket is the first), which is followed by the
    compare-and-swap loop that refills the ring entry"""
    idx = [i for i, l in enumerate(ins) if l.startswith("global_atomic_inc")]
    assert len(idx) >= 2, "expected a pop ticket and a push ticket"
    assert any(l.startswith("global_atomic_cmpswap") for l in ins[idx[-1]:idx[-1] + 600]), "no CAS loop behind the push ticket"
    return idx[-1]


def _wait_before(ins, at, window=64):
    """walk back from instruction `at`: an s_waitcnt that includes vmcnt(0) must come before any vector-memory instruction"""
    for j in range(at - 1, max(0, at - window), -1):
        if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
            return j
        assert not VMEM.match(ins[j]), "vector memory instruction between the wait and the push: %s" % ins[j]
    raise AssertionError("no s_waitcnt vmcnt(0) within %d instructions before the push atomic" % window)


@pytest.mark.parametrize("kernel", ["bjj_k_eddsa_verify_groups", "bjj_k_schnorr_verify_groups",
                                    "bjj_k_mul_var_base_tiles", "bjj_k_mul_var_base_wide_tiles"])
def test_table_stores_are_acknowledged_before_the_slot_is_pushed(kernels, kernel):
    names = [k for k in kernels if re.match(r"_Z\d+%s[A-Z]" % kernel, k)]
    assert len(names) == 1, (kernel, names)
    ins = kernels[names[0]]
    push = _push_atomic(ins)
    w = _wait_before(ins, push)
    if "tiles" in kernel:
        # 256-lane workgroups, one slot per workgroup: EVERY wave waits for its own stores, then the barrier, then one thread pushes
        bars = [i for i in range(max(0, w - 40), w) if ins[i] == "s_barrier"]
        assert bars, "no barrier in front of the push of a multi-wave workgroup"
        b = bars[-1]
        _wait_before(ins, b, window=16)   # ... and nothing that touches memory between that wait and the barrier


def test_kernels_without_rings_do_not_touch_them(kernels):
    """the fixed-base kernels and the grid-strided / persistent forms hand out table scratch by lane index: no ring atomics"""
    for k, ins in kernels.items():
        if re.match(r"_Z\d+bjj_k_(mul_fixed_base|poseidon5|mul_var_base[A-Z]|eddsa_verify[A-Z])", k):
            assert not any(l.startswith("global_atomic_inc") for l in ins), k
