"""The kernel bodies (bjj_device.hpp, poseidon.hpp, curve.hpp, fr.hpp, sign.hpp) under AddressSanitizer and
UndefinedBehaviorSanitizer: the CPU emulation suite (tests/test_emul_bodies.py, both per-lane table layouts) runs once more in a
child pytest whose harness library is built with -fsanitize=address,undefined -fno-sanitize-recover=all.  GPU sanitizers are
not available on the pool, so this is where out-of-bounds table indices, shift counts >= the type width, signed overflow
and misaligned or uninitialised-length accesses in the per-item code would show up."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_kernel_bodies_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, BJJ_EMUL_SANITIZE="1", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_emul_bodies.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=1500)
    tail = r.stdout[-6000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, tail
    assert " passed" in r.stdout and "libbjj_emul_l0_san.so" in " ".join(os.listdir(os.path.join(ROOT, "tests", "emul")))


def test_c_oracle_under_asan_and_ubsan():
    """the checker: oracle/bjj_ref.c against every reference KAT and the codec / sign / Schnorr vectors, sanitized"""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, BJJ_ORACLE_SANITIZE="1", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = [os.path.join(ROOT, "tests", f) for f in ("test_oracle_kat.py", "test_codec.py", "test_sign.py", "test_schnorr.py")]
    r = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"], cwd=ROOT,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = r.stdout[-6000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, tail
    assert " passed" in r.stdout and os.path.exists(os.path.join(ROOT, "oracle", "libbjj_oracle_san.so"))


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_copy_workers_of_the_host_pipeline_under_tsan_and_asan(san, tmp_path):
    """babyjubjub-rs_amd/csrc/copy_pool.hpp driven like run_pipelined drives it (groups one chunk ahead, ordered harvest, a
    recycled 4-deep ring), with 1, 4 and 7 workers: no data race, no out-of-bounds slice, every byte delivered"""
    exe = str(tmp_path / "emul_copy_pool")
    src = os.path.join(ROOT, "tests", "emul", "emul_copy_pool.cpp")
    c = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-o", exe, src, "-lpthread"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if c.returncode != 0 and "sanitize" in c.stdout:
        pytest.skip("sanitizer runtime not available: " + c.stdout[-300:])
    assert c.returncode == 0, c.stdout
    for workers, nbytes in ((4, (37 << 20) + 12345), (1, 1000), (7, (11 << 20) + 1)):
        r = subprocess.run([exe, str(workers), str(nbytes)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=1"))
        assert r.returncode == 0 and "copy_pool ok %d" % nbytes in r.stdout, r.stdout[-3000:]
        assert "ThreadSanitizer" not in r.stdout and "AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-3000:]
