"""Builds and runs tests/cpp/test_reference_api.cpp (the reference's tests re-stated in C++ over
babyjubjub.hpp -> libbjj_hip.so) on a GPU box."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "test_reference_api.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "test_reference_api")
    libdir = os.path.join(ROOT, "babyjubjub-rs_amd", "csrc")
    cmd = ["g++", "-O1", "-std=c++17", "-o", exe, src, "-L" + libdir, "-lbjj_hip", "-Wl,-rpath," + libdir,
           "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_cpp_mirror_compiles_and_links():
    """CPU: the header-only C++ mirror compiles and links against the C ABI."""
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_reference_tests_on_gpu():
    r = subprocess.run([_build()], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "ok (reference tests re-stated in C++)" in r.stdout
