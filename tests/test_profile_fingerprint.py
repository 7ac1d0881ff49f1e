"""The static counters bench.py quotes (profiles/hbm_traffic.json, profiles/isa_mix.json) are tied to the build they were
taken from: a fingerprint of babyjubjub-rs_amd/csrc is stored next to them, and a tree that differs gets `traffic: null` +
`stale_profile: true` instead of another build's numbers."""
import json
import os
import shutil
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd"))
import srchash  # noqa: E402


class _Info:
    window_bits = 28


def _copy_csrc(tmp_path):
    import glob
    dst = tmp_path / "csrc"
    dst.mkdir()
    for pat in srchash.PATTERNS:
        for f in glob.glob(os.path.join(srchash.CSRC, pat)):
            shutil.copy(f, dst)
    return dst


def test_one_changed_token_of_a_kernel_source_or_one_flag_changes_the_fingerprint(tmp_path):
    dst = _copy_csrc(tmp_path)
    h0 = srchash.tree_hash(str(dst))
    assert h0 == srchash.tree_hash()                       # same files, same fingerprint
    p = dst / "k_fixed.hip"
    src = p.read_text()
    code = srchash.strip_comments(src)
    # one character of CODE: the first identifier character of the last non-comment line that also occurs verbatim in the file
    line = [l for l in code.split("\n") if l.strip() and src.count(l) == 1][-1]
    at = src.index(line) + (len(line) - len(line.lstrip()))
    p.write_text(src[:at] + ("X" if src[at] != "X" else "Y") + src[at + 1:])
    assert srchash.tree_hash(str(dst)) != h0
    p.write_text(src)
    assert srchash.tree_hash(str(dst)) == h0
    # a string literal is code even when it looks like a comment
    p.write_text(src + '\nstatic const char* bjj_note = "// not a comment";\n')
    h1 = srchash.tree_hash(str(dst))
    p.write_text(src + '\nstatic const char* bjj_note = "// not a comment!";\n')
    assert len({h0, h1, srchash.tree_hash(str(dst))}) == 3
    p.write_text(src)
    # the Makefile is hashed as it is: its `#` lines are not C comments, and a flag is a different build
    mk = dst / "Makefile"
    mk.write_text(mk.read_text().replace("-O3", "-O2", 1))
    assert srchash.tree_hash(str(dst)) != h0


def test_a_comment_only_edit_keeps_the_fingerprint(tmp_path):
    """the counters describe the code objects; a corrected sentence in a header comment is the same build"""
    dst = _copy_csrc(tmp_path)
    h0 = srchash.tree_hash(str(dst))
    for name in ("k_fixed.hip", "bjj_device.hpp", "bjj_multi.inc"):
        p = dst / name
        src = p.read_text()
        lines = src.split("\n")
        k = len(lines) // 2
        edited = lines[:k] + ["// a new remark", "/* and a block", "   over two lines */", ""] + lines[k:]
        p.write_text("\n".join(edited) + "\n// trailing remark\n")
        assert srchash.tree_hash(str(dst)) == h0, name
    assert srchash.strip_comments('a = "/*"; b = \'"\'; /* x */ c = 1; // y\n\n  \nd;') == 'a = "/*"; b = \'"\';   c = 1;\nd;'


def test_bench_does_not_quote_counters_of_another_build(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    prof = {"hbm_traffic.json": {"fixed_base": {"bytes_per_launch": 1.5e9, "source": "profiles/x.md", "batch": 1 << 20,
                                                "valu_insts_per_launch": 2.5e8, "window_bits": 28, "source_hash": "feedfacefeedface"}},
            "isa_mix.json": {"_source_hash": "feedfacefeedface",
                             "bjj_k_mul_fixed_base": {"avg_issue_cycles_per_valu_inst": 4.0, "quarter_rate_share": 0.76}}}
    monkeypatch.setattr(bench, "load_profile_json", lambda name: prof.get(name, {}))
    # (a) the tree IS the profiled one: counters are quoted, and the valu block prices the ceiling at the measured clock too
    monkeypatch.setattr(bench, "source_hash_now", lambda: "feedfacefeedface")
    r = bench.roofline_block("fixed_base", 0.6, 1 << 20, _Info())
    assert r["traffic"] == 1.5e9 and "stale_profile" not in r and "feedfacefeedface" in r["traffic_source"]
    v = bench.valu_block("fixed_base", 0.6, 1 << 20, _Info(), {"available": True, "sclk_mhz": 2160.0})
    assert abs(v["frac_at_measured_clock"] / v["frac"] - 2400.0 / 2160.0) < 1e-9 and v["clock_mhz"] == 2160.0
    # (b) one byte of a kernel source flipped since: nothing static is published
    monkeypatch.setattr(bench, "source_hash_now", lambda: "0123456789abcdef")
    r = bench.roofline_block("fixed_base", 0.6, 1 << 20, _Info())
    assert r["traffic"] is None and r["traffic_source"] is None and r["stale_profile"] is True
    assert r["achieved"] > 0 and r["frac"] > 0                    # the live part of the block is unaffected
    v = bench.valu_block("fixed_base", 0.6, 1 << 20, _Info())
    assert v["stale_profile"] is True and v["frac"] is None and "insts_per_launch" not in v
    # (c) a profile without any fingerprint (the pre-round-4 files) counts as stale
    del prof["hbm_traffic.json"]["fixed_base"]["source_hash"]
    monkeypatch.setattr(bench, "source_hash_now", lambda: "feedfacefeedface")
    assert bench.roofline_block("fixed_base", 0.6, 1 << 20, _Info())["stale_profile"] is True


def test_committed_profiles_carry_a_fingerprint_or_are_flagged():
    """whatever is committed under profiles/ either matches this tree or bench.py says so"""
    sys.path.insert(0, ROOT)
    import bench
    tr = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    kh = srchash.kernel_hashes()
    for kind in ("fixed_base", "verify", "var_base"):
        r = bench.roofline_block(kind, 1.0, 1 << 20, _Info())
        e = tr.get(kind, {})
        current = (kh.get(e.get("kernel")) == e["code_hash"]) if e.get("code_hash") else (e.get("source_hash") == srchash.tree_hash())
        assert (r["traffic"] is not None) == current and (r.get("stale_profile", False) is True) == (not current)


def test_kernel_code_fingerprints_of_the_shipped_library():
    """round 6: counters are tied to the MACHINE CODE of the kernel they describe (srchash.kernel_hashes): every kernel of the library
    has one; it depends on the kernel's instructions, its descriptor (registers, LDS, scratch) and the functions it may call -- not on
    where the linker put it, and not on the other kernels"""
    import struct
    kh = srchash.kernel_hashes()
    for k in ("bjj_k_mul_fixed_base", "bjj_k_mul_fixed_base_2x256", "bjj_k_mul_fixed_base_c32", "bjj_k_mul_var_base", "bjj_k_mul_var_base_tiles",
              "bjj_k_mul_var_base_exact", "bjj_k_var_base_scan", "bjj_k_eddsa_verify_groups", "bjj_k_poseidon5", "bjj_k_sign", "bjj_k_sign_c64"):
        assert k in kh and len(kh[k]) == 16, k
    assert len(set(kh.values())) == len(kh)                       # no two kernels share a fingerprint
    assert srchash.kernel_hashes() == kh and srchash.kernel_hash("bjj_k_poseidon5") == kh["bjj_k_poseidon5"]
    assert srchash.kernel_hashes("/nonexistent/libbjj_hip.so") == {} and srchash.kernel_hash("no_such_kernel") is None
    # one flipped instruction byte of ONE kernel changes that kernel's fingerprint and no other's; a flipped code-entry offset changes none
    data = bytearray(open(srchash.LIB, "rb").read())
    target, found = "bjj_k_poseidon5", None
    for off, size in srchash._code_objects(bytes(data)):
        img = bytes(data[off:off + size])
        funcs, objs = srchash._elf_functions(img)
        for n, body in funcs.items():
            if srchash._plain(n) == target and n + ".kd" in objs:
                found = (off + img.index(body), off + img.index(objs[n + ".kd"]))
    assert found
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "lib.so")
        mod = bytearray(data)
        mod[found[0] + 64] ^= 1
        open(p, "wb").write(mod)
        k2 = srchash.kernel_hashes(p)
        assert k2[target] != kh[target] and all(k2[k] == kh[k] for k in kh if k != target)
        mod = bytearray(data)
        mod[found[1] + 16] ^= 0x40                                # KERNEL_CODE_ENTRY_BYTE_OFFSET of the descriptor
        open(p, "wb").write(mod)
        assert srchash.kernel_hashes(p) == kh
        mod = bytearray(data)
        mod[found[1] + 48] ^= 1                                   # compute_pgm_rsrc1 (register counts): a different kernel
        open(p, "wb").write(mod)
        assert srchash.kernel_hashes(p)[target] != kh[target]


def test_bench_goes_by_the_kernel_code_fingerprint_when_an_entry_has_one(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    now = {"bjj_k_mul_fixed_base": "aaaaaaaaaaaaaaaa"}
    prof = {"hbm_traffic.json": {"fixed_base": {"bytes_per_launch": 1.5e9, "source": "profiles/x.md", "batch": 1 << 20, "kernel": "bjj_k_mul_fixed_base",
                                                "valu_insts_per_launch": 2.5e8, "window_bits": 28, "source_hash": "an-older-tree", "code_hash": "aaaaaaaaaaaaaaaa"}},
            "isa_mix.json": {"_source_hash": "an-older-tree",
                             "bjj_k_mul_fixed_base": {"avg_issue_cycles_per_valu_inst": 4.0, "quarter_rate_share": 0.76, "code_hash": "aaaaaaaaaaaaaaaa"}}}
    monkeypatch.setattr(bench, "load_profile_json", lambda name: prof.get(name, {}))
    monkeypatch.setattr(bench, "kernel_hash_now", lambda k: now.get(k))
    monkeypatch.setattr(bench, "source_hash_now", lambda: "this-tree")
    # the tree has moved on (another unit was edited), the kernel's code has not: its counters are still quoted
    r = bench.roofline_block("fixed_base", 0.6, 1 << 20, _Info(), "bjj_k_mul_fixed_base")
    assert r["traffic"] == 1.5e9 and "stale_profile" not in r and "kernel code fingerprint aaaaaaaaaaaaaaaa" in r["traffic_source"]
    assert bench.valu_block("fixed_base", 0.6, 1 << 20, _Info())["frac"] > 0
    # the kernel's code changed: stale, whatever the tree fingerprint says
    now["bjj_k_mul_fixed_base"] = "bbbbbbbbbbbbbbbb"
    monkeypatch.setattr(bench, "source_hash_now", lambda: "an-older-tree")
    r = bench.roofline_block("fixed_base", 0.6, 1 << 20, _Info(), "bjj_k_mul_fixed_base")
    assert r["traffic"] is None and r["stale_profile"] is True
    assert bench.valu_block("fixed_base", 0.6, 1 << 20, _Info())["stale_profile"] is True


def test_a_fast_math_build_does_not_compile(tmp_path):
    """euclid_partial_step decides an exact-integer path with an IEEE f64 division: -ffast-math in an overriding CXXFLAGS must
    fail the build instead of changing verdicts silently (the Makefile passes -fno-fast-math explicitly)."""
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("no hipcc")
    src = tmp_path / "t.hip"
    src.write_text('#include <hip/hip_runtime.h>\n#include "%s"\n' % os.path.join(srchash.CSRC, "bjj_device.hpp"))
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fsyntax-only", str(src)]
    ok = subprocess.run(base, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert ok.returncode == 0, ok.stdout[-2000:]
    bad = subprocess.run(base + ["-ffast-math"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert bad.returncode != 0 and "build without -ffast-math" in bad.stdout
    assert "-fno-fast-math" in open(os.path.join(srchash.CSRC, "Makefile")).read()
