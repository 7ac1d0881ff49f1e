"""The Rust shim crate (rust/) cannot be compiled in this image (no cargo / rustc), so its FFI block
is checked mechanically against the C header: same function set, same arity, same pointer / usize / int kinds per
argument and per return value, same struct layout for bjj_info -- and the header itself must be plain C11.  CPU only."""
import os
import re
import subprocess

from conftest import ROOT

KIND_C = {"int": "int", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "double": "f64"}
KIND_RS = {"c_int": "int", "usize": "usize", "u32": "u32", "u64": "u64", "f64": "f64"}


def c_functions():
    txt = open(os.path.join(ROOT, "include", "bjj_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
    fns = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(bjj_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()

        def kind(t):
            t = t.strip()
            if "*" in t:
                return "ptr"
            base = [w for w in re.sub(r"\bconst\b", "", t).split() if w][0]
            return KIND_C[base]
        ps = [] if params in ("", "void") else [kind(p) for p in params.split(",")]
        fns[name] = (ps, "void" if ret == "void" else kind(ret + " "))
    return fns


def rust_functions():
    txt = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', txt, flags=re.S).group(1)
    fns = {}
    for m in re.finditer(r"pub fn (bjj_[a-z0-9_]+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block):
        name, params, ret = m.group(1), m.group(2).strip(), m.group(3)

        def kind(t):
            t = t.strip()
            return "ptr" if t.startswith("*") else KIND_RS[t]
        ps = [kind(p.split(":", 1)[1]) for p in params.split(",")] if params else []
        fns[name] = (ps, "void" if ret is None else kind(ret))
    return fns


def test_rust_extern_block_matches_the_c_header():
    c, r = c_functions(), rust_functions()
    assert len(c) >= 50
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert c[name] == r[name], (name, c[name], r[name])


def test_rust_info_struct_matches_the_c_struct():
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "bjj_hip.h")).read(), flags=re.S)
    body = re.search(r"typedef struct \{(.*?)\} bjj_info;", h, flags=re.S).group(1)
    c_fields = [(("ptr" if "*" in f else KIND_C[[w for w in f.replace("const", "").split() if w][0]]), f.split()[-1].lstrip("*"))
                for f in (x.strip() for x in body.split(";")) if f]
    rs = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    rbody = re.search(r"pub struct BjjInfo \{(.*?)\}", rs, flags=re.S).group(1)
    r_fields = [(("ptr" if t.strip().startswith("*") else KIND_RS[t.strip()]), n.strip())
                for n, t in (x.replace("pub ", "").split(":") for x in rbody.split(",") if ":" in x)]
    assert c_fields == r_fields


def test_generated_ffi_is_current():
    """rust/src/ffi.rs is generated from the header: regenerating must not change it"""
    p = os.path.join(ROOT, "rust", "src", "ffi.rs")
    before = open(p).read()
    subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_rust_ffi.py")], check=True, stdout=subprocess.DEVNULL)
    assert open(p).read() == before


def test_rust_wrappers_only_call_declared_functions():
    declared = set(rust_functions())
    for f in ("gpu.rs", "multi.rs", "babyjubjub_hip.rs"):
        src = open(os.path.join(ROOT, "rust", "src", f)).read()
        used = set(re.findall(r"ffi::(bjj_[a-z0-9_]+)", src))
        assert used <= declared, (f, used - declared)
    api = open(os.path.join(ROOT, "rust", "src", "babyjubjub_hip.rs")).read()
    for item in ("pub struct Point", "pub struct PointProjective", "pub struct Signature", "pub struct PrivateKey",
                 "pub fn mul_scalar(&self, n: &BigInt) -> Point", "pub fn public(&self) -> Point",
                 "pub fn verify(pk: Point, sig: Signature, msg: BigInt) -> bool", "pub fn decompress_point(bb: [u8; 32]) -> Result<Point, String>",
                 "pub fn sign(&self, msg: BigInt) -> Result<Signature, String>", "pub fn new_key() -> PrivateKey",
                 "pub fn add(&self, q: &PointProjective) -> PointProjective", "pub fn affine(&self) -> Point",
                 "pub fn verify_batch(", "pub fn mul_scalar_batch("):
        assert item in api, item


def test_header_is_plain_c11():
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "bjj_hip.h")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
