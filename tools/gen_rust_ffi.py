#!/usr/bin/env python3
"""Developer tool: writes rust/src/ffi.rs -- the extern "C" block of the Rust shim -- from include/bjj_hip.h, so that the
two cannot drift (tests/test_rust_shim.py re-derives the comparison independently)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_decls(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    out = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(bjj_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        ps = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        out.append((ret, name, ps))
    return out


def rust_type(ctype):
    t = re.sub(r"\s+", " ", ctype.strip())
    t = re.sub(r"\b[a-z_][a-z0-9_]*$", "", t).strip() if not t.endswith("*") and " " in t else t   # drop the parameter name
    const = "const " in t or t.startswith("const")
    base = t.replace("const", "").strip()
    stars = base.count("*")
    base = base.replace("*", "").strip()
    m = {"void": "c_void", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "int": "c_int", "size_t": "usize", "char": "c_char", "double": "f64",
         "bjj_ctx": "BjjCtx", "bjj_multi": "BjjMulti", "bjj_info": "BjjInfo"}[base]
    for _ in range(stars):
        m = ("*const " if const and _ == 0 else "*mut ") + m
    return m


def param(p):
    p = re.sub(r"\s+", " ", p)
    m = re.match(r"^(.*?)([A-Za-z_]\w*)$", p)
    ctype, name = m.group(1).strip(), m.group(2)
    name = {"in": "input", "type": "kind", "ref": "reference", "move": "mv", "fn": "func"}.get(name, name)   # Rust keywords
    return name, rust_type(ctype)


def info_fields(text):
    """the fields of `bjj_info`, in order, as Rust declarations"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct \{(.*?)\} bjj_info;", text, flags=re.S).group(1)
    out = []
    for f in (x.strip() for x in body.split(";")):
        if f:
            name, ty = param(f)
            out.append("    pub %s: %s," % (name, ty))
    return out


def main():
    header = open(os.path.join(ROOT, "include", "bjj_hip.h")).read()
    decls = c_decls(header)
    lines = ['//! `extern "C"` declarations of libbjj_hip.so -- GENERATED from ../include/bjj_hip.h by tools/gen_rust_ffi.py.',
             "//! Do not edit by hand; `tests/test_rust_shim.py` compares this block with the header.",
             "#![allow(non_camel_case_types, dead_code)]",
             "use std::os::raw::{c_char, c_int, c_void};", "",
             "/// opaque `bjj_ctx` (one device + stream + fixed-base table)", "#[repr(C)]", "pub struct BjjCtx {", "    _private: [u8; 0],", "}",
             "/// opaque `bjj_multi` (one context per device of the node)", "#[repr(C)]", "pub struct BjjMulti {", "    _private: [u8; 0],", "}", "",
             "/// `bjj_info` (include/bjj_hip.h): set `struct_size = size_of::<BjjInfo>()` before `bjj_get_info`", "#[repr(C)]",
             "pub struct BjjInfo {",
             ] + info_fields(header) + ["}", "",
             "pub const BJJ_OK: c_int = 0;", "pub const BJJ_E_INVALID: c_int = -1;", "pub const BJJ_E_NO_DEVICE: c_int = -2;",
             "pub const BJJ_E_HIP: c_int = -3;", "pub const BJJ_E_NOMEM: c_int = -4;", "pub const BJJ_E_RCCL: c_int = -5;",
             "pub const BJJ_WINDOW_AUTO: c_int = -1;", "pub const BJJ_MAX_SCALAR_BYTES: usize = 4096;",
             "pub const BJJ_SCHNORR_NONCE_BYTES: usize = 128;", "pub const BJJ_SCHNORR_S_BYTES: usize = 160;",
             "pub const BJJ_TRANSPORT_RCCL: c_int = 0;", "pub const BJJ_TRANSPORT_PEER_COPY: c_int = 1;", "",
             'extern "C" {']
    for ret, name, ps in decls:
        args = ", ".join("%s: %s" % param(p) for p in ps)
        r = "" if ret == "void" else " -> " + rust_type(ret + " x" if not ret.endswith("*") else ret)
        lines.append("    pub fn %s(%s)%s;" % (name, args, r))
    lines += ["}", ""]
    open(os.path.join(ROOT, "rust", "src", "ffi.rs"), "w").write("\n".join(lines))
    print("wrote rust/src/ffi.rs: %d functions" % len(decls))


if __name__ == "__main__":
    main()
