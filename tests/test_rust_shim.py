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
                 "pub fn verify_batch(", "pub fn mul_scalar_batch(", "pub static ref Q: BigInt", "pub mod utils;",
                 # semantics that cannot be exercised without a compiler, pinned textually: B8.mul_scalar(&s) uses |s| (the
                 # reference drops the sign, lib.rs:156), a negative msg panics where the reference's from_str().unwrap() does
                 "let s_red = b8_scalar(&s);", "let (_, mag) = n.clone().into_parts();", "non_negative(&msg); // :399",
                 "non_negative(&msg); // lib.rs:321"):
        assert item in api, item


def test_header_is_plain_c11():
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "bjj_hip.h")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


# ---- the crate's public item list against the reference's (tests/golden/reference_pub_api.json) -------------------------
def _pub_items(path):
    """same normalisation as tests/golden/make_reference_api.py"""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), txt, flags=re.S)
    out = []
    for line in txt.splitlines():
        s = line.strip()
        if not s.startswith("pub ") or s.startswith("//"):
            continue
        s = s.split("//")[0].strip()
        if s.startswith("pub static ref"):
            s = s.split(" = ")[0].split("=")[0]
        s = s.split("{")[0].strip().rstrip(",;").strip()
        out.append(re.sub(r"\s+", " ", s))
    return out


def test_every_public_item_of_the_reference_exists_with_the_same_signature():
    """`use babyjubjub_rs::{Q, utils::modinv, ..}` must keep compiling against the shim: every `pub` item of the reference's
    src/lib.rs and src/utils.rs (fixture: names + signatures, generated by tests/golden/make_reference_api.py) has to appear
    in rust/src/babyjubjub_hip.rs resp. rust/src/utils.rs with an identical signature string.  Allow-list: empty."""
    import collections
    import json
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_pub_api.json")))
    allow = set()
    for ref_file, mine in (("lib.rs", "babyjubjub_hip.rs"), ("utils.rs", "utils.rs")):
        have = collections.Counter(_pub_items(os.path.join(ROOT, "rust", "src", mine)))
        want = collections.Counter(i["item"] for i in api[ref_file])
        missing = sorted(k for k in want if k not in allow and have[k] < want[k])      # multiset: `pub x: Fr` occurs per struct
        assert missing == [], (mine, missing)
    assert len(api["lib.rs"]) >= 34 and len(api["utils.rs"]) == 6


def test_reference_api_fixture_is_current():
    """where the reference checkout exists (the build container), the fixture must equal a fresh extraction"""
    import json
    import pytest
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no reference checkout on this machine (the fixture is what travels)")
    p = os.path.join(ROOT, "tests", "golden", "reference_pub_api.json")
    before = open(p).read()
    subprocess.run(["python3", os.path.join(ROOT, "tests", "golden", "make_reference_api.py")], check=True, stdout=subprocess.DEVNULL)
    assert open(p).read() == before
    assert json.loads(before)["lib.rs"][2]["item"] == "pub static ref Q: BigInt"


# ---- rust/src/utils.rs: a line-for-line Python model of its arithmetic, pinned to the reference's vectors and the oracle ----
def _modulus(a, m):
    r = abs(a) % abs(m) * (1 if a >= 0 else -1)         # Rust `%`: truncated, sign of the dividend
    return r + m if r != 0 and (r < 0) != (m < 0) else r


def _modinv(a, q):
    if a == 0:
        return None
    r0, r1, t0, t1 = q, a, 0, 1
    while r1 != 0:
        quot = abs(r0) // abs(r1) * (1 if (r0 >= 0) == (r1 >= 0) else -1)      # truncated division
        t2 = t0 - quot * t1
        r0, r1, t0, t1 = r1, _modulus(r0, r1), t1, t2
    return _modulus(t0, q) if t0 < 0 else t0


def _legendre(a, q):
    return -1 if pow(a, (q - 1) >> 1, q) == q - 1 else 1


def _tonelli(a, q):
    if _legendre(a, q) != 1 or a == 0 or q == 2:
        return None
    if q % 4 == 3:
        return pow(a, (q + 1) >> 2, q)
    s, r = q - 1, 0
    while s % 2 == 0:
        s >>= 1
        r += 1
    n = 2
    while _legendre(n, q) != -1:
        n += 1
    x, b, z = pow(a, (s + 1) >> 1, q), pow(a, s, q), pow(n, s, q)
    if b == 0:
        return None
    while True:
        m, t = 0, b
        while t != 1:
            t = t * t % q
            m += 1
        if m == 0:
            return x
        w = z
        for _ in range(r - m - 1):
            w = w * w % q
        z = w * w % q
        x = x * w % q
        b = b * z % q
        r = m


def test_rust_utils_model_matches_reference_vectors_and_oracle(pyoracle):
    """The model above follows rust/src/utils.rs statement by statement (the crate cannot be compiled here).  Pins: the
    reference's own vectors (src/utils.rs:229-258: modinv 641883, the 2^252-ish square root -- which ROOT comes out is part
    of the contract), the oracle's restatement of modsqrt over F_r on random inputs, and the algebra."""
    import random
    src = open(os.path.join(ROOT, "rust", "src", "utils.rs")).read()
    for needle in ("pub fn modulus(a: &BigInt, m: &BigInt) -> BigInt", "pub fn modinv(a: &BigInt, q: &BigInt) -> Result<BigInt, String>",
                   '"no mod inv of Zero"', '"not a mod p square"', "let quot = &r0 / &r1;", "let mut n = two.clone();",
                   "for _ in 0..(r - m - 1)", "pub fn concatenate_arrays<T: Clone>(x: &[T], y: &[T]) -> Vec<T>"):
        assert needle in src, needle
    assert _modinv(123456789123456789123456789123456789123456789, 12345678) == 641883
    q2 = 7237005577332262213973186563042994240857116359379907606001950938285454250989
    a2 = 6536923810004159332831702809452452174451353762940761092345538667656658715568
    assert _tonelli(a2, q2) == 5464794816676661649783249706827271879994893912039750480019443499440603127256
    rng = random.Random(7)
    Q = pyoracle.Q
    for _ in range(300):
        a = rng.randrange(Q)
        want = pyoracle.modsqrt(a)
        got = _tonelli(a, Q)
        assert got == want and (got is None or got * got % Q == a)
        inv = _modinv(a, Q) if a else None
        assert a == 0 or (0 <= inv < Q and inv * a % Q == 1)
    for a, m in ((5, 3), (-5, 3), (5, -3), (-5, -3), (6, -3), (0, 7), (-7, 7)):
        assert _modulus(a, m) == a % m                     # Python's % has the sign of the modulus as well
    assert _tonelli(0, Q) is None and _tonelli(Q, Q) is None and _modinv(0, Q) is None
    assert _tonelli(2, 7) == pow(2, 2, 7) and _legendre(0, 7) == 1 and _legendre(3, 7) == -1


# ---- a lexer-level check of the uncompiled sources: the nearest thing to `cargo check` this image allows --------------------
def _rust_tokens(src):
    """Rust tokens good enough to balance delimiters: skips // and nested /* */ comments, "..." / r#"..."# / b"..." strings,
    char literals vs lifetimes.  Yields (kind, text, line)."""
    i, n, line = 0, len(src), 1
    while i < n:
        c = src[i]
        if c == "\n":
            line += 1; i += 1
        elif c.isspace():
            i += 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            depth, j = 1, i + 2
            while depth and j < n:
                if src.startswith("/*", j):
                    depth += 1; j += 2
                elif src.startswith("*/", j):
                    depth -= 1; j += 2
                else:
                    line += src[j] == "\n"; j += 1
            assert depth == 0, "unterminated block comment from line %d" % line
            i = j
        elif c == '"' or (c in "rb" and re.match(r'(?:b?r#*"|b")', src[i:])):
            m = re.match(r'(b?)(r(#*))?"', src[i:])
            raw, hashes = m.group(2) is not None, m.group(3) or ""
            j = i + m.end()
            while True:
                assert j < n, "unterminated string from line %d" % line
                if not raw and src[j] == "\\":
                    j += 2; continue
                if src[j] == '"' and src.startswith(hashes, j + 1):
                    j += 1 + len(hashes); break
                line += src[j] == "\n"; j += 1
            yield ("str", src[i:j], line)
            i = j
        elif c == "'":
            m = re.match(r"'(?:\\(?:x[0-9a-fA-F]{2}|u\{[0-9a-fA-F]+\}|.)|[^\\'])'", src[i:])
            if m:
                yield ("char", m.group(0), line); i += m.end()
            else:
                m = re.match(r"'[A-Za-z_]\w*", src[i:])
                assert m, "stray ' on line %d" % line
                yield ("lifetime", m.group(0), line); i += m.end()
        elif c.isalpha() or c == "_":
            m = re.match(r"\w+", src[i:])
            yield ("ident", m.group(0), line); i += m.end()
        elif c.isdigit():
            m = re.match(r"\d[\w.]*", src[i:])
            yield ("num", m.group(0), line); i += m.end()
        else:
            yield ("punct", c, line); i += 1


def test_rust_sources_lex_and_balance():
    """every .rs file and build.rs: comments / strings / chars terminate, (), [], {} nest properly, every `fn` item has a body
    or a `;`, no `let` without `;`-terminated statement on the same nesting level, `unsafe` only before a block / fn / impl"""
    close = {")": "(", "]": "[", "}": "{"}
    files = [os.path.join(ROOT, "rust", "build.rs")] + [os.path.join(ROOT, "rust", "src", f)
                                                        for f in sorted(os.listdir(os.path.join(ROOT, "rust", "src")))]
    for path in files:
        toks = list(_rust_tokens(open(path).read()))
        stack = []
        for k, (kind, t, line) in enumerate(toks):
            if kind != "punct":
                continue
            if t in "([{":
                stack.append((t, line))
            elif t in ")]}":
                assert stack and stack[-1][0] == close[t], "%s:%d: unbalanced %r" % (path, line, t)
                stack.pop()
        assert not stack, "%s: unclosed %r from line %d" % (path, stack[-1][0], stack[-1][1])
        for k, (kind, t, line) in enumerate(toks):
            if kind == "ident" and t == "unsafe":
                nxt = toks[k + 1][1]
                assert nxt in ("{", "fn", "impl", "extern"), "%s:%d: `unsafe %s`" % (path, line, nxt)
            if kind == "ident" and t == "fn" and toks[k + 1][0] == "ident":
                # scan to the end of the signature: the first `{` or `;` outside parentheses / angle-free brackets
                depth, j = 0, k + 2
                while True:
                    assert j < len(toks), "%s:%d: fn %s has neither a body nor a `;`" % (path, line, toks[k + 1][1])
                    tt = toks[j][1]
                    if toks[j][0] == "punct":
                        if tt in "([":
                            depth += 1
                        elif tt in ")]":
                            depth -= 1
                        elif depth == 0 and tt in "{;":
                            break
                    j += 1


def test_rust_lexer_rejects_broken_sources():
    import pytest
    for bad in ('fn f() { let s = "abc; }', "fn f() { /* open", "fn f() { g(1, 2]; }"):
        with pytest.raises(AssertionError):
            toks = list(_rust_tokens(bad))
            stack = []
            for kind, t, line in toks:
                if kind == "punct" and t in "([{":
                    stack.append(t)
                elif kind == "punct" and t in ")]}":
                    assert stack and stack.pop() == {")": "(", "]": "[", "}": "{"}[t]
            assert not stack
    assert [t for k, t, _ in _rust_tokens("impl<'a> X<'a> { fn c() -> char { '}' } }") if k in ("char", "lifetime")] == ["'a", "'a", "'}'"]
