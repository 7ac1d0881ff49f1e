"""Fingerprint of the sources the shipped library is built from.

The static counters that bench.py quotes (profiles/hbm_traffic.json: HBM bytes and VALU instructions per launch from the
rocprofv3 PMC passes; profiles/isa_mix.json: the kernels' static instruction mix) describe ONE build.  The tools that write
those files store this fingerprint next to the numbers, and bench.py compares it with the tree it runs from: when they differ
the counters are not quoted (`traffic: null`, `stale_profile: true`) instead of silently describing another build.
The fingerprint is taken over the sources WITHOUT their comments: what reaches the compiler."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
PATTERNS = ("*.hip", "*.hpp", "*.inc", "Makefile")


def strip_comments(text):
    """C / C++ source text without its comments (string and character literals respected), lines right-stripped and
    whitespace-only lines dropped: what the compiler sees, up to layout.  A comment-only edit -- correcting a sentence, adding a
    measurement to a header comment -- does not change the code objects and must not invalidate the committed counters
    (round 5 re-profiled three times for sentences)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == "/" and i + 1 < n and text[i + 1] == "/":
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif c == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        elif c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        else:
            out.append(c)
            i += 1
    lines = (l.rstrip() for l in "".join(out).split("\n"))
    return "\n".join(l for l in lines if l)


def tree_hash(csrc=CSRC):
    """sha256 (first 16 hex digits) over the names and the comment-stripped contents of csrc/*.hip, *.hpp, *.inc, and the
    Makefile as it is (compiler flags)"""
    h = hashlib.sha256()
    files = sorted(f for pat in PATTERNS for f in glob.glob(os.path.join(csrc, pat)))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            data = fh.read()
        if not f.endswith("Makefile"):
            data = strip_comments(data.decode("utf-8", "replace")).encode()
        h.update(data)
        h.update(b"\0")
    return h.hexdigest()[:16]


def legacy_tree_hash(csrc=CSRC):
    """the fingerprint of rounds 3-5 (raw file contents): only used to re-stamp files that were written with it"""
    h = hashlib.sha256()
    files = sorted(f for pat in PATTERNS for f in glob.glob(os.path.join(csrc, pat)))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(tree_hash())


# ---------------------------------------------------------------------------------------------------------------------------
# Per-kernel fingerprint of the SHIPPED code (round 6).  The tree hash above changes with every edit of csrc/: an edit of the
# variable-base unit would stale the verify kernel's counters although its machine code did not change.  What the counters
# describe is machine code, so that is what is fingerprinted: libbjj_hip.so carries one clang offload bundle per kernel
# translation unit; the gfx950 code object in it is an ELF whose .symtab names every kernel (FUNC), its 64-byte kernel
# descriptor (<name>.kd: register counts, LDS, scratch) and the device functions that were not inlined.
#   code_hash(kernel) = sha256( the kernel's instruction bytes  +  its descriptor without the code-entry offset
#                               +  every non-kernel function of its code object )
# (the last term is conservative: a kernel may call any of them).  Comment edits, new kernels next to it, and edits of other units
# leave it alone; any change of its own instructions, of its resource usage or of a callee changes it.
# ---------------------------------------------------------------------------------------------------------------------------
import re
import struct

LIB = os.path.join(CSRC, "libbjj_hip.so")
_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(data):
    """the gfx950 ELF images inside a fat binary / shared library: [(offset, size)]"""
    out = []
    for m in re.finditer(re.escape(_BUNDLE_MAGIC), data):
        b = m.start()
        (count,) = struct.unpack_from("<Q", data, b + 24)
        p = b + 32
        if count > 64:
            continue
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen]
            p += 24 + tlen
            if b"gfx950" in triple and size and data[b + off:b + off + 4] == b"\x7fELF":
                out.append((b + off, size))
    return out


def _elf_functions(img):
    """(functions, objects) of an ELF64 little-endian image: name -> bytes, from .symtab"""
    e_shoff, = struct.unpack_from("<Q", img, 0x28)
    e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", img, 0x3A)
    sh = [struct.unpack_from("<IIQQQQIIQQ", img, e_shoff + i * e_shentsize) for i in range(e_shnum)]   # name type flags addr offset size link info align entsize
    funcs, objs = {}, {}
    for s in sh:
        if s[1] != 2:      # SHT_SYMTAB
            continue
        strtab = sh[s[6]]
        for k in range(s[5] // 24):
            st_name, st_info, _other, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", img, s[4] + k * 24)
            if st_shndx == 0 or st_shndx >= e_shnum or not st_size:
                continue
            end = img.index(b"\0", strtab[4] + st_name)
            name = img[strtab[4] + st_name:end].decode()
            sec = sh[st_shndx]
            if sec[1] == 8:     # SHT_NOBITS
                continue
            body = img[sec[4] + st_value - sec[3]:sec[4] + st_value - sec[3] + st_size]
            if st_info & 15 == 2:
                funcs[name] = body
            elif st_info & 15 == 1:
                objs[name] = body
    return funcs, objs


def _plain(mangled):
    """_Z20bjj_k_mul_fixed_basePKj... -> bjj_k_mul_fixed_base"""
    m = re.match(r"_Z(\d+)", mangled)
    return mangled[m.end():m.end() + int(m.group(1))] if m else mangled


def kernel_hashes(lib=LIB):
    """{kernel name: 16 hex digits} for every kernel of the library (see above); {} when the library cannot be read"""
    try:
        data = open(lib, "rb").read()
    except OSError:
        return {}
    out = {}
    for off, size in _code_objects(data):
        funcs, objs = _elf_functions(data[off:off + size])
        kernels = {n for n in funcs if n + ".kd" in objs}
        helpers = hashlib.sha256()
        for n in sorted(set(funcs) - kernels):
            helpers.update(n.encode() + b"\0" + funcs[n] + b"\0")
        for n in kernels:
            h = hashlib.sha256()
            kd = objs[n + ".kd"]
            kd = kd[:16] + b"\0" * 8 + kd[24:]      # KERNEL_CODE_ENTRY_BYTE_OFFSET: where the linker put the code, not what it is
            h.update(funcs[n] + b"\0" + kd + b"\0" + helpers.digest())
            out[_plain(n)] = h.hexdigest()[:16]
    return out


def kernel_hash(name, lib=LIB):
    return kernel_hashes(lib).get(name)
