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
