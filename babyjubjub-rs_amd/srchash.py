"""Fingerprint of the sources the shipped library is built from.

The static counters that bench.py quotes (profiles/hbm_traffic.json: HBM bytes and VALU instructions per launch from the
rocprofv3 PMC passes; profiles/isa_mix.json: the kernels' static instruction mix) describe ONE build.  The tools that write
those files store this fingerprint next to the numbers, and bench.py compares it with the tree it runs from: when they differ
the counters are not quoted (`traffic: null`, `stale_profile: true`) instead of silently describing another build."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
PATTERNS = ("*.hip", "*.hpp", "*.inc", "Makefile")


def tree_hash(csrc=CSRC):
    """sha256 (first 16 hex digits) over the names and contents of csrc/*.hip, *.hpp, *.inc and the Makefile (compiler flags)"""
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
