#!/usr/bin/env python3
"""Developer tool (round 6): the committed counters (profiles/hbm_traffic.json, profiles/isa_mix.json) were stamped with the fingerprint
of the whole source tree; bench.py now goes by the fingerprint of each kernel's MACHINE CODE (srchash.kernel_hashes).  This tool gives
entries taken under a tree fingerprint their kernels' code fingerprints -- from a library BUILT FROM THAT TREE:

    git archive <commit> babyjubjub-rs_amd/csrc include | tar -x -C <dir>;  make -C <dir>/babyjubjub-rs_amd/csrc
    tools/restamp_code_hash.py <dir>

The proof that the library is the profiled build: the tree fingerprint of <dir>'s csrc is recomputed and must equal the stored one;
entries stored under another fingerprint are left alone.  Whether a stamped entry still describes what ships is then decided kernel by
kernel, by comparing machine code (bench.py: profile_entry_is_current)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd"))
import srchash  # noqa: E402

tree = sys.argv[1]
csrc = os.path.join(tree, "babyjubjub-rs_amd", "csrc")
th = srchash.tree_hash(csrc)
kh = srchash.kernel_hashes(os.path.join(csrc, "libbjj_hip.so"))
now = srchash.kernel_hashes()
print("tree fingerprint of %s: %s; %d kernels in its library" % (tree, th, len(kh)))
n = 0
p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
d = json.load(open(p))
for k, v in d.items():
    if isinstance(v, dict) and v.get("source_hash") == th and v.get("kernel") in kh:
        v["code_hash"] = kh[v["kernel"]]
        n += 1
        print("  %-24s %-30s %s  %s" % (k, v["kernel"], v["code_hash"], "== the library in this tree" if now.get(v["kernel"]) == v["code_hash"] else "differs from this tree: stale"))
json.dump(d, open(p, "w"), indent=1)
p = os.path.join(ROOT, "profiles", "isa_mix.json")
d = json.load(open(p))
if d.get("_source_hash") == th:
    for k, v in d.items():
        if isinstance(v, dict) and k in kh:
            v["code_hash"] = kh[k]
            n += 1
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
print("%d entries stamped" % n)
