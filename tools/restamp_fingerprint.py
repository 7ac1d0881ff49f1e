#!/usr/bin/env python3
"""Developer tool (run once, round 5): the source fingerprint changed from "raw bytes of csrc/" to "csrc/ without comments"
(babyjubjub-rs_amd/srchash.py).  Counters stored under the OLD fingerprint of exactly this tree are re-stamped with the new
one; anything stored under another fingerprint is left alone (it stays stale).  The proof that nothing but the stamp changes:
the old-style fingerprint of the working tree is recomputed here and must equal the stored one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd"))
import srchash  # noqa: E402

old, new = srchash.legacy_tree_hash(), srchash.tree_hash()
print("tree: raw-bytes fingerprint %s, comment-stripped fingerprint %s" % (old, new))
n = 0
p = os.path.join(ROOT, "profiles", "hbm_traffic.json")
d = json.load(open(p))
for k, v in d.items():
    if isinstance(v, dict) and v.get("source_hash") == old:
        v["source_hash"], v["source_hash_raw_bytes"] = new, old
        n += 1
json.dump(d, open(p, "w"), indent=1)
p = os.path.join(ROOT, "profiles", "isa_mix.json")
d = json.load(open(p))
if d.get("_source_hash") == old:
    d["_source_hash"], d["_source_hash_raw_bytes"] = new, old
    n += 1
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
print("%d entries re-stamped" % n)
