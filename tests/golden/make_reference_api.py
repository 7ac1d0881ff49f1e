#!/usr/bin/env python3
"""Writes tests/golden/reference_pub_api.json: the PUBLIC ITEM LIST of the reference crate (names and signatures only --
interface facts, no bodies) as `grep -n "pub " src/lib.rs src/utils.rs` shows it, comment blocks and test modules
excluded.  tests/test_rust_shim.py holds the Rust shim crate against this list.  Run in the build container (the
reference checkout does not travel to the GPU box): python3 tests/golden/make_reference_api.py"""
import json
import os
import re

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def pub_items(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), txt, flags=re.S)      # block comments (keep line numbers)
    cut = txt.find("#[cfg(test)]")
    if cut >= 0:
        txt = txt[:cut]
    out = []
    for no, line in enumerate(txt.splitlines(), 1):
        s = line.strip()
        if not s.startswith("pub ") or s.startswith("//"):
            continue
        s = s.split("//")[0].strip()
        if s.startswith("pub static ref"):
            s = s.split(" = ")[0]
        s = s.split("{")[0].strip().rstrip(",;").strip()
        out.append({"line": no, "item": re.sub(r"\s+", " ", s)})
    return out


def main():
    api = {"crate": "arnaucube/babyjubjub-rs v0.0.11", "lib.rs": pub_items(os.path.join(REF, "lib.rs")),
           "utils.rs": pub_items(os.path.join(REF, "utils.rs"))}
    with open(os.path.join(HERE, "reference_pub_api.json"), "w") as f:
        json.dump(api, f, indent=1)
    print("lib.rs: %d public items, utils.rs: %d" % (len(api["lib.rs"]), len(api["utils.rs"])))


if __name__ == "__main__":
    main()
