#!/usr/bin/env python3
"""Developer tool: the randomised differential soak of tests/test_gpu_soak.py over many more seeds.
usage: tools/soak_more.py [first_seed [count]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest, test_gpu_soak
import babyjubjub_rs_amd as bjj
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
orc = conftest.Oracle()
ctx = bjj.Context(0, int(os.environ.get("W", "0")))   # default table (23 bits); W=28 for the benchmark table
bad = 0
for seed in range(first, first + count):
    try:
        test_gpu_soak.test_soak_all_entry_points(ctx, orc, seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, e)
print("extended soak: seeds %d..%d, failures: %d" % (first, first + count - 1, bad))
sys.exit(1 if bad else 0)
