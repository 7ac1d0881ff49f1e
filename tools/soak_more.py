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
bad = 0
# every kernel form: the per-call defaults of a one-stream caller (K1 one workgroup per CU, K2 tiles, verify groups), then the
# forms a caller only gets while launches overlap or for very large batches, forced through the environment
# ...; the second pass also runs the signer in its constant-time form (bjj_set_signer_constant_time)
for forms in ({}, {"BJJ_K1_VARIANT": "1", "BJJ_K2_VARIANT": "0", "BJJ_VERIFY_DISPATCH": "0"}):
    os.environ.update(forms)
    ctx = bjj.Context(0, int(os.environ.get("W", "0")))   # default table (23 bits); W=28 for the benchmark table
    for k in forms:
        del os.environ[k]
    if forms:
        ctx.set_signer_constant_time(True)
        forms = dict(forms, signer_constant_time=1)
    fails = 0
    for seed in range(first, first + count):
        try:
            test_gpu_soak.test_soak_all_entry_points(ctx, orc, seed)
        except AssertionError as e:
            fails += 1
            print("FAIL seed", seed, forms, e)
    ctx.close()
    print("extended soak, kernel forms %s: seeds %d..%d, failures: %d" % (forms or "default", first, first + count - 1, fails))
    bad += fails
sys.exit(1 if bad else 0)
