#!/usr/bin/env python3
"""Developer tool: bjj_init / bjj_check_table wall time per window width on a GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import babyjubjub_rs_amd as bjj
for W in [int(a) for a in sys.argv[1:]] or [16, 21, 23]:
    t0 = time.time(); ctx = bjj.Context(0, W); t1 = time.time()
    bad = ctx.check_table(); t2 = time.time()
    print("W=%2d  table %9.1f MB  init %.3f s  check_table %.3f s  bad=%d" % (W, ctx.info().table_bytes / 1e6, t1 - t0, t2 - t1, bad))
    ctx.close()
