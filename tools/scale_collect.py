#!/usr/bin/env python3
"""Developer tool: copies the products of tools/scale_session.sh (gpurun_out/<tag>_scale/products/) into profiles/ and prints one
line per step.  usage: tools/scale_collect.py r05"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag + "_scale", "products")
for f in sorted(glob.glob(os.path.join(src, "*"))):
    name = os.path.basename(f)
    if not name.startswith(tag + "_scale"):
        name = "%s_scale_%s" % (tag, name)
    shutil.copy(f, os.path.join(ROOT, "profiles", name))
    print("profiles/" + name, os.path.getsize(f), "bytes")
