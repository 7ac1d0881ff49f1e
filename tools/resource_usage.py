#!/usr/bin/env python3
"""Developer tool: `make resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage over the kernel units) as the table of
profiles/<round>_resource_usage.txt.   usage: tools/resource_usage.py r03 > profiles/r03_resource_usage.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-s", "resource-usage"], cwd=os.path.join(ROOT, "babyjubjub-rs_amd", "csrc"), stdout=subprocess.PIPE,
                     stderr=subprocess.STDOUT, text=True).stdout
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        d = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = {"name": d.split("(")[0]}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
rows = [r for r in rows if r["name"].startswith("bjj_k_")]
print("# Per-kernel resources of the shipped build (hipcc -Rpass-analysis=kernel-resource-usage, `make resource-usage` in")
print("# babyjubjub-rs_amd/csrc; gfx950; tools/resource_usage.py).  VGPRs = registers the compiler ALLOCATES per lane (what bounds")
print("# occupancy: 512 per SIMD lane slot budget, so <= 128 -> 4 waves/SIMD, <= 168 -> 3, <= 256 -> 2).  rocprofv3's dispatch field")
print("# `VGPR_Count` (quoted in the *_summary.md files) is about HALF of this number on gfx950 (K1: 112 vs 223) -- size occupancy")
print("# from THIS table.")
print("%-36s %6s %6s %9s %7s %11s %11s %10s" % ("kernel", "VGPRs", "SGPRs", "scratch B", "w/SIMD", "VGPR spills", "SGPR spills", "LDS B/WG"))
for r in rows:
    print("%-36s %6s %6s %9s %7s %11s %11s %10s" % (r["name"], r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                                     r.get("Occupancy [waves/SIMD]", "?"), r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"),
                                                     r.get("LDS Size [bytes/block]", "?")))
