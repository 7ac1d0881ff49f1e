#!/usr/bin/env python3
"""Developer tool: condense gpurun_out/<tag>/ (rocprofv3 csv output of tools/profile_r.sh) into
profiles/<tag>_<workload>_summary.md and update profiles/hbm_traffic.json (read by bench.py)."""
import collections, csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd"))
import srchash  # noqa: E402  (fingerprint of csrc/: bench.py refuses to quote counters of another build)
tag, wl = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
KERNEL = {"fixed_base": "bjj_k_mul_fixed_base", "var_base": "bjj_k_mul_var_base_tiles", "verify": "bjj_k_eddsa_verify_groups",
          "poseidon5": "bjj_k_poseidon5"}[wl]
out = ["# rocprofv3 summary — %s, workload %s" % (tag, wl), "",
       "Command (see tools/profile_r.sh): `rocprofv3 --output-format csv --kernel-trace --stats -- python3 bench.py "
       "--workload %s --no-cpu-baseline --no-also` (default steps / warm-up, as the bench line); counters in separate `--pmc` passes." % wl, ""]
stats = os.path.join(src, "trace_%s" % wl, "trace_kernel_stats.csv")
out += ["## kernel stats (`--kernel-trace --stats`)", "", "```"] + open(stats).read().strip().splitlines() + ["```", ""]
avg_ns = None
for r in csv.DictReader(open(stats)):
    if r["Name"].startswith(KERNEL + "("):
        avg_ns = float(r["AverageNs"])
counters = {}
meta = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    p = os.path.join(src, "%s_%s" % (sub, wl), "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r["Kernel_Name"].startswith(KERNEL + "("):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                      "Accum_VGPR_Count", "SGPR_Count")}
    for k, v in agg.items():
        counters[k] = (sum(v) / len(v), len(v))
alloc = ""
try:   # the compiler's own allocation for this kernel (profiles/<round>_resource_usage.txt from `make resource-usage`)
    ru = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_resource_usage.txt"))[-1]
    for line in open(os.path.join(ROOT, "profiles", ru)):
        f = line.split()
        if f and f[0] == KERNEL:
            alloc = ("Allocated by the compiler (profiles/%s): **%s VGPRs**, %s SGPRs, %s B scratch per lane, %s waves per SIMD, %s B LDS per "
                     "workgroup.  rocprofv3's `VGPR_Count` below is about half of the allocated VGPRs on gfx950 -- do not size occupancy from it."
                     % (ru, f[1], f[2], f[3], f[4], f[7]))
except Exception:
    pass
out += ["## dispatch", ""] + ([alloc, ""] if alloc else []) + ["```", json.dumps(meta), "```", "", "## PMC counters of `%s` (mean per launch)" % KERNEL, "",
        "| counter | mean | launches |", "|---|---|---|"]
for k in sorted(counters):
    out.append("| %s | %.6g | %d |" % (k, counters[k][0], counters[k][1]))
traffic = None
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    f, w = counters["FETCH_SIZE"][0] * 1024, counters["WRITE_SIZE"][0] * 1024
    traffic = 2 * f + w
    out += ["", "## HBM-side traffic per launch", "",
            "FETCH_SIZE and WRITE_SIZE are in KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports",
            "half of the bytes of 16-B/lane reads; every global read of this kernel is a `dwordx4`, so the read side",
            "is doubled: traffic = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (WRITE_SIZE is uncalibrated in the guide;",
            "it matches the kernel's known store volume here).", "",
            "* read  = 2 x %.4g B = %.4g B" % (f, 2 * f), "* write = %.4g B" % w, "* total = %.4g B per launch" % traffic]
if avg_ns and "SQ_INSTS_VALU" in counters:
    out += ["", "## derived", "", "* kernel average %.1f us (profiled run)" % (avg_ns / 1e3),
            "* VALU wave-instructions per launch %.4g -> %.4g per SIMD (1024 SIMDs)" % (
                counters["SQ_INSTS_VALU"][0], counters["SQ_INSTS_VALU"][0] / 1024)]
    if "SQ_BUSY_CYCLES" in counters:
        cyc = counters["SQ_BUSY_CYCLES"][0] / 32.0
        out += ["* SQ_BUSY_CYCLES / 32 shader engines = %.4g cycles per launch -> effective clock %.2f GHz" % (
            cyc, cyc / avg_ns), "* VALU instructions per SIMD per busy cycle = %.3f" % (
            counters["SQ_INSTS_VALU"][0] / 1024 / cyc)]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", "%s_%s_summary.md" % (tag, wl)), "w") as fh:
    fh.write("\n".join(out) + "\n")
tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
d = json.load(open(tj)) if os.path.exists(tj) else {}
if traffic:
    # the counters are stamped with the fingerprint of the profiled kernel's MACHINE CODE (gpurun_out/<tag>/code_hashes.json, written
    # on the GPU box by tools/profile_r.sh from the library that ran): they must describe the library in this tree
    hp, cp = os.path.join(src, "source_hash.txt"), os.path.join(src, "code_hashes.json")
    profiled = open(hp).read().strip() if os.path.exists(hp) else None
    code = json.load(open(cp)).get(KERNEL) if os.path.exists(cp) else srchash.kernel_hash(KERNEL)
    if code != srchash.kernel_hash(KERNEL):
        raise SystemExit("%s in this tree's library is not the code %s profiled (%s != %s): re-profile" % (KERNEL, tag, srchash.kernel_hash(KERNEL), code))
    d[wl] = {"bytes_per_launch": traffic, "source": "profiles/%s_%s_summary.md" % (tag, wl), "batch": int(meta.get("Grid_Size", 0)) and 1 << 20,
             "source_hash": profiled or srchash.tree_hash(), "code_hash": code}
    d[wl]["kernel"] = KERNEL
    if "SQ_INSTS_VALU" in counters:
        d[wl]["valu_insts_per_launch"] = counters["SQ_INSTS_VALU"][0]
    if "SQ_BUSY_CYCLES" in counters:    # bench.py: counter-derived VALU utilisation (instructions per SIMD per busy cycle)
        d[wl]["sq_busy_cycles"] = counters["SQ_BUSY_CYCLES"][0]
    try:  # the window width the counters were collected with (bench.py only quotes them for the same configuration)
        bl = [l for l in open(os.path.join(src, "bench_%s.json" % wl)).read().splitlines() if l.startswith("{")][-1]
        d[wl]["window_bits"] = json.loads(bl)["config"]["window_bits"]
    except Exception:
        pass
    json.dump(d, open(tj, "w"), indent=1)
bj = os.path.join(src, "bench_%s.json" % wl)
if os.path.exists(bj) and os.path.getsize(bj):
    open(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, wl)), "w").write(open(bj).read())
print("\n".join(out[-14:]))
