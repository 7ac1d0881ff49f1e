#!/usr/bin/env python3
"""Developer tool: static VALU instruction mix of the shipped kernels -> profiles/isa_mix.json (read by bench.py's
`valu` block) and a readable table (profiles/<tag>_isa_mix.txt).

For every kernel translation unit `hipcc -S --cuda-device-only` produces the gfx950 assembly; instructions are counted
per kernel function (callees that are not inlined -- fr_inv_gcd -- are listed on their own) and put into the issue
classes that tools/ubench/ubench.hip measured on the MI355X (profiles/r01_ubench_valu_rates.txt, 8 waves per SIMD,
cycles per wave-instruction per SIMD):
    mad64   v_mad_u64_u32 / v_mad_i64_i32 (dependent accumulator chains, as the column multiplier issues them)  4.54
    quarter every other measured multi-pass class: v_mul_lo/hi_u32, v_add_co/addc/subb chains, v_add3_u32,
            v_lshl_add_u64, 64-bit shifts, v_alignbit_b32, f64 arithmetic and conversions, v_mad_u32_u24, ...         4.20
    plain   single-pass 32-bit VOP1/VOP2 (v_mov_b32, v_add_u32 were measured: 2.32 / 2.47; v_and/or/xor/lshl/lshr/
            v_sub_u32/v_cndmask_b32/v_bfe are ASSUMED to be of the same class)                                      2.40
The weighted mean is the kernel's average issue cost per VALU wave-instruction; 1024 SIMDs x 2.4 GHz / that mean is the
kernel's VALU-issue ceiling in wave-instructions per second.  STATIC counts: loop bodies are weighted like straight-line
code, which is acceptable here because every hot loop body is the same field arithmetic as the code around it (the mix
varies between 80 and 84 % mad64 from kernel to kernel); the dynamic count per launch (SQ_INSTS_VALU) comes from the
rocprofv3 PMC pass.

usage: tools/isa_histogram.py [tag]      (tag defaults to r02)
"""
import collections
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "babyjubjub-rs_amd", "csrc")
UNITS = ["k_fixed", "k_var", "k_hash_codec", "k_verify", "k_sign", "k_small"]
CYC = {"mad64": 4.54, "quarter": 4.20, "plain": 2.40}
PLAIN = re.compile(r"^v_(mov_b32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|lshlrev_b32|lshrrev_b32|ashrrev_i32|"
                   r"cndmask_b32|bfe_u32|bfi_b32|and_or_b32|or3_b32|lshl_or_b32|lshl_add_u32|add_lshl_u32|min_u32|max_u32|"
                   r"readfirstlane_b32|readlane_b32|writelane_b32|accvgpr_read_b32|accvgpr_write_b32|mov_b64|perm_b32|xad_u32|"
                   r"cmp_\w+_[ui]32|cmpx_\w+_[ui]32|nop)(_e32|_e64|_sdwa|_dpp)?$")


def asm_of(unit):
    out = "/tmp/isa_%s.s" % unit
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                    "-Wno-unused-value", "-S", "--cuda-device-only", "-o", out, unit + ".hip"], cwd=CSRC, check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def classify(op):
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "mad64"
    if PLAIN.match(op):
        return "plain"
    return "quarter"


def histogram(lines):
    res, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            res[cur] = {"valu": collections.Counter(), "other": collections.Counter()}
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
            continue
        if cur is None:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", l)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            res[cur]["valu"][op] += 1
        elif op.startswith(("s_", "ds_", "global_", "scratch_", "buffer_", "flat_")):
            res[cur]["other"][op.split("_")[0]] += 1
    return res


def demangled(name):
    m = re.match(r"_Z(\d+)", name)
    if m:
        k = int(m.group(1))
        return name[2 + len(m.group(1)):2 + len(m.group(1)) + k]
    m = re.match(r"_ZN3bjjL?(\d+)", name)
    if m:
        k = int(m.group(1))
        s = name.index(m.group(1)) + len(m.group(1))
        return "bjj::" + name[s:s + k]
    return name


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    with ThreadPoolExecutor(max_workers=5) as ex:
        asms = list(ex.map(asm_of, UNITS))
    mix, table = {}, ["# static VALU instruction mix per kernel (tools/isa_histogram.py; classes and cycles: see the tool's docstring)", "",
                      "%-34s %9s %7s %7s %7s %9s %10s %8s %8s" % ("kernel", "VALU", "mad64", "quarter", "plain", "avg cyc", "peak Gi/s", "scratch", "ds+glob")]
    for unit, lines in zip(UNITS, asms):
        for fn, h in histogram(lines).items():
            tot = sum(h["valu"].values())
            if tot < 50:
                continue
            cls = collections.Counter()
            for op, c in h["valu"].items():
                cls[classify(op)] += c
            avg = sum(CYC[k] * v for k, v in cls.items()) / tot
            name = demangled(fn)
            mix[name] = {"unit": unit, "valu_insts_static": tot, "mad64": cls["mad64"], "quarter": cls["quarter"], "plain": cls["plain"],
                         "quarter_rate_share": (cls["mad64"] + cls["quarter"]) / tot, "avg_issue_cycles_per_valu_inst": avg,
                         "peak_g_wave_insts_per_s": 1024 * 2.4 / avg, "scratch_insts_static": h["other"]["scratch"],
                         "top_quarter_ops": dict(collections.Counter({o: c for o, c in h["valu"].items() if classify(o) == "quarter"}).most_common(6)),
                         "top_plain_ops": dict(collections.Counter({o: c for o, c in h["valu"].items() if classify(o) == "plain"}).most_common(6))}
            table.append("%-34s %9d %6.1f%% %6.1f%% %6.1f%% %9.3f %10.1f %8d %8d" % (
                name[:34], tot, 100 * cls["mad64"] / tot, 100 * cls["quarter"] / tot, 100 * cls["plain"] / tot, avg, 1024 * 2.4 / avg,
                h["other"]["scratch"], h["other"]["ds"] + h["other"]["global"]))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    sys.path.insert(0, os.path.join(ROOT, "babyjubjub-rs_amd"))
    import srchash
    mix["_source_hash"] = srchash.tree_hash()
    kh = srchash.kernel_hashes()                # bench.py quotes a kernel's mix only while the library holds the same machine code for it
    for name, m in mix.items():
        if isinstance(m, dict) and name in kh:
            m["code_hash"] = kh[name]
    json.dump(mix, open(os.path.join(ROOT, "profiles", "isa_mix.json"), "w"), indent=1, sort_keys=True)
    open(os.path.join(ROOT, "profiles", "%s_isa_mix.txt" % tag), "w").write("\n".join(table) + "\n")
    print("\n".join(table))


if __name__ == "__main__":
    main()
