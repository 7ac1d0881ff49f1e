#!/bin/bash
# Developer tool: HBM-side traffic (FETCH_SIZE / WRITE_SIZE PMC passes) of one workload for the library named by
# BJJ_LIB_PATH (default: the in-tree one).  usage: tools/profile_traffic_only.sh <tag> <workload>
TAG=$1; WL=$2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --no-cpu-baseline --no-also --no-strong"
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$WL -o pmc -- python3 $ARGS > $OUT/pmc_fetch_$WL.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$WL -o pmc -- python3 $ARGS > $OUT/pmc_write_$WL.log 2>&1
python3 - "$OUT" "$WL" <<'PY'
import csv, sys
out, wl = sys.argv[1], sys.argv[2]
K = {"verify": "bjj_k_eddsa_verify(", "var_base": "bjj_k_mul_var_base("}[wl]
tot = {}
for c in ("fetch", "write"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open("%s/pmc_%s_%s/pmc_counter_collection.csv" % (out, c, wl))) if r["Kernel_Name"].startswith(K)]
    tot[c] = sum(v) / len(v) * 1024
print("%s: read 2 x %.4g B, write %.4g B -> %.4g B per launch (%d launches)" % (wl, tot["fetch"], tot["write"], 2 * tot["fetch"] + tot["write"], len(v)))
PY
rm -rf $OUT
