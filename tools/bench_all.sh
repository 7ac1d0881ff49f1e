#!/bin/bash
# Developer tool: one compact line per workload (value, ms/step, parity) on a GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for WL in ${@:-fixed_base var_base poseidon5 verify}; do
  python3 bench.py --workload $WL --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-also --no-strong 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); print('%-11s %10.2f M/s  %9.3f ms/step  kernel %9.3f ms  parity %s' % ('$WL', d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['parity_sample_ok']))
"
done
