#!/bin/bash
# Developer tool: one compact line per workload (value, ms/step, parity) on a GPU box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for WL in ${@:-fixed_base var_base poseidon5 verify}; do
  python3 bench.py --workload $WL --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-also --no-strong ${BENCH_ARGS} 2>&1 >/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('bench_detail: {'): continue      # the full record (stderr); stdout carries the compact line
    d=json.loads(line[len('bench_detail: '):])
    one=d.get('single_stream')
    extra=('   [1 stream: %9.3f ms, %8.2f M/s]' % (one['kernel_ms_avg'], one['value_this_rank']/1e6)) if one else ''
    print('%-11s %10.2f M/s  %9.3f ms/step  device %9.3f ms/launch (%d stream%s)  parity %s%s' % ('$WL', d['value']/1e6, d['ms_per_step'], d.get('device_ms_per_launch', d['roofline']['kernel_ms_avg']), d.get('streams',1), 's' if d.get('streams',1)>1 else '', d['parity_sample_ok'], extra))
"
done
