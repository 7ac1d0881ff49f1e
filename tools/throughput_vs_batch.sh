#!/bin/bash
# Developer tool: throughput against the batch size (one stream and the default two-stream protocol), one MI355X.
cd ${GRAFT_REPO_ROOT:-.}
for WL in fixed_base verify var_base; do
  for B in 4096 65536 262144 1048576 4194304 16777216; do
    K=$(( B >= 4194304 ? 5 : 30 ))
    python3 bench.py --workload $WL --batch $B --batches 2 --steps $K --warmup 3 --warmup-seconds 0.3 --no-cpu-baseline --no-also --no-strong 2>&1 >/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if not line.startswith('bench_detail: {'): continue      # the full record (stderr); stdout carries the compact line
    d = json.loads(line[len('bench_detail: '):]); one = d.get('single_stream')
    print('%-10s batch %9d  %9.3f ms/step %10.2f M/s (%d stream%s)%s  parity %s' % ('$WL', $B, d['ms_per_step'], d['value'] / 1e6, d.get('streams', 1), 's' if d.get('streams', 1) > 1 else '',
          ('   [1 stream: %9.3f ms %9.2f M/s]' % (one['kernel_ms_avg'], one['value_this_rank'] / 1e6)) if one else '', d['parity_sample_ok']))
"
  done
done
