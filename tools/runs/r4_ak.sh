#!/bin/bash
# Round 4: bench line with every instrumented step enqueued behind an unrecorded one (the pooling bracket no longer holds host time)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_ak
mkdir -p $O
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_$rep.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads([l for l in open('$O/bench_$rep.json').read().splitlines() if l.startswith('{')][-1])
rp=d['roofline_pooling']
print('rep $rep: %.1f samples/s  conv frac %.4f  pooling %.4f  op %.1f us  samples %s  host_ms %s' % (d['value'], d['roofline']['frac'], rp['frac'], rp['op_us_per_step'], rp['op_us_samples'], d.get('host_issue_ms', d.get('host_ms'))))
PY
done 2>&1 | tee $O/summary.txt
