#!/bin/bash
# Round 4: 128-cout layers as two 64-wide cout tiles (64 x 64 tiles, four workgroups per CU) against one 128-wide tile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_aj
mkdir -p $O
for rep in 1 2; do
  for v in wide narrow; do
    var=""; [ $v = narrow ] && var="FIERY_CONV_BN128_AS_64=1"
    env $var timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_${v}_$rep.json 2>> $O/bench.err
    python - <<PY
import json
d=json.loads([l for l in open('$O/bench_${v}_$rep.json').read().splitlines() if l.startswith('{')][-1])
print('$v rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f)  seg err %.2e' % (d['value'], d['roofline']['frac'], d['roofline']['timed_mode']['frac'], d['parity']['segmentation']['max_abs_err']))
PY
  done
done 2>&1 | tee $O/summary.txt
tail -3 $O/bench.err
