#!/bin/bash
# Round 4: tile height chosen per launch in isolation (default) against 128 / 64 forced everywhere, in the timed mode (three sample
# chains share the GPU, so a launch never runs alone)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_ai
mkdir -p $O
for rep in 1 2; do
  for v in auto 128 64; do
    var=""; [ $v != auto ] && var="FIERY_CONV_TILE_M=$v"
    env $var timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_${v}_$rep.json 2>> $O/bench.err
    python - <<PY
import json
d=json.loads([l for l in open('$O/bench_${v}_$rep.json').read().splitlines() if l.startswith('{')][-1])
print('tile $v rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f)' % (d['value'], d['roofline']['frac'], d['roofline']['timed_mode']['frac']))
PY
  done
done 2>&1 | tee $O/summary.txt
