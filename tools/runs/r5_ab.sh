#!/bin/bash
# Round 5: same-box A/B of library builds.  LIBS="a.so b.so" (paths from the repo root; '-' = in-tree), REPS=2, EXTRA="bench args"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r5_ab}
O=gpurun_out/$TAG
mkdir -p $O
LIBS=${LIBS:-"tools/ab/libfiery_hip_r4.so -"}
REPS=${REPS:-2}
# throw-away run: the clock settles, the image pages in
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > /dev/null 2>&1
for rep in $(seq 1 $REPS); do
  i=0
  for lib in $LIBS; do
    i=$((i+1))
    var=""; [ "$lib" != "-" ] && var="FIERY_HIP_LIB=$PWD/$lib"
    env $var FIERY_BENCH_DUMP=$O/dump_${i}_$rep.json timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs $EXTRA 2>$O/err_${i}_$rep.txt > $O/b_${i}_$rep.json
    python - <<PY
import json
try:
    d=json.loads(open('$O/b_${i}_$rep.json').read().strip().splitlines()[-1])
    r=d['roofline']; rp=d['roofline_pooling']
    print('%-40s rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f) kernel ms %.3f  pool %.1f us (%.4f)' % ('$lib', d['value'], r['frac'], r['timed_mode']['frac'], r['kernel_ms_per_step'], rp['op_us_per_step'], rp['frac']))
except Exception as e:
    print('$lib rep $rep FAILED', e, open('$O/err_${i}_$rep.txt').read()[-1500:])
PY
  done
done 2>&1 | tee $O/summary.txt
python tools/launches_table.py $O/dump_1_$REPS.json $O/dump_2_$REPS.json > $O/launches_ab.txt 2>&1
head -30 $O/launches_ab.txt
