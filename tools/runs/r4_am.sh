#!/bin/bash
# Round 4: the first GRU block (bias rows per image and border class) on the dense, per-kind row epilogue (today) against the
# pointer-arithmetic epilogue it took before (prebias), one box, alternating processes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_am
mkdir -p $O
for rep in 1 2; do
  for v in prebias today; do
    lib=$PWD/fiery_amd/libfiery_hip.so; [ $v = prebias ] && lib=$PWD/tools/ab/libfiery_hip_prebias.so
    FIERY_HIP_LIB=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_${v}_$rep.json 2>> $O/bench.err
    python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f)' % (d['value'], d['roofline']['frac'], d['roofline']['timed_mode']['frac']))
PY
  done
done 2>&1 | tee $O/summary.txt
