#!/bin/bash
# Round 4: the convolution's epilogue in parts (tuning build): staged + barrier / scale-shift arrival / row loop; then the bench
# line with the scale / shift requested before the staging
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_ag
mkdir -p $O
TRACE_PERSISTENT=0 FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_tuning.so timeout 300 python tools/runs/r4_conv_trace.py 2>&1 | grep -v amdgpu.ids | tee $O/conv_epilogue_parts.txt
for rep in 1 2; do
  for v in prerows today; do
    lib=$PWD/fiery_amd/libfiery_hip.so; [ $v = prerows ] && lib=$PWD/tools/ab/libfiery_hip_prerows.so
    FIERY_HIP_LIB=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_${v}_$rep.json 2>> $O/bench.err
    python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f)' % (d['value'], d['roofline']['frac'], d['roofline']['timed_mode']['frac']))
PY
  done
done 2>&1 | tee $O/summary.txt
