#!/bin/bash
# Round 4: the row epilogue instantiated per kind (BatchNorm + ReLU: ~15 vector instructions per row instead of ~35) against the
# run-time-dispatched loop, one box, alternating processes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_af
mkdir -p $O
for rep in 1 2 3; do
  for v in prerows today; do
    lib=$PWD/fiery_amd/libfiery_hip.so; [ $v = prerows ] && lib=$PWD/tools/ab/libfiery_hip_prerows.so
    FIERY_HIP_LIB=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_${v}_$rep.json 2>> $O/bench.err
    python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f)  parity seg %.2e' % (d['value'], d['roofline']['frac'], d['roofline']['timed_mode']['frac'] if 'timed_mode' in d['roofline'] else d.get('roofline_timed',{}).get('frac',0), d['parity']['segmentation']['max_abs_err']))
PY
  done
done 2>&1 | tee $O/summary.txt
FIERY_HIP_LIB=$PWD/fiery_amd/libfiery_hip.so timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep -v amdgpu.ids > $O/microbench_today.txt
FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_prerows.so timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep -v amdgpu.ids > $O/microbench_prerows.txt
paste -d'\n' $O/microbench_prerows.txt $O/microbench_today.txt | head -60
