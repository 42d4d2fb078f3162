#!/bin/bash
# Round 4: operand requests re-issued right after the store that frees their register (a full stage to land) against all stores
# first / all requests after (half a stage), one box, alternating processes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_ah
mkdir -p $O
for rep in 1 2 3; do
  for v in lateloads today; do
    lib=$PWD/fiery_amd/libfiery_hip.so; [ $v = lateloads ] && lib=$PWD/tools/ab/libfiery_hip_lateloads.so
    FIERY_HIP_LIB=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_${v}_$rep.json 2>> $O/bench.err
    python - <<PY
import json
d=json.loads(open('$O/bench_${v}_$rep.json').read().strip().splitlines()[-1])
print('$v rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f)  parity seg %.2e' % (d['value'], d['roofline']['frac'], d['roofline']['timed_mode']['frac'] if 'timed_mode' in d['roofline'] else d.get('roofline_timed',{}).get('frac',0), 0))
PY
  done
done 2>&1 | tee $O/summary.txt
FIERY_HIP_LIB=$PWD/fiery_amd/libfiery_hip.so timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep -v amdgpu.ids > $O/microbench_today.txt
FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_lateloads.so timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep -v amdgpu.ids > $O/microbench_lateloads.txt
paste -d'\n' $O/microbench_lateloads.txt $O/microbench_today.txt | head -60
