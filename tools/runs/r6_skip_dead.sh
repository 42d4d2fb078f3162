#!/bin/bash
# pooling with dead slices stepped over (HEAD) beside -DFIERY_POOL_SKIP_DEAD_SLICES=0 (tools/ab/libfiery_hip_noskip.so): pon and baseline, one box, alternating
mkdir -p gpurun_out/skip; O=gpurun_out/skip
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "pool or lift_splat or hot_path" > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
for r in 1 2; do for lib in base noskip; do
  if [ $lib = base ]; then unset FIERY_HIP_LIB; else export FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$lib.so; fi
  for cfg in "--config literature/pon_setting.yml --precision bf16" ""; do
  timeout 600 python bench.py $cfg --steps 20 --warmup 3 --no-cpu-baseline --single-parity-draw --no-from-images --no-bf16-mode --no-secondary-configs 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); p=d['roofline_pooling']; print('$lib', d['config'].get('workload','')[:28], d['value'], p['frac'], p['op_us_samples'])"
  done; done; done | tee $O/ab.txt
