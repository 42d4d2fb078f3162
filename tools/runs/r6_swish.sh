#!/bin/bash
# the swish epilogue with the hardware exp / rcp sigmoid (HEAD) beside libm's (tools/ab/libfiery_hip_libmswish.so: -DFIERY_SWISH_FAST=0): forward() from images, one box, alternating; trunk parity tests
mkdir -p gpurun_out/swish; O=gpurun_out/swish
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "image or trunk or encoder or backbone or lift_head" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for r in 1 2; do for lib in base libmswish; do
  if [ $lib = base ]; then unset FIERY_HIP_LIB; else export FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$lib.so; fi
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --single-parity-draw --no-bf16-mode --no-secondary-configs 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); f=d['forward_from_images']; print('$lib', 'from images ms', f['ms_per_step'], 'kernels total', f.get('kernels_ms_total'), [(x['name'][-22:], x['ms']) for x in f['kernels'][:3]], 'headline', d['value'])"
done; done | tee $O/ab.txt
