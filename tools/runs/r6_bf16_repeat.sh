#!/bin/bash
# Bound for a split-operand form (fp32 values as three bf16 terms, six products): the bf16 mode's step with every bf16 MFMA
# issued six times (tools/ab/libfiery_hip_rep6.so: -DFIERY_BF16_REPEAT=6, wrong results) beside the bf16 mode as built.
mkdir -p gpurun_out
for lib in base rep6 base rep6; do
  if [ $lib = base ]; then unset FIERY_HIP_LIB; else export FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$lib.so; fi
  timeout 600 python bench.py --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline --single-parity-draw --no-secondary-configs --no-from-images --no-bf16-mode 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d.get('kernel_ms_per_step'), d['roofline'].get('kernel'), d['roofline'].get('frac'))"
done 2>&1 | tee gpurun_out/bf16_repeat.txt
