#!/bin/bash
# the 35-channel temporal layers padded to whole 32-channel stages (so that the split tile kernels take them) beside the 8-aligned layout
mkdir -p gpurun_out/pad32; O=gpurun_out/pad32
A="--steps 20 --warmup 3 --no-cpu-baseline --single-parity-draw --no-secondary-configs --no-from-images --no-bf16-mode"
for r in 1 2; do for v in 1 0; do
  FIERY_TEMPORAL_PAD32=$v FIERY_BENCH_DUMP=$O/l_pad$v.json timeout 900 python bench.py $A 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('pad32=$v', d['value'], d['ms_per_step'], 'conv ms', r['kernel_ms_per_step'])"
done; done | tee $O/ab.txt
python tools/launches_table.py $O/l_pad1.json | head -24 > $O/t_pad1.txt
