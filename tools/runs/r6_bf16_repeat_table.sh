#!/bin/bash
# per-launch times: fp32 forms as served / bf16 mode / bf16 mode with every MFMA issued six times (the split form's bound)
O=gpurun_out/rep6; mkdir -p $O
A="--steps 20 --warmup 3 --no-cpu-baseline --single-parity-draw --no-secondary-configs --no-from-images --no-bf16-mode"
FIERY_BENCH_DUMP=$O/l_f32.json timeout 600 python bench.py $A > $O/b_f32.json 2>/dev/null
FIERY_BENCH_DUMP=$O/l_bf16.json timeout 600 python bench.py $A --precision bf16 > $O/b_bf16.json 2>/dev/null
FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_rep6.so FIERY_BENCH_DUMP=$O/l_rep6.json timeout 600 python bench.py $A --precision bf16 > $O/b_rep6.json 2>/dev/null
for k in f32 bf16 rep6; do python tools/launches_table.py $O/l_$k.json > $O/t_$k.txt 2>&1; done
