#!/bin/bash
# Round 3, step m: the fp32 halo loop held to 128 registers (four workgroups per CU: 1,024 slots) against three per CU
# (tools/ab/libfiery_hip_halo_f32_w3.so = -DFIERY_HALO_F32_WAVES=3) and against the scalar-addressed loop.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_m
mkdir -p $O
for mode in off halo4 halo3; do
  unset FIERY_HIP_LIB; export FIERY_CONV_HALO_F32=1
  [ $mode = off ] && export FIERY_CONV_HALO_F32=0
  [ $mode = halo3 ] && export FIERY_HIP_LIB=$R/tools/ab/libfiery_hip_halo_f32_w3.so
  echo "== $mode" >> $O/microbench_f32.txt
  timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep "k3 s1" >> $O/microbench_f32.txt
  timeout 300 python bench.py --steps 20 --warmup 3 --no-from-images --no-cpu-baseline > $O/bench_fp32_$mode.json 2>> $O/bench.err
  echo "$mode $(grep -h -o '"value": [0-9.]*' $O/bench_fp32_$mode.json | head -1) $(grep -h -o '"frac": [0-9.]*' $O/bench_fp32_$mode.json | head -1)"
done
cat $O/microbench_f32.txt
