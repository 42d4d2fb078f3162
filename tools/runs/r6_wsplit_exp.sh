#!/bin/bash
# timing experiments on the split Winograd kernel (-DW_EXP=n, wrong results on purpose): where its time is
mkdir -p gpurun_out/wsplit; O=gpurun_out/wsplit
for r in 1 2; do for lib in ${LIBS:-base wsx1 wsx2 wsx3 wsx4}; do
  if [ $lib = base ]; then unset FIERY_HIP_LIB; else export FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$lib.so; fi
  echo -n "$lib  "; FORM=wsplit timeout 200 python tools/runs/r5_wino_times.py 2>&1 | tail -1; done; done | tee $O/exp_times.txt
