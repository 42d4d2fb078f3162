#!/bin/bash
# Round 4: raised wave priority for the conv epilogue (variant library) against the default, alternating on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_l
mkdir -p $O
for rep in 1 2 3; do
  for v in default prio3; do
    lib=$PWD/fiery_amd/libfiery_hip.so; [ $v = prio3 ] && lib=$PWD/tools/ab/libfiery_hip_prio3.so
    echo "== $v (rep $rep)"
    FIERY_HIP_LIB=$lib timeout 600 python bench.py --steps 20 --warmup 3 --no-from-images --no-bf16-mode --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['timed_mode']['frac'])"
  done
done | tee $O/bench_ab.txt
for v in default prio3; do
    lib=$PWD/fiery_amd/libfiery_hip.so; [ $v = prio3 ] && lib=$PWD/tools/ab/libfiery_hip_prio3.so
    echo "== $v"
    FIERY_HIP_LIB=$lib timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep "^conv"
done | tee $O/conv_ab.txt
