#!/bin/bash
# Round 3, step p: four workgroups per CU for the 128 x 32 scalar-addressed kernel again (tools/ab/libfiery_hip_tail4.so =
# -DFIERY_TAIL_FOUR_PER_CU=1), now that the third-stage epilogue requests its residual rows in two halves (16 -> 4 spilled
# registers, none in a loop): phase probe of the tail launch and the bench line, both ways, alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_p
mkdir -p $O
for rep in 1 2; do
  for lib in three four; do
    [ $lib = four ] && export FIERY_HIP_LIB=$R/tools/ab/libfiery_hip_tail4.so || unset FIERY_HIP_LIB
    echo "== $lib per CU (run $rep)" >> $O/tail_probe.txt
    timeout 200 python tools/runs/r3_tail_probe.py 2>&1 | grep -v amdgpu.ids >> $O/tail_probe.txt
    timeout 300 python bench.py --steps 20 --warmup 3 --no-from-images --no-cpu-baseline --no-bf16-mode > $O/bench_${lib}_$rep.json 2>> $O/bench.err
    echo "$lib $rep $(grep -h -o '"value": [0-9.]*' $O/bench_${lib}_$rep.json | head -1) $(grep -h -o '"frac": [0-9.]*' $O/bench_${lib}_$rep.json | head -1)"
  done
done
unset FIERY_HIP_LIB
cat $O/tail_probe.txt
