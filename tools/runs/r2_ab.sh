#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_base.so FIERY_BENCH_DUMP=$O/launches_base.json timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_base.json 2> $O/bench_base.err
FIERY_BENCH_DUMP=$O/launches_cur.json timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cur.json 2> $O/bench_cur.err
FIERY_HEADS_FUSED=0 FIERY_BENCH_DUMP=$O/launches_unfused.json timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
for f in base cur unfused; do cut -c1-330 $O/bench_$f.json; done
