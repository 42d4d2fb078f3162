#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2c3
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16 or conv_igemm" 2>&1 | tail -40 | tee $O/gputests_bf16.txt
echo "== conv fp32"; timeout 300 python tools/microbench.py conv --reps 10 2>&1 | grep "^conv" | head -7 | tee $O/conv_f32.txt
echo "== conv bf16"; CONV_PRECISION=bf16 timeout 300 python tools/microbench.py conv --reps 10 2>&1 | grep "^conv" | head -7 | tee $O/conv_bf16.txt
timeout 600 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images > $O/bench_bf16.log 2>&1; tail -1 $O/bench_bf16.log | cut -c1-3000
timeout 600 python bench.py --steps 10 --warmup 3 --precision bf16 --config literature/pon_setting.yml --no-from-images > $O/bench_pon_bf16.log 2>&1; tail -1 $O/bench_pon_bf16.log | cut -c1-3000
FIERY_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --layout frames --steps 10 --warmup 2 --no-cpu-baseline --no-from-images > $O/bench_frames.log 2>&1; tail -4 $O/bench_frames.log | cut -c1-1500
