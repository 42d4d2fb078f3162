#!/bin/bash
# Round 5 evidence run at HEAD: GPU parity suite (ledger), smoke(), the default bench line (fp32 headline + bf16 mode + secondary
# configs pon / lyft-7), eager one-stream and frames-layout lines at one rank, conv / pooling microbenchmarks, training-step timing,
# kernel traces in both launch modes, PMC traffic passes, MFMA counter pass, the fp32 matrix pipe's ceiling on this box.
# Files are copied to profiles/r5_* afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_final
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
cp $R/gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
hipcc --offload-arch=gfx950 -O3 -w tools/probe/mfma_ceiling.hip -o /tmp/mfma_ceiling && timeout 120 /tmp/mfma_ceiling > $O/mfma_ceiling.txt 2>&1
FIERY_BENCH_DUMP=$O/launches.json timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-graph --no-sample-streams --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_batch_layout_eager_one_stream.json 2>> $O/bench.err
for ex in all_to_all all_gather; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 FIERY_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --layout frames --exchange $ex --no-cpu-baseline --no-from-images > $O/bench_frames_layout_${ex}_rccl_1rank.json 2>> $O/bench.err
done
grep -h -o '"value": [0-9.]*' $O/bench_batch_layout_eager_one_stream.json $O/bench_frames_layout_*_1rank.json
FIERY_BENCH_DUMP=$O/launches_bf16.json timeout 600 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images --no-cpu-baseline > $O/bench_baseline_bf16.json 2>> $O/bench.err
python tools/launches_table.py $O/launches.json > $O/launches_table.txt 2>&1; head -30 $O/launches_table.txt
python tools/launches_table.py $O/launches_bf16.json > $O/launches_table_bf16.txt 2>&1
timeout 200 tools/probe/_bin/pool_ceiling 9 0.9409 > $O/pool_ceiling.txt 2>&1; tail -2 $O/pool_ceiling.txt
timeout 300 python tools/runs/r5_wino_check.py > $O/winograd_check.txt 2>&1; tail -3 $O/winograd_check.txt
timeout 300 python tools/runs/r5_sk_check.py > $O/stream_k_check.txt 2>&1; tail -2 $O/stream_k_check.txt
{
  echo "# round 5 - tools/microbench.py conv, fp32 form"
  timeout 300 python tools/microbench.py conv --reps 20 2>&1 | grep -v amdgpu.ids
  echo "# voxel pooling op"
  timeout 300 python tools/microbench.py pool --reps 20 2>&1 | grep -v amdgpu.ids
} > $O/microbench.txt
{
  echo "# round 5 - one training step of the path (forward + backward + SGD step from the lifted features), baseline.yml, B = 2, tools/time_train_step.py"
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step
  echo "# the same graph with PyTorch-ROCm operators for convolution / BatchNorm / upsampling (MIOpen, ATen)"
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --torch-conv 2>&1 | grep time_train_step
} > $O/train_step.txt
grep time_train $O/train_step.txt | cut -c1-200
cd /tmp
for mode in one_stream sample_streams; do
  extra=""; [ $mode = one_stream ] && extra="--no-sample-streams"
  rm -rf /tmp/kt_$mode
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$mode -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs $extra > $O/kt_$mode.log 2>&1
  db=$(find /tmp/kt_$mode -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_$mode.csv "round 5 ($mode): rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs $extra"
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs --no-graph --no-sample-streams > $O/pmc_$ctr.log 2>&1
  python $R/tools/pmc_dump.py "/tmp/pmc_$ctr/**/*.db" > $O/pmc_$ctr.txt 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_traffic.json
# the same two passes in the bf16 mode (round 4's lines had no counter traffic for it)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc16_$ctr
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc16_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --precision bf16 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs --no-graph --no-sample-streams > $O/pmc16_$ctr.log 2>&1
  python $R/tools/pmc_dump.py "/tmp/pmc16_$ctr/**/*.db" > $O/pmc16_$ctr.txt 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc16_FETCH_SIZE.txt $O/pmc16_WRITE_SIZE.txt $O/pmc_traffic_bf16.json
rm -rf /tmp/pmc_mfma
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace -d /tmp/pmc_mfma -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs --no-graph --no-sample-streams > $O/pmc_mfma.log 2>&1
python $R/tools/pmc_dump.py "/tmp/pmc_mfma/**/*.db" > $O/pmc_mfma.txt 2>&1
python $R/tools/pmc_mfma.py $O/pmc_mfma.txt > $O/mfma_util.txt 2>&1; tail -8 $O/mfma_util.txt
head -10 $O/kernel_stats_one_stream.csv
