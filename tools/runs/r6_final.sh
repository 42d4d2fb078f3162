#!/bin/bash
# Round 6 evidence run at HEAD: GPU parity suite (ledger), smoke(), the default bench line (fp32 headline + bf16 mode + pon / lyft-7),
# eager one-stream and frames-layout lines at one rank, pooling A/B + phases, ceilings, kernel stats of exactly N steps with the
# convolution forms frozen (stats / N = per-step kernel times), PMC traffic passes for all four workloads, MFMA counter pass.
# Files are copied to profiles/r6_* afterwards (index profiles/README_r6.md).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r6_final
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
cp $R/gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
hipcc --offload-arch=gfx950 -O2 -w tools/probe/split_bf16_probe.hip -o /tmp/split_probe && timeout 120 /tmp/split_probe > $O/split_bf16_probe.txt 2>&1; tail -2 $O/split_bf16_probe.txt
{ for r in 1 2; do for f in wino wsplit; do echo -n "$f  "; FORM=$f timeout 200 python tools/runs/r5_wino_times.py 2>&1 | tail -1; done; done; } > $O/winograd_split_times.txt; tail -2 $O/winograd_split_times.txt | cut -c1-200
timeout 300 python tools/runs/r6_split_times.py 2>&1 | grep -v amdgpu.ids > $O/split_tile_times.txt; timeout 300 python tools/runs/r6_wgrad_times.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_times.txt
hipcc --offload-arch=gfx950 -O3 -w tools/probe/mfma_ceiling.hip -o /tmp/mfma_ceiling && timeout 120 /tmp/mfma_ceiling > $O/mfma_ceiling.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w tools/probe/pool_ceiling.hip -o /tmp/pool_ceiling && timeout 200 /tmp/pool_ceiling 9 0.9409 > $O/pool_ceiling.txt 2>&1; tail -2 $O/pool_ceiling.txt
FIERY_BENCH_DUMP=$O/launches.json timeout 1200 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-graph --no-sample-streams --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/bench_batch_layout_eager_one_stream.json 2>> $O/bench.err
for ex in all_to_all all_gather; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 FIERY_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --layout frames --exchange $ex --no-cpu-baseline --no-from-images > $O/bench_frames_layout_${ex}_rccl_1rank.json 2>> $O/bench.err
done
grep -h -o '"value": [0-9.]*' $O/bench_batch_layout_eager_one_stream.json $O/bench_frames_layout_*_1rank.json
FIERY_BENCH_DUMP=$O/launches_bf16.json timeout 600 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images --no-cpu-baseline > $O/bench_baseline_bf16.json 2>> $O/bench.err
python tools/launches_table.py $O/launches.json > $O/launches_table.txt 2>&1; head -30 $O/launches_table.txt
python tools/launches_table.py $O/launches_bf16.json > $O/launches_table_bf16.txt 2>&1
LIBS="-" VARIANTS="NO_RANKS=1 - NO_RANKS=1,FIERY_POOL_PREPASS_LEAN=0" TAG=r6_final/pool bash tools/runs/r6_pool_libs.sh > /dev/null 2>&1; cat $O/pool/summary.txt
VARIANT="NO_RANKS=1" TAG=r6_final/pool_trace bash tools/runs/r6_pool_trace.sh > /dev/null 2>&1; cat $O/pool_trace/phases.txt
timeout 300 python tools/runs/r5_wino_check.py > $O/winograd_check.txt 2>&1; tail -3 $O/winograd_check.txt
timeout 300 python tools/runs/r5_sk_check.py > $O/stream_k_check.txt 2>&1; tail -2 $O/stream_k_check.txt
{
  echo "# round 6 - one training step of the path (forward + backward + SGD step from the lifted features), baseline.yml, B = 2, tools/time_train_step.py"
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 2>&1 | grep time_train_step
} > $O/train_step.txt
grep time_train $O/train_step.txt | cut -c1-200
cd /tmp
# exactly N steps per process, convolution forms frozen from a table a previous process measured: stats / N = per-step kernel times
timeout 300 python $R/tools/runs/r6_profile_steps.py tune /tmp/forms.json > $O/forms_tune.txt 2>&1; tail -1 $O/forms_tune.txt
cp /tmp/forms.json $O/conv_forms.json
for mode in one_stream sample_streams sample_streams_graph; do
  extra=""; [ $mode = sample_streams ] && extra="--sample-streams"; [ $mode = sample_streams_graph ] && extra="--sample-streams --graph"
  rm -rf /tmp/kt_$mode
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$mode -o kt -- python $R/tools/runs/r6_profile_steps.py run /tmp/forms.json --steps 20 $extra > $O/kt_$mode.log 2>&1
  db=$(find /tmp/kt_$mode -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_$mode.csv "round 6 ($mode): rocprofv3 --kernel-trace --stats -- python tools/runs/r6_profile_steps.py run forms.json --steps 20 $extra  (EXACTLY 20 steps, frozen forms: total_us / 20 = per step)"
done
pmc_pair() {   # $1 tag, rest: bench args
  tag=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${tag}_$ctr
    timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_${tag}_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs --no-graph --no-sample-streams "$@" > $O/pmc_${tag}_$ctr.log 2>&1
    python $R/tools/pmc_dump.py "/tmp/pmc_${tag}_$ctr/**/*.db" > $O/pmc_${tag}_$ctr.txt 2>&1
  done
  python $R/tools/pmc_traffic.py $O/pmc_${tag}_FETCH_SIZE.txt $O/pmc_${tag}_WRITE_SIZE.txt $O/pmc_traffic_$tag.json
}
pmc_pair f32
pmc_pair bf16 --precision bf16
pmc_pair pon_bf16 --config literature/pon_setting.yml --precision bf16
pmc_pair lyft7_bf16 --config lyft/baseline.yml --cams 7 --precision bf16
rm -rf /tmp/pmc_mfma
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace -d /tmp/pmc_mfma -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs --no-graph --no-sample-streams > $O/pmc_mfma.log 2>&1
python $R/tools/pmc_dump.py "/tmp/pmc_mfma/**/*.db" > $O/pmc_mfma.txt 2>&1
python $R/tools/pmc_mfma.py $O/pmc_mfma.txt > $O/mfma_util.txt 2>&1; tail -8 $O/mfma_util.txt
head -12 $O/kernel_stats_one_stream.csv
