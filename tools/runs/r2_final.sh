#!/bin/bash
# Round 2, final evidence at HEAD: the whole GPU parity suite (with the achieved-error ledger), smoke(), the bench line (fp32 and
# bf16 operands), the training-step timings (HIP operators / MIOpen convolutions / the reference's operator sequence) and the
# kernel trace of the default bench command in both launch modes.  Files are copied to profiles/r2_* afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_final
mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
cp $R/gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
FIERY_BENCH_DUMP=$O/launches.json timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cut -c1-300 $O/bench.json
timeout 600 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images > $O/bench_bf16.json 2>> $O/bench.err
cut -c1-200 $O/bench_bf16.json
{
  echo "# round 2 - one training step of the path (forward + backward + SGD step from the lifted features), baseline.yml, B = 2, tools/time_train_step.py"
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --profile 2>&1 | grep -v "^\[W\|amdgpu.ids\|^$\|_warn_once\|Only events"
  echo "# the same graph with PyTorch-ROCm operators for convolution / BatchNorm / upsampling (MIOpen, ATen)"
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --torch-conv 2>&1 | grep time_train_step
  echo "# the reference's own operator sequence (adds avg_pool3d + interpolate for the pyramid pooling)"
  timeout 600 python tools/time_train_step.py --batch 2 --steps 5 --reference-ops 2>&1 | grep time_train_step
} > $O/train_step.txt
grep time_train_step $O/train_step.txt
cd /tmp
for mode in ${FIERY_FINAL_TRACES:-one_stream sample_streams}; do
  extra=""; [ $mode = one_stream ] && extra="--no-sample-streams"
  rm -rf /tmp/kt_$mode
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$mode -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-from-images $extra > $O/kt_$mode.log 2>&1
  db=$(find /tmp/kt_$mode -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_$mode.csv "round 2 ($mode): rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-from-images $extra"
done
head -6 $O/kernel_stats_one_stream.csv
