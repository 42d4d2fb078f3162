#!/bin/bash
# Round 2 evidence run: GPU parity suite (with the achieved-error ledger), the bench line, kernel traces in both launch modes,
# PMC traffic passes and the MFMA counter pass; files are copied to profiles/r2_* afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_final
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
cp $R/gpurun_out/parity_errors.json $O/parity_errors.json 2>/dev/null
FIERY_BENCH_DUMP=$O/launches.json timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
FIERY_BENCH_DUMP=$O/launches_bf16.json timeout 600 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images > $O/bench_bf16.json 2>> $O/bench.err
timeout 600 python bench.py --steps 10 --warmup 3 --precision bf16 --config lyft/baseline.yml --cams 7 --no-from-images > $O/bench_lyft7_bf16.json 2>> $O/bench.err
cut -c1-200 $O/bench_lyft7_bf16.json
cd /tmp
for mode in one_stream sample_streams; do
  extra=""; [ $mode = one_stream ] && extra="--no-sample-streams"
  rm -rf /tmp/kt_$mode
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$mode -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-from-images $extra > $O/kt_$mode.log 2>&1
  db=$(find /tmp/kt_$mode -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_$mode.csv "round 2 ($mode): rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-from-images $extra"
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-images --no-graph --no-sample-streams > $O/pmc_$ctr.log 2>&1
  python $R/tools/pmc_dump.py "/tmp/pmc_$ctr/**/*.db" > $O/pmc_$ctr.txt 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_traffic.json
rm -rf /tmp/pmc_mfma
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace -d /tmp/pmc_mfma -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-images --no-graph --no-sample-streams > $O/pmc_mfma.log 2>&1
python $R/tools/pmc_dump.py "/tmp/pmc_mfma/**/*.db" > $O/pmc_mfma.txt 2>&1
python $R/tools/pmc_mfma.py $O/pmc_mfma.txt > $O/mfma_util.txt 2>&1; tail -8 $O/mfma_util.txt
head -8 $O/kernel_stats_one_stream.csv
