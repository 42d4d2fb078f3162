#!/bin/bash
# Session 3, call 1: state of HEAD on a fresh box - GPU parity suite, bench line (+ per-launch dump), rocprofv3 kernel
# traces in both launch modes, the two PMC traffic passes, conv/pool microbenches.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/s3
mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
FIERY_BENCH_DUMP=$O/launches.json timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
cut -c1-600 $O/bench.json
cd /tmp
for mode in one_stream sample_streams; do
  extra=""; [ $mode = one_stream ] && extra="--no-sample-streams"
  rm -rf /tmp/kt_$mode
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_$mode -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra > $GRAFT_REPO_ROOT/$O/kt_$mode.log 2>&1
  db=$(find /tmp/kt_$mode -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py "$db" $GRAFT_REPO_ROOT/$O/kernel_stats_$mode.csv "round 1, session 3 ($mode): rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline $extra"
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $GRAFT_REPO_ROOT/$O/pmc_$ctr.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py "/tmp/pmc_$ctr/**/*.db" > $GRAFT_REPO_ROOT/$O/pmc_$ctr.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.txt $O/pmc_WRITE_SIZE.txt $O/pmc_traffic.json > /dev/null 2>&1
timeout 300 python tools/microbench.py pool --probe --reps 10 > $O/pool_probe.txt 2>&1
head -12 $O/kernel_stats_one_stream.csv
grep "pool\|probe" $O/pool_probe.txt | head
