#!/bin/bash
# GPU-box diagnostic batch (round 1, session 2): conv clocks, pooling unit-order/tile sweep, per-launch table, SQ counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
timeout 300 python tools/microbench.py conv --clk --reps 10 > $O/conv_clk.txt 2>&1
for o in 0 1; do
  FIERY_POOL_ORDER=$o POOL_TILES=20480,13334,10000,8000,6667,5000 timeout 300 python tools/microbench.py pool --reps 10 > $O/pool_order$o.txt 2>&1
done
FIERY_BENCH_DUMP=$O/launches.json timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq -- python tools/microbench.py conv --reps 2 > $O/pmc_run.txt 2>&1
python tools/pmc_dump.py "$O/pmc_sq/**/*.db" > $O/pmc_sq.txt 2>&1
rm -rf $O/pmc_sq
tail -3 $O/conv_clk.txt; tail -2 $O/pool_order1.txt; cat $O/bench.json | cut -c1-400
