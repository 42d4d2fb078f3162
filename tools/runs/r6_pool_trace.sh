#!/bin/bash
# Round 6: kernel trace of the pooling A/B harness: per-op phases (back-to-back / behind a 512 MB copy) and a raw timeline of a few ops
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_b}
mkdir -p $O
ROUNDS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o pool -- python tools/runs/r6_pool_ab.py ${VARIANT:-"-"} > $O/run.txt 2>&1
grep "^\[" $O/run.txt
db=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1)
python tools/rocprof_pool_phases.py $db | tee $O/phases.txt
python tools/rocprof_timeline.py $db k_rank_columns 10 ${ROWS:-12} | tee $O/timeline.txt
rm -rf $O/prof
