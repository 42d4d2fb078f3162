#!/bin/bash
# Round 5: where one training step goes - kernel table (rocprofv3 --kernel-trace --stats) of tools/time_train_step.py [extra args, e.g. --from-images]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_train_prof
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/tr_prof -o tr -- python tools/time_train_step.py --batch 2 --steps 5 "$@" > $O/run.log 2>&1
grep time_train_step $O/run.log | tee $O/summary.txt
db=$(find /tmp/tr_prof -name '*.db' | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_train.csv "round 5: rocprofv3 --kernel-trace --stats -- python tools/time_train_step.py --batch 2 --steps 5 $* (2 warm-up + 5 timed steps + set-up)" | tail -1
head -40 $O/kernel_stats_train.csv | cut -c1-170
