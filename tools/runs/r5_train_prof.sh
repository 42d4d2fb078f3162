#!/bin/bash
# Round 5: where one training step's 55 ms goes - kernel table (rocprofv3 --kernel-trace --stats) of tools/time_train_step.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_train_prof
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o tr -- python tools/time_train_step.py --batch 2 --steps 5 > $O/run.log 2>&1
grep time_train_step $O/run.log | tee $O/summary.txt
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' | tee -a $O/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# kernels by total time (whole process: warm-up + 5 timed steps + set-up); total {tot/1e6:.1f} ms')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:150]}")
PY
rm -rf $O/prof/*/*.db 2>/dev/null
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
