#!/bin/bash
# Round 4: kernel trace of the image trunk + lift head on the engine (54 images of 224 x 480)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r4_n
mkdir -p $O
cd /tmp
rm -rf /tmp/kt_trunk
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt_trunk -o kt -- python $R/tools/runs/trunk_profile.py > $O/kt.log 2>&1
db=$(find /tmp/kt_trunk -name "*.db" | head -1)
python $R/tools/rocprof_summary.py "$db" $O/kernel_stats_trunk.csv "round 4: rocprofv3 --kernel-trace --stats -- python tools/runs/trunk_profile.py (4 passes of the trunk, 54 images)" 2>&1 | tail -1
head -30 $O/kernel_stats_trunk.csv | cut -c1-180
tail -25 $O/kt.log | cut -c1-200
