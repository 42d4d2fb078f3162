#!/bin/bash
# Round 5: kernel table of the forward pass from images (trunk + lift head + hot path), hipGraph replays
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_x
mkdir -p $O
timeout 600 python tools/runs/r5_images_profile.py 2>&1 | tail -2 | tee $O/time.txt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_img -o kt -- python tools/runs/r5_images_profile.py > $O/kt.log 2>&1
db=$(find /tmp/kt_img -name '*.db' | head -1)
python tools/rocprof_summary.py "$db" $O/kernel_stats_from_images.csv "round 5: rocprofv3 --kernel-trace --stats -- python tools/runs/r5_images_profile.py (tuning + 13 replays of the forward pass from 54 images)" | tail -2
head -45 $O/kernel_stats_from_images.csv | cut -c1-180
