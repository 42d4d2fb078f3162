#!/bin/bash
# Round 6: in-situ durations of the pooling kernels (and any other) inside EXACTLY 20 one-stream steps, per library build:
# rocprofv3 kernel stats of tools/runs/r6_profile_steps.py with frozen convolution forms (LIBS="- name ...", KERNELS=regex)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${TAG:-r6_insitu}
mkdir -p $O
cd /tmp
timeout 300 python $R/tools/runs/r6_profile_steps.py tune /tmp/forms.json > /dev/null 2>&1
for rep in 1 2; do for lib in $LIBS; do
  var=""; [ "$lib" != "-" ] && var="FIERY_HIP_LIB=$R/tools/ab/libfiery_hip_$lib.so"
  rm -rf /tmp/kt_$lib
  env $var timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$lib -o kt -- python $R/tools/runs/r6_profile_steps.py run /tmp/forms.json --steps 20 ${EXTRA} > /tmp/kt_$lib.log 2>&1
  db=$(find /tmp/kt_$lib -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py "$db" /tmp/kt_$lib.csv x > /dev/null
  echo "== $lib (rep $rep)"; grep -E "${KERNELS:-k_voxel_pool|k_rank_columns|k_lift_geometry}" /tmp/kt_$lib.csv
done; done | tee $O/summary.txt
