#!/bin/bash
# Round 6: where the pooling prepass spends its 20 us - timing experiments (wrong results on purpose: -DPRE_EXP=n builds), rocprof kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r6_i}
mkdir -p $O
for lib in $LIBS; do
  var=""; [ "$lib" != "-" ] && var="FIERY_HIP_LIB=$PWD/tools/ab/libfiery_hip_$lib.so"
  rm -rf /tmp/prof_$lib
  env $var ROUNDS=1 REPS=20 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$lib -o pool -- python tools/runs/r6_pool_ab.py "NO_RANKS=1" > /tmp/run_$lib.txt 2>&1
  db=$(ls /tmp/prof_$lib/*.db /tmp/prof_$lib/*/*.db 2>/dev/null | head -1)
  echo "== $lib"; python tools/rocprof_pool_phases.py $db
done | tee $O/prepass_experiments.txt
