#!/bin/bash
# Round 5, first GPU call: the two probes round 4 left unrun.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_a
mkdir -p $O
timeout 120 tools/probe/_bin/xcd_partials_probe 2>&1 | tee $O/xcd_partials.txt
timeout 120 python tools/probe/run_chained_gpu.py 2>&1 | tee $O/chained.txt
