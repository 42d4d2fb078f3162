#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s3b
mkdir -p $O
(timeout 300 python tools/runs/probe_rows.py 10 2>&1 | grep "rows dealt"
cd /tmp; rm -rf /tmp/pmcx
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcx -o pmc -- python $GRAFT_REPO_ROOT/tools/runs/probe_rows.py 2 > /tmp/pmcx.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_dump.py "/tmp/pmcx/**/*.db" 2>&1 | grep "k_read_columns\|k_read_planes") 2>&1 | tee $O/probe_rows.txt
