#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_q
mkdir -p $O
for t in 3 12; do
hipcc --offload-arch=gfx950 -O3 -w -DTILES=$t tools/probe/mfma_epilogue_probe.hip -o /tmp/mfma_epilogue_probe && timeout 120 /tmp/mfma_epilogue_probe 2>&1
done | tee $O/mfma_epilogue_probe.txt
