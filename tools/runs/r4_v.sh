#!/bin/bash
# Round 4: does the slower half of the pooling kernel's first round follow the dispatch order or the frames (addresses)?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_v
mkdir -p $O
TRACE=1 TRACE_DUMP=$O ROUNDS=1 REPS=20 timeout 600 python tools/runs/r4_pool_ab.py "PY_CLEAN=1" "PY_CLEAN=1,FIERY_POOL_LATE_PRIO=-1" 2>&1 | grep -v amdgpu.ids | tee $O/pool_reverse.txt
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob('gpurun_out/r4_v/trace_*.npy')):
    t=np.load(f); t=t[t[:,0]>0]; us=(t-t[:,0].min())/100.
    rows=(us[:512,2]-us[:512,1])
    print(f.split('/')[-1], 'rows by item block of 64:', [round(float(rows[i*64:(i+1)*64].mean()),1) for i in range(8)])
PY
