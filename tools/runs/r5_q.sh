#!/bin/bash
# Round 5: Winograd with four / eight wavefronts per workgroup, same box, alternating: per-shape launch times, then the step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_q
mkdir -p $O
timeout 60 python tools/runs/r5_wino_times.py > /dev/null 2>&1
for rep in 1 2; do for w in 4 8; do
  echo "== $w wavefronts (rep $rep)"
  FIERY_WINOGRAD_WAVES=$w timeout 120 python tools/runs/r5_wino_times.py 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee $O/times.txt
FIERY_WINOGRAD_WAVES=8 timeout 200 python tools/runs/r5_wino_check.py 2>&1 | tail -3 | tee $O/check8.txt
for rep in 1 2; do for w in 4 8; do
  FIERY_WINOGRAD_WAVES=$w timeout 300 python bench.py --steps 20 --warmup 5 --no-from-images --no-bf16-mode --no-secondary-configs 2>/dev/null > $O/b.json
  python - <<PY
import json
d=json.loads([l for l in open('$O/b.json').read().splitlines() if l.startswith('{')][-1])
r=d['roofline']
print('$w wavefronts rep $rep: %.1f samples/s  conv executed frac %.4f kernel ms %.3f | parity seg %s flow %s' % (d['value'], r['frac'], r['kernel_ms_per_step'], d['parity']['segmentation']['max_abs_err'], d['parity']['instance_flow']['max_abs_err']))
PY
done; done 2>&1 | tee $O/step.txt
