#!/bin/bash
# Round 5: Winograd form: GPU correctness + per-form timing, then the step with and without it (same box, alternating)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_e
mkdir -p $O
timeout 300 python tools/runs/r5_wino_check.py 2>&1 | tail -14 | tee $O/wino_check.txt
for rep in 1 2; do
for v in "FIERY_CONV_WINOGRAD=0" "FIERY_CONV_WINOGRAD=1"; do
  env $v FIERY_BENCH_DUMP=$O/dump_$(echo $v | tr -d ' =')_$rep.json timeout 300 python bench.py --steps 20 --warmup 5 --no-from-images --no-bf16-mode --no-secondary-configs 2>$O/err.txt > $O/b_$(echo $v | tr -d ' =')_$rep.json
  python - <<PY
import json
try:
    d=json.loads(open('$O/b_$(echo $v | tr -d ' =')_$rep.json').read().strip().splitlines()[-1])
    r=d['roofline']; rp=d['roofline_pooling']
    par=d.get('parity', {})
    print('%-24s rep $rep: %.1f samples/s  conv frac %.4f kernel ms %.3f  pool %.1f us | parity max_abs_err: %s' % ('$v', d['value'], r['frac'], r['kernel_ms_per_step'], rp['op_us_per_step'], {k: v['max_abs_err'] for k, v in par.items()}))
except Exception as e:
    print('$v rep $rep FAILED', e, open('$O/err.txt').read()[-1500:])
PY
done; done 2>&1 | tee $O/summary.txt
python tools/launches_table.py $O/dump_FIERY_CONV_WINOGRAD0_2.json > $O/launches_direct.txt 2>&1
python tools/launches_table.py $O/dump_FIERY_CONV_WINOGRAD1_2.json > $O/launches_wino.txt 2>&1
head -24 $O/launches_wino.txt
