#!/bin/bash
# the step with the split Winograd form among the candidates, beside the step without it (FIERY_CONV_WINOGRAD_SPLIT=0), alternating
mkdir -p gpurun_out/wsplit; O=gpurun_out/wsplit
A="--steps 20 --warmup 3 --no-cpu-baseline --no-secondary-configs --no-from-images --no-bf16-mode"
for r in 1 2; do for v in 1 0; do
  FIERY_CONV_WINOGRAD_SPLIT=$v FIERY_BENCH_DUMP=$O/l_split$v.json timeout 900 python bench.py $A > $O/b_split${v}_$r.json 2>$O/err_$v.txt
  python - $O/b_split${v}_$r.json $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
r=d['roofline']
print('split' if sys.argv[2]=='1' else 'fp32 ', d['value'], d['ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'dom', r['kernel'][:40], r['frac'], 'parity', {k: v for k, v in d['parity'].items() if k in ('max_abs_err','worst','within_1e-4_literal','outputs')} if isinstance(d.get('parity'), dict) else d.get('parity'))
PY
done; done 2>&1 | tee $O/bench_ab.txt
python tools/launches_table.py $O/l_split1.json > $O/t_split1.txt; python tools/launches_table.py $O/l_split0.json > $O/t_split0.txt
