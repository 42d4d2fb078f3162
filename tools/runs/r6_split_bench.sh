#!/bin/bash
# the step with the split TILE form among the candidates, beside the step without it (FIERY_CONV_SPLIT=0), alternating
mkdir -p gpurun_out/split; O=gpurun_out/split
A="--steps 20 --warmup 3 --no-cpu-baseline --no-secondary-configs --no-from-images --no-bf16-mode"
for r in 1 2; do for v in 1 0; do
  FIERY_CONV_SPLIT=$v FIERY_BENCH_DUMP=$O/l_split$v.json timeout 900 python bench.py $A > $O/b_split${v}_$r.json 2>$O/err_$v.txt
  python - $O/b_split${v}_$r.json $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
r=d['roofline']
print('tiles+' if sys.argv[2]=='1' else 'tiles-', d['value'], d['ms_per_step'], 'conv ms', r['kernel_ms_per_step'], 'dom', r['kernel'][:30], r['frac'], 'all', r['all_convolutions']['frac'], {k: (v['launches'], v['ms_per_step']) for k, v in r['by_form'].items()})
PY
done; done 2>&1 | tee $O/bench_ab.txt
python tools/launches_table.py $O/l_split1.json > $O/t_split1.txt
