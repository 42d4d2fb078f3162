#!/bin/bash
# Round 5: stream-K convolution form: GPU correctness (stream-K against the tile form, many parts, repeated) + same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_c
mkdir -p $O
timeout 300 python tools/runs/r5_sk_check.py 2>&1 | tail -30 | tee $O/sk_check.txt
for rep in 1 2; do
for v in "FIERY_STREAM_K=0" "FIERY_STREAM_K=1" "FIERY_STREAM_K=1 NSS=1" "FIERY_STREAM_K=0 NSS=1"; do
  extra=""; [[ "$v" == *NSS=1* ]] && extra="--no-sample-streams"
  env ${v% NSS=1} FIERY_BENCH_DUMP=$O/dump_$(echo $v | tr -d ' =')_$rep.json timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs $extra 2>$O/err.txt > $O/b.json
  python - <<PY
import json
try:
    d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
    r=d['roofline']; rp=d['roofline_pooling']
    print('%-28s rep $rep: %.1f samples/s  conv frac %.4f (timed %.4f) kernel ms %.3f  pool %.1f us (%.4f)' % ('$v', d['value'], r['frac'], r['timed_mode']['frac'], r['kernel_ms_per_step'], rp['op_us_per_step'], rp['frac']))
except Exception as e:
    print('$v rep $rep FAILED', e, open('$O/err.txt').read()[-1500:])
PY
done; done 2>&1 | tee $O/summary.txt
python tools/launches_table.py $O/dump_FIERY_STREAM_K0_2.json $O/dump_FIERY_STREAM_K1_2.json > $O/launches_ab.txt 2>&1
head -32 $O/launches_ab.txt
