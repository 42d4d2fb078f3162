#!/bin/bash
# Round 5: whole-step parity under switches (which convolution form breaks it?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_k
mkdir -p $O
for v in "FIERY_CONV_WINOGRAD=1" "FIERY_CONV_WINOGRAD=1 FIERY_WINOGRAD_GENERAL_EPILOGUE=1" "FIERY_CONV_WINOGRAD=1 FIERY_STREAM_K=0" "FIERY_CONV_WINOGRAD=1 NSS=1" "FIERY_CONV_WINOGRAD=1 NOGRAPH=1" "FIERY_CONV_WINOGRAD=1"; do
  extra=""; [[ "$v" == *NSS=1* ]] && extra="--no-sample-streams"; [[ "$v" == *NOGRAPH=1* ]] && extra="--no-graph"
  vv=${v% NSS=1}; vv=${vv% NOGRAPH=1}
  env $vv timeout 300 python bench.py --steps 10 --warmup 3 --no-from-images --no-bf16-mode --no-secondary-configs $extra 2>$O/err.txt > $O/b.json
  python - <<PY
import json
try:
    d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
    par=d.get('parity', {})
    print('%-62s %.1f samples/s | %s | parity: %s' % ('$v', d['value'], d['config']['launch'][:30], {k: v['max_abs_err'] for k, v in par.items() if k in ('segmentation', 'instance_flow')}))
except Exception as e:
    print('$v FAILED', e, open('$O/err.txt').read()[-800:])
PY
done 2>&1 | tee $O/summary.txt
