#!/bin/bash
# the timed step in its launch modes on one box, alternating: graph + one stream per sample (default) / graph + one stream for the batch
mkdir -p gpurun_out/modes; O=gpurun_out/modes
A="--steps 20 --warmup 3 --no-cpu-baseline --single-parity-draw --no-secondary-configs --no-from-images --no-bf16-mode"
for r in 1 2 3; do for m in "" "--no-sample-streams"; do
  timeout 900 python bench.py $A $m 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('mode [$m]', d['value'], d['ms_per_step'], d.get('timed_mode'))"
done; done | tee $O/modes.txt
