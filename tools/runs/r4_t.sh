#!/bin/bash
# Round 4: does the timed region of bench.py run at the clock the chip settles at?  (warm-up length against samples/s)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_t
mkdir -p $O
for cfg in "20 3" "20 100" "20 300" "100 300" "20 3"; do
  set -- $cfg
  echo "== steps $1 warmup $2"
  timeout 600 python bench.py --steps $1 --warmup $2 --no-from-images --no-bf16-mode --no-cpu-baseline --no-secondary-configs | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['timed_mode']['frac'], d['roofline_pooling']['op_us_per_step'])"
done | tee $O/warmup_ab.txt
