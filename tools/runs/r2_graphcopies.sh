#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for extra in "" "--no-graph" "--no-sample-streams" "--no-graph --no-sample-streams"; do
  echo "== bench $extra"
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-from-images $extra 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['config']['launch'])"
done
