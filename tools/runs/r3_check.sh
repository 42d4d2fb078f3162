#!/bin/bash
# Round 3: the tree at HEAD once more after the evidence run (the halo block was touched again for its fp32 form): the whole GPU
# suite, smoke(), the fp32 and the bf16 bench lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_check
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; cut -c1-260 $O/bench.json
timeout 300 python bench.py --steps 20 --warmup 3 --precision bf16 --no-from-images --no-cpu-baseline > $O/bench_baseline_bf16.json 2>> $O/bench.err
grep -h -o '"value": [0-9.]*' $O/bench_baseline_bf16.json | head -1
timeout 300 python bench.py --steps 10 --warmup 3 --precision bf16 --config literature/pon_setting.yml --no-from-images --no-cpu-baseline > $O/bench_pon_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --precision bf16 --config lyft/baseline.yml --cams 7 --no-from-images --no-cpu-baseline > $O/bench_lyft7_bf16.json 2>> $O/bench.err
grep -h -o '"value": [0-9.]*' $O/bench_pon_bf16.json $O/bench_lyft7_bf16.json | grep -v '"value": 0\.'
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bf16_mode', d.get('bf16_mode'))"
