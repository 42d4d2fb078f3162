#!/bin/bash
# Round 4, last commit: the default bench line, smoke(), the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4_check
mkdir -p $O
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench.err
cut -c1-260 $O/bench_default.json | tail -1
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -2
