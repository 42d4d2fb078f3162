#!/bin/bash
# The scaling runs of the hot path on ONE node: bench.py at 1 / 2 / 4 / 8 ranks (one process per GPU, RCCL over xGMI), both
# layouts - 'batch' (every rank owns whole samples, no data-path collective: BASELINE.json configs[1] per GPU) and 'frames'
# (frames sharded for pooling, one exchange of the pooled BEV maps in front of the temporal model: configs[2]; all-to-all-v and
# all-gather).  Each line carries per-rank `exchange_ms` / `bytes_received` (ranks.devices[*]) so a SCALE record explains itself.
#   tools/runs/scale.sh [max_ranks] [steps] [warmup]        -> gpurun_out/scale/*.json + summary.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
MAXN=${1:-8}; STEPS=${2:-20}; WARMUP=${3:-5}
O=gpurun_out/scale
mkdir -p $O
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
for n in 1 2 4 8; do
  [ $n -gt $MAXN ] && break
  [ $n -gt $NGPU ] && { echo "only $NGPU GPUs visible: stopping before $n ranks"; break; }
  for mode in "batch" "frames all_to_all" "frames all_gather"; do
    set -- $mode; layout=$1; ex=${2:-all_to_all}
    tag=${n}gpu_${layout}_${ex}
    common="--gpus $n --steps $STEPS --warmup $WARMUP --layout $layout --exchange $ex --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs"
    if [ $n -eq 1 ] && [ "$layout" = "batch" ]; then
      timeout 900 python bench.py $common > $O/$tag.json 2> $O/$tag.err
    else
      # (one rank through the process group too: FIERY_BENCH_FORCE_DIST=1 - the exchange then really is an RCCL call)
      FIERY_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port $((29500 + n)) bench.py $common > $O/$tag.json 2> $O/$tag.err
    fi
    python - <<PY
import json
try:
    d = json.loads([l for l in open('$O/$tag.json').read().strip().splitlines() if l.startswith('{')][-1])     # (RCCL prints its versions behind the line)
    ex = [(r.get('exchange_ms'), r.get('bytes_received')) for r in d['ranks']['devices']]
    print('%-28s %8.1f samples/s  %7.3f ms/step  launch: %s  exchange (ms, bytes received) per rank: %s' % ('$tag', d['value'], d['ms_per_step'], d['config']['launch'][:40], ex))
except Exception as e:
    print('$tag FAILED', e, open('$O/$tag.err').read()[-800:])
PY
  done
done 2>&1 | tee $O/summary.txt
