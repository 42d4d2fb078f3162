#!/bin/bash
# Round 5: graph capture on the library's capture stream (stream-K workspace ready): the bit-equality test, the frames layout
# replayed from a graph at one rank (through RCCL), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5_p
mkdir -p $O
timeout 600 python -m pytest tests/test_calibration_table.py tests/test_gpu_parity.py -q -x -m gpu -k "table_mode or graph_replay or per_sample_streams or frame_sharded or forward_graph" 2>&1 | tail -4 | tee $O/pytest.txt
for ex in all_to_all all_gather; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 FIERY_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --layout frames --exchange $ex --no-cpu-baseline --no-from-images --no-bf16-mode --no-secondary-configs > $O/frames_$ex.json 2> $O/frames_$ex.err
  python - <<PY
import json
try:
    d=json.loads(open('$O/frames_$ex.json').read().strip().splitlines()[-1])
    print('frames layout, $ex, 1 rank through RCCL: %.1f samples/s | %s | ranks: %s' % (d['value'], d['config']['launch'], [(r.get('exchange_ms'), r.get('bytes_received'), r.get('kind')) for r in d['ranks']['devices']]))
except Exception as e:
    print('frames $ex FAILED', e, open('$O/frames_$ex.err').read()[-1500:])
PY
done 2>&1 | tee $O/frames.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary-configs > $O/bench.json 2> $O/bench.err
python - <<PY | tee $O/bench_summary.txt
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; rp=d['roofline_pooling']
print('batch layout: %.1f samples/s (%.3f ms) | conv executed frac %.4f, direct-form-equivalent %.1f TFLOP/s, algorithmic speed-up %.3f | by form: %s' % (d['value'], d['ms_per_step'], r['frac'], r['direct_form_equivalent_tflops'], r['algorithmic_speedup']['value'], {k: (v['launches'], v['ms_per_step'], v['executed_frac_of_peak']) for k, v in r['by_form'].items()}))
print('pooling: %.1f us, frac %.4f, of ceiling %s' % (rp['op_us_per_step'], rp['frac'], rp.get('ceiling_from_profiles', {}).get('frac_of_ceiling')))
print('parity:', {k: (v['max_abs_err'], v['within_1e-4']) for k, v in d['parity'].items()}, d['parity_literal_1e-4'])
print('bf16 mode:', d.get('bf16_mode', {}).get('value'), '| from images:', d.get('forward_from_images', {}).get('ms_per_step'), 'ms | cpu baseline', d['cpu_baseline']['value'])
PY
