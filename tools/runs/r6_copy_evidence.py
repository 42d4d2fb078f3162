#!/usr/bin/env python
"""Copies the files of `tools/runs/r6_final.sh` from gpurun_out/r6_final into profiles/r6_* (bench lines reduced to their JSON line,
PMC summaries stamped with the commit they were measured at) and prints the figures the documents quote."""
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S, P = os.path.join(ROOT, 'gpurun_out', 'r6_final'), os.path.join(ROOT, 'profiles')
H = subprocess.run(['git', '-C', ROOT, 'log', '-1', '--format=%h'], capture_output=True, text=True).stdout.strip()


def cp(a, b):
    shutil.copy(os.path.join(S, a), os.path.join(P, b))


def json_line(path):
    for line in open(path):
        if line.startswith('{'):
            return line
    raise ValueError(path)


for a in ('bench.json', 'bench_batch_layout_eager_one_stream.json', 'bench_frames_layout_all_to_all_rccl_1rank.json',
          'bench_frames_layout_all_gather_rccl_1rank.json', 'bench_baseline_bf16.json'):
    open(os.path.join(P, 'r6_' + a), 'w').write(json_line(os.path.join(S, a)))
for m in ('one_stream', 'sample_streams', 'sample_streams_graph'):
    cp(f'kernel_stats_{m}.csv', f'r6_kernel_stats_{m}.csv')
for a in ('launches.json', 'launches_bf16.json', 'launches_table.txt', 'launches_table_bf16.txt', 'mfma_util.txt', 'mfma_ceiling.txt',
          'pool_ceiling.txt', 'pytest_gpu.txt', 'parity_errors.json', 'smoke.txt', 'winograd_check.txt', 'stream_k_check.txt',
          'train_step.txt', 'conv_forms.json', 'split_bf16_probe.txt', 'winograd_split_times.txt', 'split_tile_times.txt', 'wgrad_times.txt'):
    cp(a, 'r6_' + a)
cp('pool/summary.txt', 'r6_pool_ab_final.txt')
cp('pool_trace/phases.txt', 'r6_pool_phases.txt')
cp('pool_trace/timeline.txt', 'r6_pool_timeline.txt')
for tag, name in (('f32', 'r6_pmc_traffic.json'), ('bf16', 'r6_pmc_traffic_bf16.json'), ('pon_bf16', 'r6_pmc_traffic_pon_bf16.json'),
                  ('lyft7_bf16', 'r6_pmc_traffic_lyft7_bf16.json')):
    t = json.load(open(os.path.join(S, f'pmc_traffic_{tag}.json')))
    t['_measured_at_commit'] = H
    t['_command'] = (f'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline '
                     f'--no-from-images --no-bf16-mode --no-secondary-configs --no-graph --no-sample-streams [{tag}] (tools/runs/r6_final.sh)')
    json.dump(t, open(os.path.join(P, name), 'w'), indent=1)
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        cp(f'pmc_{tag}_{c}.txt', f'r6_pmc_{tag}_{c}.txt')
print('copied at', H)
d = json.loads(json_line(os.path.join(S, 'bench.json')))
r, p = d['roofline'], d['roofline_pooling']
print('value', d['value'], 'ms', d['ms_per_step'])
print(r['kernel'][:24], r['achieved'], r['frac'], r['all_convolutions'], 'kernel_ms', r['kernel_ms_per_step'], 'raw', r['kernel_ms_per_step_raw_brackets'],
      'ovh', r['bracket_overhead_us'])
print('pool', p['frac'], p['op_us_samples'], (p['traffic_from_profiles'] or {}).get('bytes_per_launch'))
print('parity', d['parity_literal_1e-4']['outputs_passing'], [x['max_abs_err']['segmentation'] for x in d['parity_draws']])
print('cpu', round(d['cpu_baseline']['value'], 3), d['cpu_baseline']['cores'], 'bf16', d['bf16_mode']['value'])
for k, v in d['secondary_configs'].items():
    print(k, v.get('value'), v['roofline_pooling']['frac'], v['roofline']['bound_per_layer']['frac'])
print('images', d['forward_from_images']['ms_per_step'])
