#!/usr/bin/env python
"""Headline benchmark: BEV samples/s of the FIERY camera-to-BEV hot path on MI355X.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (geometry -> voxel pooling -> ego-warp -> temporal model -> present
distribution -> SpatialGRU future prediction -> decoder) over one batch of synthetic input that is already
resident in HBM: BASELINE.json configs[1], `baseline.yml`, 6 cameras x 3 past frames -> 200x200 BEV, batch 3
per GPU, fp32.  The inputs are the image encoder's outputs (the trunk is upstream of the path).
With N > 1 every rank runs its own batch of 3 (the path is independent per sample: no data-path collective),
so the job processes 3*N samples per step: weak scaling.  `--layout frames` measures BASELINE.json configs[2] instead:
the global batch's frames are split across the ranks for geometry + pooling, ONE RCCL all-gather moves the pooled BEV
maps, then every rank runs the temporal / future / decoder stack of its own samples (same work per rank, plus the
exchange).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X dense bf16 matrix peak (no sparsity)
PEAK_HBM_GBS = 8000.0             # MI355X HBM3E spec bandwidth


def pmc_workload(config, precision, n_cam, batch):
    """Which committed pair of PMC passes belongs to this run's workload (None: none was taken for it)."""
    if batch != 3:
        return None
    if config == 'baseline.yml' and n_cam == 6:
        return 'f32' if precision == 'f32' else 'bf16'
    if config == 'literature/pon_setting.yml' and n_cam == 6 and precision == 'bf16':
        return 'pon_bf16'
    if config == 'lyft/baseline.yml' and n_cam == 7 and precision == 'bf16':
        return 'lyft7_bf16'
    return None


def pmc_traffic(kernels, mode='f32'):
    """HBM bytes per launch of `kernels` (summed) from the COMMITTED PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over
    this bench command - baseline.yml, fp32, batch 3 - folded by tools/pmc_traffic.py with the gfx950 correction), with the
    file they come from and the commit that file was last changed in; None when the summary is absent.  Callers report it for
    THAT workload only: any other configuration's line carries null.  Counters cannot be collected from inside the timed
    process, so this is the one roofline field not measured live - hence its name in the line, `traffic_from_profiles`."""
    import subprocess
    if mode is None:
        return None
    names = ('r6_pmc_traffic.json', 'r5_pmc_traffic.json', 'r4_pmc_traffic.json', 'r3_pmc_traffic.json', 'r2_pmc_traffic.json', 'r1_pmc_traffic.json')
    if mode == 'bf16':                                   # (round 5: the same two passes over `bench.py --precision bf16`)
        names = ('r6_pmc_traffic_bf16.json', 'r5_pmc_traffic_bf16.json')
    elif mode != 'f32':                                  # (round 6: pon and lyft-7 in bf16 mode - BASELINE.json configs[3] / [4])
        names = (f'r6_pmc_traffic_{mode}.json',)
    for name in names:
        path = os.path.join(ROOT, 'profiles', name)
        try:
            table = json.load(open(path))
            per_kernel = {}
            for k in kernels:
                hits = [v['traffic_bytes'] for key, v in table.items() if isinstance(v, dict) and (key == k or key.startswith(k))]
                if not hits:
                    raise KeyError(k)
                per_kernel[k] = hits[0]
        except (OSError, KeyError, ValueError):
            continue
        try:
            commit = subprocess.run(['git', '-C', ROOT, 'log', '-1', '--format=%h', '--', os.path.join('profiles', name)],
                                    capture_output=True, text=True, timeout=10).stdout.strip() or None
        except (OSError, subprocess.SubprocessError):
            commit = None
        # (a GPU box's snapshot has no .git: the summary itself names the commit its passes ran at)
        commit = commit or table.get('_measured_at_commit')
        return {'bytes_per_launch': sum(per_kernel.values()), 'per_kernel': per_kernel, 'file': 'profiles/' + name, 'commit': commit,
                'measured_at_commit': table.get('_measured_at_commit'),
                'what': '2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over this command (not collected in this run)'}
    return None


def pool_ceiling():
    """The pooling op's ceiling on an MI355X box as the committed probe measured it (tools/probe/pool_ceiling.hip ->
    profiles/r*_pool_ceiling.txt): the time of a kernel that moves exactly the op's algorithmic bytes and does nothing else, in the
    best pattern found (pure streaming, and the op-shaped (channel, frame) units over the encoder's native layout)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_pool_ceiling.txt')))
    if not files:
        return None
    text = open(files[-1]).read()
    m = re.search(r'CEILING: ([0-9.]+) us for the op\'s ([0-9.]+) MB .*\(streaming ([0-9.]+) us, op-shaped ([0-9.]+) us\)', text)
    if not m:
        return None
    same_box = re.search(r'same box, bench.py: pooling op ([0-9.]+) us', text)
    return {'ceiling_us': float(m.group(1)), 'algorithmic_mb': float(m.group(2)), 'streaming_us': float(m.group(3)),
            'op_shaped_us': float(m.group(4)), 'op_us_on_the_probe_box': float(same_box.group(1)) if same_box else None,
            'file': 'profiles/' + os.path.basename(files[-1]),
            'what': 'tools/probe/pool_ceiling.hip: exactly the algorithmic reads + the output planes, nothing computed; '
                    'streaming = one contiguous buffer, op-shaped = 576 (channel, frame) units over the (n, C, D, H, W) tensor'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=3, help='samples per GPU (baseline.yml BATCHSIZE)')
    ap.add_argument('--config', default='baseline.yml')
    ap.add_argument('--cams', type=int, default=0, help='cameras per frame (0 = the preset\'s IMAGE.NAMES; lyft runs use 7)')
    ap.add_argument('--exchange', choices=('all_gather', 'all_to_all'), default='all_to_all',
                    help="frames layout: 'all_gather' moves every pooled frame to every rank, 'all_to_all' each frame to the one "
                         'rank that owns its sample (fiery_amd.parallel.FrameScatter)')
    ap.add_argument('--layout', choices=('batch', 'frames'), default='batch',
                    help='batch: every rank owns whole samples, no collective.  frames: frames sharded for pooling, one '
                         'all-gather of the BEV maps, then batch-sharded (BASELINE.json configs[2])')
    ap.add_argument('--precision', choices=('f32', 'bf16'), default='f32',
                    help='matrix-core precision of the convolutions: f32 = the reference\'s arithmetic (configs[1], the parity '
                         'configuration); bf16 = operands rounded at the matrix cores, fp32 accumulate (configs[3] / [4])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--single-parity-draw', action='store_true', help='parity against the oracle on this run\'s inputs only (default: three input draws, the worst counts)')
    ap.add_argument('--no-secondary-configs', action='store_true',
                    help='skip the secondary lines of BASELINE.json configs[3] / [4] (pon_setting.yml bf16; lyft/baseline.yml with 7 cameras, bf16, one GPU)')
    ap.add_argument('--no-from-images', action='store_true', help='skip the secondary forward-from-images timing')
    ap.add_argument('--no-bf16-mode', action='store_true', help='skip the secondary timing of the same step with bf16 matrix-core operands')
    ap.add_argument('--no-graph', action='store_true', help='enqueue every launch from the host each step instead of '
                                                            'replaying the captured hipGraph')
    ap.add_argument('--no-sample-streams', action='store_true', help='one stream for the whole batch instead of one '
                                                                      'independent chain (HIP stream) per sample')
    ap.add_argument('--fused', action='store_true', help='feed depth logits + features to the fused lift-splat kernel '
                                                         'instead of the materialised outer product')
    return ap.parse_args()


def parity_rows(got, want):
    """Per output: the max-abs error over ALL samples and per sample, the literal 1e-4 bar and the scaled bar of the tests."""
    rows = {}
    for k, v in want.items():
        if v is None:
            continue
        err = (got[k] - v).abs()
        per_sample = err.reshape(err.shape[0], -1).max(dim=1).values
        worst, ref_max = per_sample.max().item(), v.abs().max().item()
        rows[k] = {'max_abs_err': float(f'{worst:.3e}'), 'max_abs_err_per_sample': [float(f'{e:.3e}') for e in per_sample.tolist()],
                   'ref_abs_max': round(ref_max, 3),
                   'within_1e-4': bool(worst <= 1e-4),                                  # (the fp32 configuration's literal bar)
                   'within_scaled': bool(worst <= 1e-4 * max(1.0, ref_max)),          # (the bar tests/test_gpu_parity.py asserts)
                   # the literal bar element by element: how many of the output's values pass it
                   'elements_within_1e-4': int((err <= 1e-4).sum().item()), 'elements': v.numel()}
    return rows


def cpu_baseline(cfg, sd, lifted, K, E, ego, runs=3):
    """The oracle (a port of the reference's CPU path on the same ATen CPU kernels) on the host cores, on a bounded sample of
    the same workload: one batch element (median of `runs` after a warm-up) and, once, the whole batch of the headline
    configuration (BASELINE.md's north-star row is quoted at that batch).  Returns the baseline dict and the WHOLE batch's
    outputs (the parity check of the GPU result rides on them: every sample, not the first)."""
    from oracle import bev_stack
    cores = len(os.sched_getaffinity(0))
    try:                                               # a cgroup CPU quota caps the usable cores below the affinity mask
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(cores)
    one = [t[:1].cpu() for t in (lifted, K, E, ego)]
    whole = [t.cpu() for t in (lifted, K, E, ego)]
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    times = []
    with torch.no_grad():
        out = bev_stack.bev_hot_path(sd_cpu, cfg, *one)    # warm-up (first-call allocator / oneDNN primitive caches)
        for _ in range(runs):
            t0 = time.perf_counter()
            out = bev_stack.bev_hot_path(sd_cpu, cfg, *one)
            times.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        out = bev_stack.bev_hot_path(sd_cpu, cfg, *whole)       # (its outputs: every sample of the batch, for the parity check)
        t_batch = time.perf_counter() - t0
    dt = sorted(times)[len(times) // 2]
    n_batch = whole[0].shape[0]
    return {'value': 1.0 / dt, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'1 sample (batch 1 of the same workload), median of {runs} runs after 1 warm-up '
                      f'({", ".join(f"{t:.2f}" for t in times)} s)',
            'at_batch': {'batch': n_batch, 'value': round(n_batch / t_batch, 4), 'unit': 'samples/s',
                         'sample': f'the whole batch of {n_batch} once ({t_batch:.2f} s), same cores, after the runs above'}}, out


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    use_dist = world > 1 or os.environ.get('FIERY_BENCH_FORCE_DIST') == '1'     # (the env var: 1-rank RCCL dry run)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)     # RCCL over xGMI

    from fiery_amd import ops
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.model import Fiery
    from fiery_amd.synthetic import make_inputs, make_lifted_features

    cfg = get_preset_cfg(args.config)
    torch.manual_seed(0)
    model = Fiery(cfg).eval()
    from fiery_amd.synthetic import randomise_weights
    sd = randomise_weights(model)                          # random-init weights, non-trivial BN statistics
    model = model.to(dev)
    model.conv_precision = args.precision
    model.sample_streams = not args.no_sample_streams
    # the timed step computes everything from its inputs: the camera matrices on the device every step, not looked up in the
    # table of host-inverted calibrations that is the library's default (same step time: 293.3 against 293.6 samples/s on one box)
    model.camera_matrix_mode = 'device'

    B, rf, nf = args.batch, model.receptive_field, model.n_future
    n_cam = args.cams or len(cfg.IMAGE.NAMES)
    D = model.depth_channels
    fh, fw = cfg.IMAGE.FINAL_DIM[0] // 8, cfg.IMAGE.FINAL_DIM[1] // 8
    C = cfg.MODEL.ENCODER.OUT_CHANNELS
    frames_layout = args.layout == 'frames'
    if frames_layout:
        # the layout's entry point takes the GLOBAL batch (B samples per rank: B * world) and pools only this rank's share of
        # its frames.  Calibrations and ego-motion of every sample are generated everywhere (one generator per rank's
        # samples, so sample b is the same tensor whatever the world size); of the lifted features - 1.1 GB per rank - a rank
        # only materialises its own samples' (the frames it pools) inside a device tensor of the global shape.
        assert use_dist or world == 1
        parts = [make_inputs(B, rf + nf, n_cam, with_image=False, seed=r) for r in range(world)]
        K, E, ego = (torch.cat([p[i] for p in parts]) for i in (1, 2, 3))
        _, _, mine = make_lifted_features(B * rf * n_cam, C, D, (fh, fw), seed=100 + rank)
        lifted = mine.view(B, rf, n_cam, C, D, fh, fw)
    else:
        _, K, E, ego = make_inputs(B, rf + nf, n_cam, with_image=False, seed=rank)
        dl, ft, lifted = make_lifted_features(B * rf * n_cam, C, D, (fh, fw), seed=100 + rank)
        lifted = lifted.view(B, rf, n_cam, C, D, fh, fw)
    K_d, E_d, ego_d = K.to(dev), E.to(dev), ego.to(dev)
    if frames_layout:
        from fiery_amd.parallel import block_range, sharded_bev_forward
        lifted_d = torch.empty((B * world,) + tuple(lifted.shape[1:]), device=dev)
        assert block_range(B * world * rf, world, rank) == (rank * B * rf, (rank + 1) * B * rf)     # this rank pools its own samples' frames
        lifted_d[rank * B:(rank + 1) * B].copy_(lifted)
        from fiery_amd.parallel import sharded_bev_forward_graph
        eager_step = lambda: sharded_bev_forward(model, K_d, E_d, ego_d, lifted=lifted_d, layout='frames', exchange=args.exchange)[0]
        # (round 5: the exchange - enqueued by torch.distributed on the capturing stream - is a node of the captured graph)
        graph_step = lambda: sharded_bev_forward_graph(model, K_d, E_d, ego_d, lifted=lifted_d, layout='frames', exchange=args.exchange)[0]
    elif args.fused:
        dl_d = dl.view(B, rf, n_cam, D, fh, fw).to(dev)
        ft_d = ft.view(B, rf, n_cam, C, fh, fw).to(dev)
        eager_step = lambda: model.bev_forward(None, K_d, E_d, ego_d, depth_logits=dl_d, features=ft_d)
        graph_step = lambda: model.bev_forward_graph(None, K_d, E_d, ego_d, depth_logits=dl_d, features=ft_d)
    else:
        lifted_d = lifted.to(dev)
        eager_step = lambda: model.bev_forward(lifted_d, K_d, E_d, ego_d)
        graph_step = lambda: model.bev_forward_graph(lifted_d, K_d, E_d, ego_d)
    # The timed step replays the whole path (every kernel, same work) from one captured hipGraph; the first call
    # captures it and is checked against the eager path.
    step, launch_mode = eager_step, 'host enqueue per launch'
    if use_dist:
        # every rank runs rank 0's convolution forms (they differ in fp32 rounding; each rank's own timings could pick differently)
        from fiery_amd.parallel import share_conv_forms
        with torch.no_grad():
            eager_step()                                   # (tunes this rank's launches)
        torch.cuda.synchronize()
        share_conv_forms(freeze=False)                 # (later modes - the one-stream instrumented steps - may still tune their shapes)
    if not args.no_graph and graph_step is not None:
        try:
            with torch.no_grad():
                ref = {k: v.clone() for k, v in eager_step().items() if v is not None}
                got = graph_step()
                torch.cuda.synchronize()
                for k, v in ref.items():
                    assert (got[k] - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item()), k
            step, launch_mode = graph_step, 'hipGraph replay (one graph launch per step)'
        except Exception as e:                                       # noqa: BLE001  (report, then measure the eager path)
            launch_mode = f'host enqueue per launch (graph capture failed: {repr(e)[:160]})'

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        elapsed = time.perf_counter() - t0
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert all(torch.isfinite(v).all() for v in out.values() if v is not None)

    # one instrumented replica of the step, after the timed region, so the event records do not perturb `value`:
    # HIP events around every launch of the dominant kernel on the stream it is launched on
    roofline = pooling = None
    host_ms = None
    # (rank 0's figures; in the frames layout every step holds a collective, so there all ranks walk through the same steps)
    if rank == 0 or (frames_layout and use_dist and world > 1):
        torch.cuda.synchronize()
        t_host = time.perf_counter()
        with torch.no_grad():
            step()                                         # enqueue only: how long the host needs to issue one step
        host_ms = (time.perf_counter() - t_host) * 1e3
        torch.cuda.synchronize()
        # The kernel rooflines are per-kernel figures: they are taken with the whole batch on ONE stream, where a launch
        # has the GPU to itself (with a chain per sample three kernels share the CUs and every bracket stretches).
        streams_on, model.sample_streams = model.sample_streams, False
        with torch.no_grad():
            eager_step()                                   # also tunes this mode's launches, outside the brackets
            torch.cuda.synchronize()
            # five instrumented steps (same kernels, launched one by one so each can be bracketed): the figures below come from
            # the step whose convolution time is the median, the pooling op's from the median over the five (one step's
            # bracket of a 275 us op moves by +-5 us from run to run).  Each is enqueued BEHIND an unrecorded step: the GPU must
            # not wait for the host inside a bracket, and pooling is the first thing a step launches - behind a synchronise its
            # bracket would hold the host's time to issue the two kernels (~15 us; until round 4 four of the five did).
            instrumented = []
            for _ in range(5):
                ops.PROFILE_SINK = None
                # backlog: since round 5 the GPU finishes a step faster than the host issues one launch by launch, so an unrecorded
                # step in front no longer keeps the queue full - the GPU spins for ~25 ms instead while the host enqueues the
                # instrumented step (the pooling op's bracket spans three dispatches: host gaps between them must not be in it)
                # (round 6: ... and behind the sleep an UNRECORDED step, then the instrumented one: 25 ms of a spinning workgroup let
                # the core clock sag, and the brackets of the matrix-bound launches read 12 % longer than the same kernels in
                # rocprofv3's trace of back-to-back steps - 8.15 against 7.21 ms per step; the sleep is long enough for the host to
                # enqueue both steps, so no bracket holds host time, and the instrumented step starts where a served step does:
                # right behind the previous step's last kernels)
                try:
                    torch.cuda._sleep(150_000_000)
                    eager_step()
                except Exception:                          # noqa: BLE001
                    eager_step()
                ops.PROFILE_SINK = []
                eager_step()
                torch.cuda.synchronize()
                instrumented.append(ops.PROFILE_SINK)
            # What a bracket holds besides its kernel: the dispatch latency between the start event and the kernel and between the
            # kernel and the end event.  Measured here with a bracket around ONE tiny launch (B1) and around TWO (B2), medians of
            # 40 behind a sleep: B2 - B1 is a tiny kernel plus the in-stream gap, so 2 B1 - B2 is what the events add.  The kernel
            # times below are the brackets minus that (the raw sums are reported beside them); rocprofv3's durations of the same
            # kernels (profiles/r6_kernel_stats_one_stream.csv, exactly 20 steps) are the cross-check.
            tiny_k, tiny_e = K_d.reshape(-1, 3, 3)[:1].contiguous(), E_d.reshape(-1, 4, 4)[:1].contiguous()
            lib_ = model.engine().lib
            b12 = []
            for n_launch in (1, 2):
                try:
                    torch.cuda._sleep(40_000_000)
                except Exception:                          # noqa: BLE001
                    pass
                evs = []
                for _ in range(40):
                    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_.record()
                    for _j in range(n_launch):
                        lib_.camera_matrices(tiny_k, tiny_e)
                    e_.record()
                    evs.append((s_, e_))
                torch.cuda.synchronize()
                b12.append(sorted(s_.elapsed_time(e_) * 1e3 for s_, e_ in evs)[20])
            bracket_overhead_us = max(0.0, 2.0 * b12[0] - b12[1])
        torch.cuda.synchronize()
        model.sample_streams = streams_on
        ops.PROFILE_SINK = None
        ovh_ms = bracket_overhead_us * 1e-3

        class _Net:                                        # an (event, event) pair read as the bracket minus the events' own latency
            def __init__(self, s_, e_):
                self.s, self.e = s_, e_

            def elapsed_time(self, _unused=None):
                t_ = self.s.elapsed_time(self.e)
                return max(t_ - ovh_ms, 0.5 * t_)
        raw_conv_ms = [sum(s.elapsed_time(e) for k, s, e, w, _ in rr if k == 'conv_igemm') for rr in instrumented]
        raw_pool_us = sorted(sum(s.elapsed_time(e) * 1e3 for k, s, e, _, _ in rr if k == 'voxel_pool') for rr in instrumented)
        instrumented = [[(k, _Net(s, e), None, w, d) for k, s, e, w, d in rr] for rr in instrumented]
        conv_time = lambda rr: sum(s.elapsed_time(e) for k, s, e, w, _ in rr if k == 'conv_igemm')
        recs = sorted(instrumented, key=conv_time)[len(instrumented) // 2]
        pool_samples = sorted(sum(s.elapsed_time(e) * 1e3 for k, s, e, _, _ in rr if k == 'voxel_pool') for rr in instrumented)
        conv = [(s.elapsed_time(e) * 1e-3, w) for k, s, e, w, _ in recs if k == 'conv_igemm']
        # launches by the matrix-core form they ran in (a bf16 run keeps fp32 for the layers the bf16 kernel does not take).  A
        # Winograd F(2x2, 3x3) launch EXECUTES 16 / 36 of the direct form's matrix flops: the MFMA roofline below is on executed
        # flops; the direct-form-equivalent ("algorithmic") flops of those launches and the speed-up they stand for are reported
        # separately, never against the matrix peak.
        # The SPLIT Winograd form (round 6) runs those 16 / 36 on the bf16 matrix cores with every fp32 operand as three bf16 terms:
        # six bf16 MFMAs per product block - its executed flops are 6 x 16 / 36 of the direct count, priced against the bf16 peak.
        WINO = 16.0 / 36.0
        # The split TILE form ('f32 split', 'f32 split + chain' for the Bottleneck tails): six bf16 MFMAs per fp32 product.
        peak_of = lambda form: PEAK_BF16_MFMA_TFLOPS if ('bf16' in str(form) or 'split' in str(form)) else PEAK_F32_MFMA_TFLOPS

        def executed_parts(w, d):                                         # [(executed matrix flops, peak of their instruction)]
            form = str(d[-1])
            if 'winograd' in form:
                return [(w * WINO * (6.0 if 'split' in form else 1.0), peak_of(form))]
            if form.startswith('f32 split'):
                return [(6.0 * w, PEAK_BF16_MFMA_TFLOPS)]                # (a chained tail's 1x1 products run split too)
            return [(w, peak_of(form))]
        executed_of = lambda w, d: sum(x_ for x_, _ in executed_parts(w, d))
        pipe_s_of = lambda w, d: sum(x_ / (pk_ * 1e12) for x_, pk_ in executed_parts(w, d))          # seconds at the matrix peak
        pipe_by_form = {}
        by_form = {}
        for k, s_, e_, w, d in recs:
            if k == 'conv_igemm':
                t_, f_, x_, n_ = by_form.get(d[-1], (0.0, 0.0, 0.0, 0))
                by_form[d[-1]] = (t_ + s_.elapsed_time(e_) * 1e-3, f_ + w, x_ + executed_of(w, d), n_ + 1)
                pipe_by_form[d[-1]] = pipe_by_form.get(d[-1], 0.0) + pipe_s_of(w, d)
        by_prec = {}                                                   # 'f32' (all fp32 forms) / 'bf16': time, executed flops, launches
        for form, (t_, f_, x_, n_) in by_form.items():
            key = 'bf16' if form == 'bf16' else 'f32'
            a = by_prec.get(key, (0.0, 0.0, 0))
            by_prec[key] = (a[0] + t_, a[1] + x_, a[2] + n_)
        # pooling: algorithmic bytes with N_kept, counted from the voxel ranks the op left in its workspace
        pool, kept_frac = [], None
        for k, s_, e_, _, d in recs:
            if k == 'voxel_pool':
                nbytes, n_kept = ops.pool_algorithmic_bytes(d)
                pool.append((s_.elapsed_time(e_) * 1e-3, nbytes))
                kept_frac = n_kept / d['points']
        dump = os.environ.get('FIERY_BENCH_DUMP')
        if dump and rank == 0:                                           # per-launch table for kernel tuning
            rows = [dict(kind=k, us=round(s.elapsed_time(e) * 1e3, 2), work=w, detail=d if k == 'conv_igemm' else None)
                    for k, s, e, w, d in recs]
            json.dump(rows, open(dump, 'w'))
        # per-layer bound (see `bound_per_layer` below): detail = (kT, kH, kW, stride, cin, cout, images, Hout, Wout, form)
        ideal_s, mfma_bound_s, hbm_bound_s, hbm_bound_launches = 0.0, 0.0, 0.0, 0
        for k, s_, e_, w, d in recs:
            if k != 'conv_igemm':
                continue
            kT, kH, kW, stride, cin, cout, n_img, Ho, Wo, form = d
            px_out = n_img * Ho * Wo
            nbytes = 4.0 * (px_out * stride * stride * cin + px_out * cout + cin * cout * kT * kH * kW)
            t_m, t_h = pipe_s_of(w, d), nbytes / (PEAK_HBM_GBS * 1e9)
            ideal_s += max(t_m, t_h)
            mfma_bound_s += t_m
            hbm_bound_s += t_h
            hbm_bound_launches += t_h > t_m
        t_conv, f_conv = sum(t for t, _ in conv), sum(w for _, w in conv)
        layerwise = {'ideal_ms_per_step': round(ideal_s * 1e3, 3), 'frac': round(ideal_s / t_conv, 4) if t_conv else None,
                     'mfma_only_ms': round(mfma_bound_s * 1e3, 3), 'hbm_only_ms': round(hbm_bound_s * 1e3, 3),
                     'launches_hbm_bound': int(hbm_bound_launches), 'launches': len(conv),
                     'what': 'sum over launches of max(executed flops / matrix peak of the form, algorithmic bytes / 8 TB/s) '
                             'divided by the measured convolution time of the step (one stream, HIP events)'}
        x_conv = sum(x_ for _, _, x_, _ in by_form.values())             # executed matrix flops of the step's convolutions
        # (fp32-instruction equivalents: a split launch's six bf16 MFMAs stand for one fp32 product block)
        x_conv_f32eq = sum(f_ * (WINO if 'winograd' in str(k_) else 1.0) for k_, (_, f_, _, _) in by_form.items())
        # the dominant kernel = the form that holds most of the time; its flops against ITS peak
        dom = max(by_prec, key=lambda k_: by_prec[k_][0])
        t_dom, f_dom, n_dom = by_prec[dom]
        # all launches of that precision together: the time-weighted busy fraction of the matrix pipe each form runs on
        pipe_s = sum(v_ for k_, v_ in pipe_by_form.items() if (k_ == 'bf16') == (dom == 'bf16'))
        # ... and inside that precision the FORM that holds most of the time is the kernel the line names (round 5: the Winograd
        # kernel took over from the direct implicit GEMM): its own launches, executed flops and time
        KERNEL_OF = {'f32': 'k_conv_igemm (fp32 MFMA implicit GEMM, direct tile forms)', 'f32 stream-K': 'k_conv_igemm<SK> (fp32 MFMA implicit GEMM, stream-K)',
                     'f32 winograd': 'k_conv_winograd (Winograd F(2x2,3x3) on the fp32 matrix cores)',
                     'f32 split': 'k_conv_igemm<SPLIT> (implicit GEMM, fp32 operands as three bf16 terms, six products each on the bf16 matrix cores)',
                     'f32 split + chain': 'k_conv_igemm<SPLIT, CHAIN> (Bottleneck tails: the 3x3 and the chained 1x1s as three-term bf16 products)',
                     'f32 winograd split': 'k_conv_winograd, split form (Winograd F(2x2,3x3), fp32 operands as three bf16 terms, six products each on the '
                                           'bf16 matrix cores, fp32 accumulation: fp32 accuracy)',
                     'bf16': 'k_conv_igemm (bf16 operands, fp32 accumulate, MFMA implicit GEMM)'}
        dom_form = max((k_ for k_ in by_form if (k_ == 'bf16') == (dom == 'bf16')), key=lambda k_: by_form[k_][0])
        t_df, f_df, x_df, n_df = by_form[dom_form]
        peak = peak_of(dom_form)
        roofline = {'kernel': KERNEL_OF.get(str(dom_form), str(dom_form)),
                    'bound': 'mfma', 'achieved': round(pipe_by_form[dom_form] / t_df * peak, 2),
                    'peak': peak, 'unit': 'TFLOP/s', 'frac': round(pipe_by_form[dom_form] / t_df, 4),
                    'launches': n_df, 'avg_launch_us': round(t_df / n_df * 1e6, 2),
                    'share_of_conv_time': round(t_df / t_conv, 4),
                    'flops': 'EXECUTED matrix flops (a Winograd launch executes 16/36 of the direct form\'s 2*|out|*Cin*9; its split form six bf16 '
                             'products per fp32 product, priced against the bf16 peak); the direct-form count is in algorithmic_gflop / '
                             'direct_form_equivalent_tflops, never against the peak',
                    # every convolution launch of the step together (what `frac` was until round 5)
                    'all_convolutions': {'frac': round(pipe_s / t_dom, 4), 'launches': n_dom, 'avg_launch_us': round(t_dom / n_dom * 1e6, 2),
                                         'what': 'sum over launches of executed flops / the peak of the matrix instruction the form uses, over their time'},
                    'by_precision': {k_: {'launches': n_, 'ms_per_step': round(t_ * 1e3, 3), 'tflops': round(f_ / t_ / 1e12, 2)}
                                     for k_, (t_, f_, n_) in by_prec.items()},
                    # the forms the launches ran in: direct tiles ('f32'), stream-K, Winograd F(2x2, 3x3); executed = matrix flops
                    # the MFMAs really did, algorithmic = the direct form's flops for the same layers (SURVEY 8d's count)
                    'by_form': {str(k_): {'launches': n_, 'ms_per_step': round(t_ * 1e3, 3), 'executed_tflops': round(x_ / t_ / 1e12, 2),
                                          'executed_frac_of_peak': round(pipe_by_form[k_] / t_, 4), 'peak': peak_of(k_),
                                          'algorithmic_gflop': round(f_ / 1e9, 1), 'executed_gflop': round(x_ / 1e9, 1)}
                                for k_, (t_, f_, x_, n_) in by_form.items()},
                    'executed_gflop_per_step': round(x_conv / 1e9, 1),
                    'algorithmic_speedup': {'value': round(f_conv / x_conv_f32eq, 3),
                                            'what': 'direct-form flops of the step / matrix flops executed (Winograd F(2x2,3x3) layers execute 16/36); '
                                                    'a separate figure - `achieved` / `frac` are on EXECUTED flops'},
                    'direct_form_equivalent_tflops': round(f_conv / t_conv / 1e12, 2),
                    'traffic': None,             # (no counters in a timed run; the committed passes' figure follows)
                    'traffic_from_profiles': pmc_traffic(['convolutions (all forms)'], pmc_workload(args.config, args.precision, n_cam, B))
                                             or pmc_traffic(['k_conv_igemm (all tile shapes)'], pmc_workload(args.config, args.precision, n_cam, B)),
                    # per-layer roofline: a layer is bound by the matrix pipe OR by HBM - min over the two of what each allows;
                    # the step's ideal time is the sum over its launches of max(executed flops / matrix peak, algorithmic bytes /
                    # 8 TB/s) (bytes: the layer's input and output pixels x channels x 4 B + its weights, read / written once)
                    'bound_per_layer': layerwise,
                    'algorithmic_gflop_per_step': round(f_conv / 1e9, 1), 'kernel_ms_per_step': round(t_conv * 1e3, 3),
                    # (kernel_ms_per_step can exceed the line's ms_per_step: it is the sum of the launches' brackets with the whole
                    # batch on ONE stream, each kernel alone on the GPU; the timed step runs one stream per sample, three launches
                    # sharing the CUs)
                    'kernel_ms_mode': 'one stream, whole batch per launch, eager: sum of per-launch HIP-event brackets, each minus the events\' own '
                                      'latency (bracket_overhead_us, measured live: 2 x the bracket of one tiny launch - the bracket of two)',
                    'kernel_ms_per_step_raw_brackets': round(sorted(raw_conv_ms)[len(raw_conv_ms) // 2], 3),
                    'bracket_overhead_us': round(bracket_overhead_us, 2),
                    'bracket_calibration_us': {'one_tiny_launch': round(b12[0], 2), 'two_tiny_launches': round(b12[1], 2)},
                    'measured': 'HIP events around every launch of the median of five instrumented steps after the timed region, whole '
                                'batch on one stream (`--no-sample-streams` mode): the kernel alone on the GPU',
                    # the same kernel in the TIMED launch mode (hipGraph, one chain per sample: kernels of different
                    # samples share the CUs, so no per-launch bracket exists): its flops over the whole step time - a lower
                    # bound of what it reaches there, because the step also holds every other kernel
                    'timed_mode': {'achieved': round(x_conv / (elapsed / args.steps) / 1e12, 2), 'unit': 'TFLOP/s',
                                   'frac': round(x_conv / (elapsed / args.steps) / 1e12 / peak, 4),
                                   'what': 'executed conv flops of a step / timed ms_per_step (lower bound: the step holds all kernels)'},
                    'step_tflops': round(x_conv / (elapsed / args.steps) / 1e12, 2)}
        if pool:
            t_pool, b_pool = pool_samples[len(pool_samples) // 2] * 1e-6, sum(w for _, w in pool)
            gbs = b_pool / t_pool / 1e9
            pooling = {'kernel': 'k_rank_columns + k_voxel_pool_compact (op boundary projection_to_birds_eye_view)', 'bound': 'hbm',
                       'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4),
                       'traffic': None,          # (no counters in a timed run; the committed passes' figure - prepass included - follows)
                       'traffic_from_profiles': pmc_traffic(['k_voxel_pool', 'fiery::k_rank_columns'], pmc_workload(args.config, args.precision, n_cam, B)),
                       'algorithmic_mb_per_step': round(b_pool / 1e6, 1),
                       'op_us_per_step': round(t_pool * 1e6, 1), 'op_us_samples': [round(v, 1) for v in pool_samples],
                       'op_us_samples_raw_brackets': [round(v, 1) for v in raw_pool_us], 'bracket_overhead_us': round(bracket_overhead_us, 2),
                       'kept_fraction': round(kept_frac, 4),
                       'bytes': '4*C*N_kept + 12*N + 4*C*X*Y per frame (SURVEY 8d), N_kept counted from the ranks the op left'}
            ceil_ = pool_ceiling() if args.config == 'baseline.yml' and n_cam == 6 and B == 3 else None
            if ceil_:
                # the box's own ceiling for this op (committed probe): what fraction of it the op reaches
                ceil_['frac_of_ceiling'] = round(ceil_['ceiling_us'] / (t_pool * 1e6), 4)
                ceil_['frac_of_op_shaped_ceiling'] = round(ceil_['op_shaped_us'] / (t_pool * 1e6), 4)
                pooling['ceiling_from_profiles'] = ceil_

    # secondary figure: the whole `forward()` from images - image trunk and lift head on the engine as well -
    # a few eager passes after everything above, reported beside the headline, never as `value`
    from_images = None
    if rank == 0 and world == 1 and not args.no_from_images:
        image = torch.randn(B, rf + nf, n_cam, 3, *cfg.IMAGE.FINAL_DIM, device=dev)
        fwd, how = (lambda: model(image, K_d, E_d, ego_d)), 'eager launches'
        if not args.no_graph:
            try:
                model.forward_graph(image, K_d, E_d, ego_d)
                fwd, how = (lambda: model.forward_graph(image, K_d, E_d, ego_d)), 'hipGraph replay'
            except Exception:                                        # noqa: BLE001  (measure the eager path instead)
                pass
        with torch.no_grad():
            for _ in range(2):
                fwd()
            torch.cuda.synchronize()
            t_img = time.perf_counter()
            for _ in range(5):
                fwd()
            torch.cuda.synchronize()
        ms_img = (time.perf_counter() - t_img) / 5 * 1e3
        from_images = {'ms_per_step': round(ms_img, 3), 'samples_per_s': round(B / ms_img * 1e3, 2),
                       'what': f'Fiery.forward from {B * rf * n_cam} images of {cfg.IMAGE.FINAL_DIM[0]}x{cfg.IMAGE.FINAL_DIM[1]}: '
                               f'EfficientNet trunk + lift head + the hot path, {how}, mean of 5'}
        # where that pass spends its time, by entry point of the library: one eager pass, launched one by one behind a GPU-side
        # sleep (the host must not stand inside a bracket), HIP events around every launch; convolutions by form with their
        # executed flops against the fp32 matrix peak, the HBM-bound kernels by time only (their bytes are not counted here)
        try:
            from fiery_amd import native as native_
            streams_keep, model.sample_streams = model.sample_streams, False      # (one stream: a bracket holds its kernel alone)
            with torch.no_grad():
                model(image, K_d, E_d, ego_d)
                torch.cuda.synchronize()
                try:
                    torch.cuda._sleep(120_000_000)
                except Exception:                                    # noqa: BLE001
                    pass
                native_.CALL_SINK, ops.PROFILE_SINK = [], []
                model(image, K_d, E_d, ego_d)
                torch.cuda.synchronize()
            calls, convs = native_.CALL_SINK, ops.PROFILE_SINK
            native_.CALL_SINK, ops.PROFILE_SINK = None, None
            model.sample_streams = streams_keep
            table = {}
            for name, s_, e_ in calls:
                if name in ('fiery_conv_fwd', 'fiery_voxel_pool_fwd'):
                    continue                                         # (bracketed with their work below)
                t_, n_ = table.get(name, (0.0, 0))
                table[name] = (t_ + s_.elapsed_time(e_), n_ + 1)
            rows = [{'name': k_, 'launches': n_, 'ms': round(t_, 3)} for k_, (t_, n_) in table.items()]
            conv_forms = {}
            for k_, s_, e_, w_, d_ in convs:
                if k_ == 'conv_igemm':
                    key = f'fiery_conv_fwd [{d_[-1]}]'
                    x_ = w_ * (16.0 / 36.0) * (6.0 if 'split' in str(d_[-1]) else 1.0) if 'winograd' in str(d_[-1]) else w_
                    if str(d_[-1]).startswith('f32 split'):
                        x_ = 6.0 * w_                                # (chained 1x1 parts included: an upper bound for the tails)
                    t_, f_, n_ = conv_forms.get(key, (0.0, 0.0, 0))
                    conv_forms[key] = (t_ + s_.elapsed_time(e_), f_ + x_, n_ + 1)
                elif k_ == 'voxel_pool':
                    nb, _ = ops.pool_algorithmic_bytes(d_)
                    ms_ = s_.elapsed_time(e_)
                    rows.append({'name': 'fiery_voxel_pool_fwd', 'launches': 1, 'ms': round(ms_, 3), 'GB/s': round(nb / ms_ / 1e6, 1),
                                 'frac': round(nb / ms_ / 1e6 / PEAK_HBM_GBS, 4), 'bound': 'hbm'})
            for key, (t_, f_, n_) in conv_forms.items():
                pk = PEAK_BF16_MFMA_TFLOPS if ('bf16' in key or 'split' in key) else PEAK_F32_MFMA_TFLOPS
                rows.append({'name': key, 'launches': n_, 'ms': round(t_, 3), 'TFLOP/s': round(f_ / t_ / 1e9, 2),
                             'frac': round(f_ / t_ / 1e9 / pk, 4), 'bound': 'mfma (executed flops)'})
            rows.sort(key=lambda r: -r['ms'])
            total_ = sum(r['ms'] for r in rows)
            for r in rows:
                r['share'] = round(r['ms'] / total_, 4)
            from_images['kernels'] = rows
            from_images['kernels_ms_total'] = round(total_, 3)
            from_images['kernels_what'] = ('one eager pass with the whole batch on one stream, every launching entry point of libfiery_hip.so '
                                           'bracketed by HIP events (sum of brackets; the timed figure above is the captured graph with a chain per sample)')
        except Exception as e_:                                      # noqa: BLE001  (the headline line must not depend on it)
            from_images['kernels'] = {'error': repr(e_)[:200]}
            native_.CALL_SINK, ops.PROFILE_SINK = None, None
            model.sample_streams = streams_keep
        del image

    # the frames layout's exchange on its own: HIP events around the collective of a few eager steps, on every rank (each step holds
    # the collective, so all ranks walk through the same steps), and the bytes this rank receives from the others
    exchange = None
    if frames_layout:
        from fiery_amd import parallel
        parallel.EXCHANGE_SINK = []
        with torch.no_grad():
            for _ in range(5):
                eager_step()
        torch.cuda.synchronize()
        ms = sorted(s_.elapsed_time(e_) for s_, e_, _ in parallel.EXCHANGE_SINK)
        exchange = {'exchange_ms': round(ms[len(ms) // 2], 4) if ms else 0.0,
                    'bytes_received': int(parallel.EXCHANGE_SINK[0][2]) if parallel.EXCHANGE_SINK else 0,
                    'kind': args.exchange + (' (no collective issued: one rank keeps its own frames)' if not ms else '')}
        parallel.EXCHANGE_SINK = None

    # who took part: every rank's device, as the process group sees it (a SCALE record then shows N ranks on N devices)
    me = {'rank': rank, 'local_rank': local_rank, 'device': f'cuda:{torch.cuda.current_device()}',
          'name': torch.cuda.get_device_name(), 'pid': os.getpid()}
    if exchange is not None:
        me.update(exchange)
    try:
        me['pci_bus_id'] = torch.cuda.get_device_properties(torch.cuda.current_device()).pci_bus_id
    except AttributeError:
        pass
    ranks = [me]
    if use_dist:
        import torch.distributed as dist
        ranks = [None] * dist.get_world_size()
        dist.all_gather_object(ranks, me)
    if rank == 0:
        line = {
            'metric': f'BEV samples/s ({n_cam} cams x {rf} frames -> {model.bev_size[0]}x{model.bev_size[1]} BEV, hot path from '
                      'encoder outputs to output dict)',
            'value': round(B * world * args.steps / elapsed, 3), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32' if args.precision == 'f32' else 'bf16 (matrix-core operands; fp32 accumulate, activations, epilogues)',
            # (what "f32" means on the matrix cores since round 6; `roofline.by_form` says which launches ran how)
            'arithmetic': ('fp32 tensors, fp32 accumulation; matrix products on v_mfma_f32_32x32x2_f32 or - the split forms - as six bf16 partial '
                           'products of operands written as three bf16 terms each (exact), measured not less accurate than the fp32 instruction '
                           '(profiles/r6_split_bf16_probe.txt); `fp32_instruction_only`: the step without them') if args.precision == 'f32' else None,
            'data': 'synthetic',
            'config': {'workload': f'{args.config}: {n_cam} cams x {rf} past frames -> {model.bev_size[0]}x{model.bev_size[1]} BEV, '
                                   f'{nf} future frames, batch {B} per GPU, {"fp32" if args.precision == "f32" else "bf16 convolutions"}, '
                                   f'{"fused lift-splat from depth+features" if args.fused else "lifted features (n,C,D,h,w) resident in HBM"}',
                       'global_batch': B * world,
                       'parallelism': (f'frames sharded x{world} for geometry + pooling, one {"all-to-all-v (frames to their sample owners)" if args.exchange == "all_to_all" else "all-gather"} of the pooled BEV maps '
                                       f'({"RCCL" if use_dist else "local copy: 1 rank, no process group"}), then batch-sharded'
                                       if frames_layout else f'batch-sharded x{world}, no data-path collective'),
                       'launch': launch_mode + (', one stream per sample' if model.sample_streams else ''),
                       'camera_matrices': 'computed on the device every step (no calibration table)'},
            'roofline': roofline, 'roofline_pooling': pooling,
            # (the same two tables as inside `roofline`, at the top level: what ran in which form, and the timed mode's figure)
            'roofline_by_form': roofline.get('by_form') if roofline else None,
            'roofline_timed_mode': roofline.get('timed_mode') if roofline else None,
            'timed_mode': launch_mode + (', one stream per sample' if model.sample_streams else ', one stream'),
            'host_enqueue_ms_per_step': round(host_ms, 3),
            'forward_from_images': from_images,
            'ranks': {'world_size': world, 'backend': ('nccl (RCCL), world ' + str(torch.distributed.get_world_size())) if use_dist else 'none (one process)',
                      'devices': ranks},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'], want = cpu_baseline(cfg, sd, lifted, K, E, ego)
            line['speedup_vs_cpu_baseline'] = round(line['value'] / line['cpu_baseline']['value'], 1)
            # parity of this run: EVERY sample of the GPU step against the oracle's outputs for the same batch - achieved
            # max-abs error per output over the whole batch, with both bars side by side: the literal 1e-4 of the north star
            # (`within_1e-4`) and the scaled bound the parity tests assert, 1e-4 * max(1, |ref|_inf) (`within_scaled`)
            # ... over THREE input draws (one draw's segmentation error sits within 10 % of the literal bar either side of it: the
            # line prints the worst draw, not a lucky one): this run's inputs, the inputs of the GPU parity test of the same
            # configuration (tests/test_gpu_parity.py::test_hot_path_baseline_batch3_...), and a third seed pair
            with torch.no_grad():
                got = {k: (None if v is None else v.float().cpu()) for k, v in step().items()}
            draws = [{'inputs': f'this run (make_inputs seed {rank}, make_lifted_features seed {100 + rank})', 'rows': parity_rows(got, want)}]
            if not args.fused and not frames_layout and not args.single_parity_draw:
                from oracle import bev_stack
                sd_cpu = {k: v.cpu() for k, v in sd.items()}
                for s_in, s_lift in ((0, 1), (7, 8)):
                    _, K2, E2, ego2 = make_inputs(B, rf + nf, n_cam, with_image=False, seed=s_in)
                    _, _, l2 = make_lifted_features(B * rf * n_cam, C, D, (fh, fw), seed=s_lift)
                    l2 = l2.view(B, rf, n_cam, C, D, fh, fw)
                    with torch.no_grad():
                        want2 = bev_stack.bev_hot_path(sd_cpu, cfg, l2, K2, E2, ego2)
                        got2 = {k: (None if v is None else v.float().cpu())
                                for k, v in model.bev_forward(l2.to(dev), K2.to(dev), E2.to(dev), ego2.to(dev)).items()}
                    draws.append({'inputs': f'make_inputs seed {s_in}, make_lifted_features seed {s_lift}', 'rows': parity_rows(got2, want2)})
                    del l2, want2, got2
            worst = {}
            for k in draws[0]['rows']:
                worst[k] = max((d_['rows'][k] for d_ in draws), key=lambda r: r['max_abs_err'])
            line['parity'] = worst
            line['parity_draws'] = [{'inputs': d_['inputs'], 'max_abs_err': {k: r['max_abs_err'] for k, r in d_['rows'].items()},
                                     'outputs_within_1e-4': sum(1 for r in d_['rows'].values() if r['within_1e-4'])} for d_ in draws]
            line['parity_literal_1e-4'] = {'outputs_passing': sum(1 for r in worst.values() if r['within_1e-4']),
                                           'outputs': len(worst),
                                           'outputs_passing_scaled': sum(1 for r in worst.values() if r['within_scaled']),
                                           'samples_compared': int(next(v for v in want.values() if v is not None).shape[0]) * len(draws),
                                           'draws': len(draws),
                                           'what': f'all {B} samples of {len(draws)} input draws against the oracle (whole batches on the host cores), the '
                                                   'WORST draw per output; literal: max-abs <= 1e-4; scaled: max-abs <= 1e-4 * max(1, |ref|_inf)'}
            # BASELINE.md's own probe of the unmodified reference on CPU (this configuration, other host): context for `cpu_baseline`
            line['cpu_baseline']['baseline_md_reference_probe'] = {'value': 0.369, 'unit': 'samples/s', 'cores': 8, 'batch': 3,
                                                                   'what': 'BASELINE.md: the reference itself, batch 3, 8 cores of the survey container'}
        # secondary figure (BASELINE.json configs[3] / [4] are bf16 configurations): the same step with bf16 matrix-core
        # operands, timed the same way after everything above - beside the headline, never as `value`; its outputs are
        # compared with the fp32 step's of this run (the accuracy side of that mode: DESIGN.md section 4)
        if world == 1 and args.precision == 'f32' and not args.no_bf16_mode and not frames_layout:
            try:
                with torch.no_grad():
                    ref32 = {k: v.float().clone() for k, v in step().items() if v is not None}
                    model.conv_precision = 'bf16'
                    model.refresh_engine()
                    bf_step = eager_step if (args.no_graph or graph_step is None) else graph_step
                    for _ in range(max(args.warmup, 2)):
                        got16 = bf_step()
                    torch.cuda.synchronize()
                    t16 = time.perf_counter()
                    for _ in range(args.steps):
                        got16 = bf_step()
                    torch.cuda.synchronize()
                    ms16 = (time.perf_counter() - t16) / args.steps * 1e3
                    line['bf16_mode'] = {
                        'value': round(B / ms16 * 1e3, 3), 'unit': 'samples/s', 'ms_per_step': round(ms16, 3),
                        'dtype': 'bf16 matrix-core operands; fp32 accumulation, epilogues and tensors in HBM',
                        'max_abs_diff_vs_fp32_step': {k: float(f'{(got16[k].float() - v).abs().max().item():.3e}') for k, v in ref32.items()},
                        # the step's convolution flops over the whole timed step (a lower bound on the kernels' own rate),
                        # against the dense bf16 matrix peak of the guide (2.5 PFLOP/s)
                        'step_tflops': (round(roofline['algorithmic_gflop_per_step'] / ms16, 1)
                                        if roofline and roofline.get('algorithmic_gflop_per_step') else None),
                        'frac_of_bf16_peak': (round(roofline['algorithmic_gflop_per_step'] / ms16 / 2500.0, 4)
                                              if roofline and roofline.get('algorithmic_gflop_per_step') else None),
                        'what': f'the same workload and launch mode, {args.steps} steps after {max(args.warmup, 2)} warm-up'}
            except Exception as e:                                   # noqa: BLE001  (the headline line must not depend on it)
                line['bf16_mode'] = {'error': repr(e)[:200]}
            finally:
                model.conv_precision = 'f32'
        # secondary lines (BASELINE.json configs[3] and configs[4] at one GPU): this script again, in a process of its own,
        # for pon_setting.yml (400 x 200 BEV, the pooling stress) and lyft/baseline.yml with 7 cameras, both with bf16
        # matrix-core operands - each with its own `roofline` / `roofline_pooling`; beside the headline, never as `value`
        if (world == 1 and not args.no_secondary_configs and args.precision == 'f32' and args.config == 'baseline.yml'
                and not frames_layout and not args.fused and not args.cams):
            import subprocess
            line['secondary_configs'] = {}
            for key, extra in (('pon_setting_bf16', ['--config', 'literature/pon_setting.yml', '--precision', 'bf16']),
                               ('lyft_baseline_7cams_bf16', ['--config', 'lyft/baseline.yml', '--cams', '7', '--precision', 'bf16'])):
                cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(args.warmup), '--batch', str(B),
                       '--no-cpu-baseline', '--no-from-images', '--no-bf16-mode', '--no-secondary-configs'] + extra
                try:
                    # (without FIERY_BENCH_DUMP: the per-launch table belongs to the headline step, not to the last secondary run)
                    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                                         env={k: v for k, v in os.environ.items() if k != 'FIERY_BENCH_DUMP'})
                    sub = json.loads(res.stdout.strip().splitlines()[-1])
                    line['secondary_configs'][key] = {k: sub.get(k) for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'config', 'roofline',
                                                                             'roofline_pooling', 'host_enqueue_ms_per_step')}
                except Exception as e:                               # noqa: BLE001  (the headline line must not depend on it)
                    line['secondary_configs'][key] = {'error': repr(e)[:200]}
            # ... and the headline workload once more with the split forms taken out of the candidates: every matrix product on the
            # fp32 instruction (v_mfma_f32_32x32x2_f32) - what the step ran as until round 6, beside the headline, never as `value`
            cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(args.warmup), '--batch', str(B),
                   '--no-cpu-baseline', '--no-from-images', '--no-bf16-mode', '--no-secondary-configs']
            try:
                env = {k: v for k, v in os.environ.items() if k != 'FIERY_BENCH_DUMP'}
                env.update(FIERY_CONV_WINOGRAD_SPLIT='0', FIERY_CONV_SPLIT='0', FIERY_TEMPORAL_PAD32='0')      # (its own best layout: round 5's)
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
                sub = json.loads([l_ for l_ in res.stdout.strip().splitlines() if l_.startswith('{')][-1])
                line['fp32_instruction_only'] = {'value': sub.get('value'), 'unit': sub.get('unit'), 'ms_per_step': sub.get('ms_per_step'),
                                                 'roofline': {k: (sub.get('roofline') or {}).get(k) for k in ('kernel', 'achieved', 'peak', 'frac', 'kernel_ms_per_step')},
                                                 'what': 'the same workload in a process of its own with FIERY_CONV_WINOGRAD_SPLIT=0 FIERY_CONV_SPLIT=0 FIERY_TEMPORAL_PAD32=0 (round 5\'s forms and layout)'}
            except Exception as e:                                   # noqa: BLE001
                line['fp32_instruction_only'] = {'error': repr(e)[:200]}
        print(json.dumps(line), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()                      # rank 0 ran its instrumented step meanwhile: leave together
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
