"""Time one training step (forward + backward of the BEV path from the lifted features) on the GPU, with the per-operator
split: python tools/time_train_step.py [--preset baseline.yml] [--batch 2] [--steps 5] [--torch-conv]

--torch-conv substitutes PyTorch-ROCm's operators (MIOpen convolution, ATen BatchNorm / interpolate) for the HIP ones in the
same graph; --reference-ops additionally takes the pyramid pooling operator for operator as the reference does (avg_pool3d +
interpolate) - the step the reference's own modules run on this GPU.
Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--preset', default='baseline.yml')
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--cams', type=int, default=6)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--torch-conv', action='store_true')
    ap.add_argument('--reference-ops', action='store_true',
                    help="the reference's own operator sequence on PyTorch-ROCm: MIOpen convolutions, avg_pool3d + interpolate for the "
                         'pyramid pooling, F.interpolate for the decoder (implies --torch-conv)')
    ap.add_argument('--profile', action='store_true', help='torch profiler table of one step')
    ap.add_argument('--ops', action='store_true', help='with --profile: also the operator table grouped by input shape')
    ap.add_argument('--from-images', action='store_true', help='the step from camera images (image trunk + lift head under autograd too)')
    ap.add_argument('--torch-trunk', action='store_true', help='with --from-images: trunk and lift head on PyTorch-ROCm operators (MIOpen)')
    ap.add_argument('--sites', action='store_true', help='ATen kernels of one step by fiery_amd call site (self device time)')
    args = ap.parse_args()
    from fiery_amd.config import get_preset_cfg
    from fiery_amd.model import Fiery
    from fiery_amd.synthetic import make_inputs, make_lifted_features, randomise_weights
    from fiery_amd.train_graph import TrainGraph
    cfg = get_preset_cfg(args.preset)
    torch.manual_seed(0)
    model = Fiery(cfg)
    randomise_weights(model)
    model = model.cuda().train()
    B, n, rf = args.batch, args.cams, model.receptive_field
    _, K, E, ego = [None if t is None else t.cuda() for t in make_inputs(B, rf + model.n_future, n, with_image=False)]
    fh, fw = cfg.IMAGE.FINAL_DIM[0] // cfg.MODEL.ENCODER.DOWNSAMPLE, cfg.IMAGE.FINAL_DIM[1] // cfg.MODEL.ENCODER.DOWNSAMPLE
    _, _, lifted = make_lifted_features(B * rf * n, model.encoder_out_channels, model.depth_channels, (fh, fw), seed=1)
    lifted = lifted.view(B, rf, n, model.encoder_out_channels, model.depth_channels, fh, fw).cuda().requires_grad_()
    labels = torch.randn(B, 1 + model.n_future, 6, *model.bev_size, device='cuda') if model.n_future > 0 else None
    conv = (lambda x, w, s, p, lib: F.conv2d(x, w, None, s, p)) if (args.torch_conv or args.reference_ops) else None
    graph = TrainGraph(model, conv2d=conv)
    graph.whole_plane_pooling_as_means = not args.reference_ops
    params = [p for name, p in model.named_parameters() if args.from_images or not name.startswith('encoder.')]
    if args.from_images:
        graph.hip_trunk = not args.torch_trunk
        image = torch.randn(B, rf + model.n_future, n, 3, *cfg.IMAGE.FINAL_DIM, device='cuda')
    opt = torch.optim.SGD(params, lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        lifted.grad = None
        out = graph.forward(image, K, E, ego, labels) if args.from_images else graph.bev_forward(lifted, K, E, ego, labels)
        loss = sum((v ** 2).mean() for v in out.values() if v is not None)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    line = dict(tool='time_train_step', preset=args.preset, batch=B, cams=n, start=('images, trunk on ' + ('PyTorch-ROCm' if args.torch_trunk else 'the HIP kernels')) if args.from_images else 'lifted features', conv='reference operator sequence (MIOpen / ATen)' if args.reference_ops else 'torch(MIOpen)' if args.torch_conv else 'hip',
                ms_per_step=round(ms, 2), samples_per_s=round(B / ms * 1e3, 2), peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2**30, 2))
    if args.profile:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25), file=sys.stderr)
        if args.ops:
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
                step()
                torch.cuda.synchronize()
            print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=40,
                                                                     max_shapes_column_width=70), file=sys.stderr)
    if args.sites:
        # which lines of fiery_amd / the autograd engine put ATen copy / element-wise kernels on the GPU, by self device time
        from collections import defaultdict
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        sites = defaultdict(lambda: [0.0, 0])
        for ev in prof.events():
            if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
                continue
            dev_us = getattr(ev, 'self_device_time_total', None) or getattr(ev, 'self_cuda_time_total', 0)
            if dev_us <= 0:
                continue
            frames = [f for f in (ev.stack or []) if 'fiery_amd' in f]
            shapes = str(ev.input_shapes)[:60] if ev.input_shapes else ''
            node, up = '', ev.cpu_parent
            while up is not None:                 # backward: the autograd node the engine was evaluating (or accumulating after)
                if up.name.startswith('autograd::engine::evaluate_function: '):
                    node = up.name.split(': ', 1)[1]
                    break
                up = up.cpu_parent
            key = (ev.name, frames[0].split('fiery_amd/')[-1][:60] if frames else f'(backward of {node})' if node else '(other)', shapes)
            sites[key][0] += dev_us
            sites[key][1] += 1
        total = sum(v[0] for v in sites.values())
        print(f'ATen kernels of one step by call site: {total / 1e3:.2f} ms of device time', file=sys.stderr)
        for (name, site, shapes), (us, n_) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("FIERY_SITES", "45"))]:
            print(f'  {us / 1e3:7.3f} ms {n_:4d}x  {name:28s} {site:62s} {shapes}', file=sys.stderr)
    print(json.dumps(line))


if __name__ == '__main__':
    main()
