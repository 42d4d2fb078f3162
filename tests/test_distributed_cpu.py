"""world_size-2 (and 3) gloo runs of the sharding bookkeeping in fiery_amd/parallel.py on the CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fiery_amd.parallel import ShardedBevPath, block_range, gather_blocks, owner_of


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _frame_value(f):
    """A recognisable fake 'pooled BEV map' for global frame f."""
    return torch.full((2, 3, 3), float(f)) + torch.arange(18, dtype=torch.float32).view(2, 3, 3) / 100.0


def _worker(rank, world, port, batch, S, results, layout='auto', exchange='all_gather'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        calls = []

        def pool_frames(lo, hi, out=None):
            calls.append((lo, hi))
            return torch.stack([_frame_value(f) for f in range(lo, hi)]) if hi > lo else torch.zeros(0, 2, 3, 3)

        def stack(bev, blo, bhi):
            return {'sum': bev.view(bhi - blo, S, -1).sum(dim=(1, 2)), 'range': (blo, bhi), 'bev': bev.clone()}

        sharder = ShardedBevPath(layout=layout, exchange=exchange)
        out = sharder.run(batch, S, pool_frames, stack)
        out_again = sharder.run(batch, S, pool_frames, stack)              # the exchange buffers are reused
        assert (out is None) == (out_again is None) and (out is None or torch.equal(out['bev'], out_again['bev']))
        del calls[len(calls) // 2:]
        # gather_blocks round trip with an uneven split
        n = 5
        lo, hi = block_range(n, world, rank)
        full = gather_blocks(torch.arange(lo, hi, dtype=torch.float32).view(-1, 1), n)
        moved = [sc.bytes_received for key, sc in sharder._exchange.items() if key[0] == 'a2a']
        results[rank] = dict(layout=sharder.layout(batch), calls=calls, out=out, full=full, n_exchange=len(sharder._exchange), moved=moved)
    finally:
        dist.destroy_process_group()


def _run(world, batch, S, layout='auto', exchange='all_gather'):
    port = _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_worker, args=(world, port, batch, S, results, layout, exchange), nprocs=world, join=True)
        return dict(results)


def test_block_partition_covers_everything_once():
    for n in (1, 3, 9, 24, 72):
        for world in (1, 2, 4, 8):
            owned = [i for r in range(world) for i in range(*block_range(n, world, r))]
            assert owned == list(range(n))
            assert all(lo <= i < hi for i in range(n) for lo, hi in [block_range(n, world, owner_of(i, n, world))])


def test_batch_sharding_needs_no_collective():
    res = _run(world=2, batch=4, S=3)
    assert res[0]['layout'] == res[1]['layout'] == 'batch'
    assert res[0]['calls'] == [(0, 6)] and res[1]['calls'] == [(6, 12)]          # frames of the owned samples only
    assert res[0]['out']['range'] == (0, 2) and res[1]['out']['range'] == (2, 4)
    want = torch.stack([sum(_frame_value(b * 3 + t).sum() for t in range(3)) for b in range(4)])
    got = torch.cat([res[0]['out']['sum'], res[1]['out']['sum']])
    assert torch.allclose(got, want)
    assert torch.equal(res[0]['full'].view(-1), torch.arange(5.0)) and torch.equal(res[1]['full'], res[0]['full'])


def test_frame_sharding_all_gathers_the_bev_maps():
    """batch 1 on 2 ranks (and on 3): frames split for lift-splat, one all-gather, full stack per rank."""
    for world in (2, 3):
        res = _run(world=world, batch=1, S=3)
        pooled = sorted(f for r in range(world) for lo, hi in res[r]['calls'] for f in range(lo, hi))
        assert pooled == [0, 1, 2]                                               # every frame pooled exactly once
        want = torch.stack([_frame_value(f) for f in range(3)])
        for r in range(world):
            assert res[r]['layout'] == 'frames'
            assert torch.equal(res[r]['out']['bev'], want)                       # every rank sees all frames, in order
            assert res[r]['out']['range'] == (0, 1)


@pytest.mark.parametrize('world,batch', [(2, 1), (3, 2), (3, 4), (2, 5), (3, 3)])
def test_frame_sharding_with_the_all_to_all_exchange_moves_each_frame_to_its_owner_only(world, batch):
    """`FrameScatter`: frames split over the ranks for lift-splat, one all-to-all-v, every sample's S frames arrive - in
    order - at the one rank that owns the sample and nowhere else; surplus ranks (batch < world) pool and own nothing."""
    from fiery_amd.parallel import sample_owner_ranges
    S = 3
    res = _run(world=world, batch=batch, S=S, layout='frames', exchange='all_to_all')
    pooled = sorted(f for r in range(world) for lo, hi in res[r]['calls'] for f in range(lo, hi))
    assert pooled == list(range(batch * S))                                      # every frame pooled exactly once
    owners = sample_owner_ranges(batch, world)
    covered = []
    for r in range(world):
        blo, bhi = owners[r]
        if bhi == blo:
            assert res[r]['out'] is None
            continue
        assert res[r]['out']['range'] == (blo, bhi)
        want = torch.stack([_frame_value(f) for f in range(blo * S, bhi * S)])
        assert torch.equal(res[r]['out']['bev'], want)                           # its samples' frames, in order, nothing else
        covered.extend(range(blo, bhi))
        assert res[r]['n_exchange'] == 1                                         # allocated once, reused by the second call
        own = set(range(*block_range(batch * S, world, r)))
        n_foreign = len([f for f in range(blo * S, bhi * S) if f not in own])
        assert res[r]['moved'] == [n_foreign * 18 * 4]                           # bytes that crossed ranks: foreign frames only
    assert covered == list(range(batch))


@pytest.mark.parametrize('exchange', ['all_gather', 'all_to_all'])
def test_baseline_config_2_bookkeeping_at_eight_ranks(exchange):
    """BASELINE.json configs[2]: baseline.yml, global batch 24 over 8 ranks, frames sharded for lift-splat (72 frames, 9 per
    rank), one exchange, then 3 samples per rank.  With block partitions of both frames and samples a rank pools exactly its
    own samples' frames: the all-to-all-v moves nothing across ranks, the all-gather hands every rank all 72 maps."""
    from fiery_amd.parallel import sample_owner_ranges
    world, batch, S = 8, 24, 3
    res = _run(world=world, batch=batch, S=S, layout='frames', exchange=exchange)
    assert sorted(f for r in range(world) for lo, hi in res[r]['calls'] for f in range(lo, hi)) == list(range(72))
    assert sample_owner_ranges(batch, world) == [(3 * r, 3 * r + 3) for r in range(world)]
    for r in range(world):
        assert res[r]['layout'] == 'frames'
        assert res[r]['calls'] == [(9 * r, 9 * r + 9)]                          # one pooling call: this rank's nine frames
        assert res[r]['out']['range'] == (3 * r, 3 * r + 3)
        want = torch.stack([_frame_value(f) for f in range(9 * r, 9 * r + 9)])
        assert torch.equal(res[r]['out']['bev'], want)
        if exchange == 'all_to_all':
            assert res[r]['moved'] == [0]                                        # every frame already sits with its sample's owner
        assert torch.equal(res[r]['full'], torch.arange(5, dtype=torch.float32).view(-1, 1))


# ---- the real entry point (`sharded_bev_forward`) on the CPU-simulated kernels, 2 ranks over gloo -------------------
def _sim_worker(rank, world, port, batch, layout, results, exchange='all_gather'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fiery_amd import native
        from fiery_amd.model import Fiery
        from fiery_amd.parallel import sharded_bev_forward
        from fiery_amd.synthetic import make_inputs, make_lifted_features
        from tests.helpers import randomise_weights, tiny_cfg
        from tests.sim.build_sim import build
        torch.set_num_threads(1)
        cfg = tiny_cfg('baseline.yml', bev=8, **{'MODEL.FUTURE_PRED.N_GRU_BLOCKS': 1, 'MODEL.FUTURE_PRED.N_RES_LAYERS': 1,
                                                  'N_FUTURE_FRAMES': 1})
        torch.manual_seed(0)
        model = Fiery(cfg).eval()
        randomise_weights(model)
        model._lib = native.Lib(build())
        rf, n = model.receptive_field, 2
        _, K, E, ego = make_inputs(batch, rf + model.n_future, n, with_image=False, seed=7)
        fh, fw = cfg.IMAGE.FINAL_DIM[0] // 8, cfg.IMAGE.FINAL_DIM[1] // 8
        _, _, lifted = make_lifted_features(batch * rf * n, 64, model.depth_channels, (fh, fw), seed=8)
        lifted = lifted.view(batch, rf, n, 64, model.depth_channels, fh, fw)
        noise = torch.randn(batch, 1, 32, generator=torch.Generator().manual_seed(9))
        with torch.no_grad():
            out, rng = sharded_bev_forward(model, K, E, ego, lifted=lifted, noise=noise, layout=layout, exchange=exchange)
            out2, _ = sharded_bev_forward(model, K, E, ego, lifted=lifted, noise=noise, layout=layout, exchange=exchange)   # buffers reused
            whole = model.bev_forward(lifted, K, E, ego, None, noise)
        out, out2 = out or {}, out2 or {}                     # (a rank that owns no sample gets None)
        results[rank] = dict(range=rng or (0, 0), out={k: v.clone() for k, v in out.items() if v is not None},
                             again={k: v.clone() for k, v in out2.items() if v is not None},
                             whole={k: v.clone() for k, v in whole.items() if v is not None},
                             n_exchange=len(model._sharder._exchange))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('batch,layout,exchange', [(2, 'batch', 'all_gather'), (1, 'auto', 'all_gather'), (3, 'frames', 'all_gather'),
                                                   (3, 'frames', 'all_to_all'), (1, 'frames', 'all_to_all')])
def test_sharded_bev_forward_on_the_simulated_kernels(batch, layout, exchange):
    """Every rank's share of `sharded_bev_forward` equals the same samples of the single-process pass - batch layout (no
    collective), frame layout (one all-gather through the preallocated exchange) and the small-batch latency mode."""
    from tests.sim.build_sim import build
    build()                                                  # compile once, before the ranks start
    world = 2
    port = _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_sim_worker, args=(world, port, batch, layout, results, exchange), nprocs=world, join=True)
        res = dict(results)
    covered = set()
    for r in range(world):
        lo, hi = res[r]['range']
        covered.update(range(lo, hi))
        for k, v in res[r]['out'].items():
            want = res[r]['whole'][k][lo:hi]
            assert v.shape == want.shape, k
            assert (v - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()), (k, r)
            assert torch.equal(res[r]['again'][k], v), k
        assert res[r]['n_exchange'] == (0 if (layout == 'batch') else 1)        # allocated once, reused by the second call
    assert covered == set(range(batch))


# ---- SyncBatchNorm on the kernels: statistics over all processes ------------------------------------------------------
def _syncbn_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fiery_amd import native
        from fiery_amd.modules import ResidualBottleneck
        from fiery_amd.train_graph import TrainGraph
        from tests.helpers import randomise_weights
        from tests.sim.build_sim import build
        torch.set_num_threads(1)
        lib = native.Lib(build())
        g = torch.Generator().manual_seed(11)
        x_all = torch.randn(4, 64, 10, 12, generator=g)
        gy_all = torch.randn(4, 32, 5, 6, generator=g)
        torch.manual_seed(0)
        block = ResidualBottleneck(64, 32, downsample=True)
        randomise_weights(block)
        whole = ResidualBottleneck(64, 32, downsample=True)
        whole.load_state_dict(block.state_dict())
        block = torch.nn.SyncBatchNorm.convert_sync_batchnorm(block).train()
        whole.train()
        graph = TrainGraph(None, lib)
        # this rank's half of the batch through the SyncBatchNorm block ...
        lo, hi = rank * 2, rank * 2 + 2
        x = x_all[lo:hi].clone().requires_grad_()
        y = graph.bottleneck(x, block)
        y.backward(gy_all[lo:hi])
        # ... against the whole batch through the plain-BatchNorm block in one process
        xw = x_all.clone().requires_grad_()
        yw = graph.bottleneck(xw, whole)
        yw.backward(gy_all)
        results[rank] = dict(
            y=(y.detach() - yw.detach()[lo:hi]).abs().max().item(), dx=(x.grad - xw.grad[lo:hi]).abs().max().item(),
            stats=max((b - dict(whole.named_buffers())[n]).abs().max().item() for n, b in block.named_buffers() if 'running' in n),
            grads={n: p.grad.clone() for n, p in block.named_parameters()}, whole={n: p.grad.clone() for n, p in whole.named_parameters()},
            scale=yw.detach().abs().max().item(), converted=sum(isinstance(m, torch.nn.SyncBatchNorm) for m in block.modules()))
    finally:
        dist.destroy_process_group()


def test_sync_batchnorm_on_the_kernels_equals_the_whole_batch_in_one_process():
    """A down-sampling Bottleneck whose BatchNorms were converted by `nn.SyncBatchNorm.convert_sync_batchnorm` (what
    `sync_batchnorm=True` of train.py:37 does), two processes with half the batch each: outputs, input gradients and running
    statistics equal the whole batch in one process; the parameter gradients are the local sums (they add up to the whole
    batch's - DistributedDataParallel averages them afterwards)."""
    from tests.sim.build_sim import build
    build()
    world, port = 2, _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_syncbn_worker, args=(world, port, results), nprocs=world, join=True)
        res = dict(results)
    for r in range(world):
        assert res[r]['converted'] == 4
        assert res[r]['y'] <= 1e-5 * max(1.0, res[r]['scale']) and res[r]['dx'] <= 1e-4 and res[r]['stats'] <= 1e-5
    for name, want in res[0]['whole'].items():
        total = res[0]['grads'][name] + res[1]['grads'][name]
        assert torch.allclose(total, want, rtol=1e-4, atol=1e-4 * max(1.0, want.abs().max().item())), name


# ---- train.py:35 end to end: DistributedDataParallel around the training graph -------------------------------------------
def _ddp_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fiery_amd import native
        from fiery_amd.model import Fiery
        from tests.helpers import forward_case, randomise_weights, tiny_cfg
        from tests.sim.build_sim import build
        torch.set_num_threads(1)
        lib = native.Lib(build())
        cfg = tiny_cfg('baseline.yml', bev=8, **{'MODEL.FUTURE_PRED.N_GRU_BLOCKS': 1, 'MODEL.FUTURE_PRED.N_RES_LAYERS': 1, 'N_FUTURE_FRAMES': 1,
                                                  'TIME_RECEPTIVE_FIELD': 2})
        torch.manual_seed(0)
        model = Fiery(cfg)
        state = {k: v.clone() for k, v in randomise_weights(model).items()}
        lifted, K, E, ego, labels, noise = forward_case(cfg, model.receptive_field, model.n_future, model.depth_channels, model.bev_size, 4, 2,
                                                        with_labels=True, with_noise=True)

        def loss_of(out):
            g = torch.Generator().manual_seed(5)
            return sum((v * torch.randn(v.shape[1:], generator=g)).sum() for k, v in sorted(out.items()) if v is not None)

        def step(net, sl):
            loss_of(net(lifted[sl].clone(), K[sl], E[sl], ego[sl], labels[sl], noise[sl])).backward()

        # what train.py:35 sets up (DDP's hooks fire on the module's forward: route the hot-path entry through it); this process
        # takes samples 2 rank, 2 rank + 1
        model._lib = lib
        model.train()
        model.forward = lambda *a, **kw: model.bev_forward(*a, **kw)
        wrapped = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)
        step(wrapped, slice(2 * rank, 2 * rank + 2))
        got = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None and not n.startswith('encoder.')}
        # this process's share alone, without DDP: the test averages the two processes' by hand
        single = Fiery(cfg)
        single.load_state_dict(state)
        single.train()
        single._lib = lib
        step(lambda *a: single.bev_forward(*a), slice(2 * rank, 2 * rank + 2))
        own = {n: p.grad.clone() for n, p in single.named_parameters() if p.grad is not None and not n.startswith('encoder.')}
        results[rank] = dict(got=got, own=own)
    finally:
        dist.destroy_process_group()


def test_distributed_data_parallel_around_the_training_graph():
    """What train.py:35 sets up - DistributedDataParallel, two processes with two samples each - around the training graph
    (simulated kernels, gloo): its reducer sees the gradients the HIP operators' autograd Functions produce and leaves every
    process with the average over the processes.  (SyncBatchNorm is covered above at block level: torch refuses to wrap
    SyncBatchNorm modules on the host.)"""
    from tests.sim.build_sim import build
    build()
    world, port = 2, _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_ddp_worker, args=(world, port, results), nprocs=world, join=True)
        res = dict(results)
    want = {n: sum(res[r]['own'][n] for r in range(world)) / world for n in res[0]['own']}
    for r in range(world):
        got = res[r]['got']
        assert set(got) == set(want) and len(got) > 50
        top = max(v.abs().max().item() for v in want.values())
        for name, g in want.items():
            assert (got[name] - g).abs().max().item() <= 1e-4 * g.abs().max().item() + 1e-5 * top, name


def _forms_worker(rank, world, port, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from fiery_amd import ops
        from fiery_amd.parallel import share_conv_forms
        sig = (1, 3, 3, 1, 128, 128, (16, 0), 1, 0, 0, 0, 0, 0, 0)
        # every rank "measured" something else for the same launch, rank 1 also a shape of its own
        ops.load_form_table({(sig, (3, 200, 200)): ['wino', 128, 'sk'][rank % 3], (sig, (rank, 50, 50)): 64}, frozen=False)
        table = share_conv_forms()
        results[rank] = (table, ops.form_table(), ops.FORM_TABLE_FROZEN)
    finally:
        dist.destroy_process_group()


def test_ranks_take_rank_zeros_convolution_forms(tmp_path):
    """`share_conv_forms`: the forms differ in fp32 rounding, so every rank runs rank 0's measured choices (frozen: nothing is
    timed afterwards); and the table survives a round trip through its JSON file (FIERY_CONV_FORM_TABLE)."""
    from fiery_amd import ops
    world, port = 3, _free_port()
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(_forms_worker, args=(world, port, results), nprocs=world, join=True)
        results = dict(results)
    sig = (1, 3, 3, 1, 128, 128, (16, 0), 1, 0, 0, 0, 0, 0, 0)
    want = {(sig, (3, 200, 200)): 'wino', (sig, (0, 50, 50)): 64}
    for rank in range(world):
        table, installed, frozen = results[rank]
        assert table == want and installed == want and frozen
    before, before_frozen = ops.form_table(), ops.FORM_TABLE_FROZEN
    try:
        ops.load_form_table(want, frozen=False)
        path = str(tmp_path / 'forms.json')
        ops.save_form_table(path)
        assert ops.read_form_table(path) == want
    finally:
        ops.load_form_table(before, frozen=before_frozen)
