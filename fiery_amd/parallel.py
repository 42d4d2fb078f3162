"""Multi-GPU execution of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The path shards naturally: every (batch, frame) lift-splat is independent, and
everything after pooling is independent per batch element.  Two layouts:

* 'batch' (default whenever the batch covers the ranks): rank r owns a contiguous block of samples end to
  end; no data-path collective at all - what `bench.py --gpus N` measures (weak scaling).
* 'frames' (default for batches smaller than the world, e.g. single-sample latency on 8 GPUs; selectable for
  any batch - BASELINE.json configs[2], `bench.py --layout frames`): the B*S frames are split across ranks
  for geometry + lift-splat, ONE all-gather moves the pooled BEV maps ((C, X, Y) fp32 = 10.24 MB per frame at
  baseline.yml) so that every rank holds all frames, then each rank runs the temporal / future-prediction /
  decoder stack for the samples it owns.  On the 8-GPU fully connected xGMI mesh an all-gather drives all
  seven links of a GPU at once, so the exchange is bandwidth-bound per link on 1/7 of the data rather than a
  7-step ring.  The exchange buffers are allocated once (`FrameExchange`): the pooling kernel writes its
  frames straight into the send buffer, and the receive buffer is reused every step.
"""
import torch
import torch.distributed as dist


# When a list is installed here (bench.py does, for its instrumented steps), every exchange appends (start_event, end_event,
# bytes_received) recorded on the stream the collective is enqueued on.
EXCHANGE_SINK = None


def _timed_exchange(fn, bytes_received, device):
    if EXCHANGE_SINK is None or torch.device(device).type != 'cuda':
        return fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    out = fn()
    end.record()
    EXCHANGE_SINK.append((start, end, bytes_received))
    return out


def share_conv_forms(group=None, src=0, freeze=True):
    """Every rank takes rank `src`'s table of measured convolution forms (`ops.FORM_TABLE`): the forms differ in fp32
    rounding, and ranks that timed their own candidates may have picked differently - after this call every rank runs the
    same arithmetic (and, with `freeze`, times nothing more: shapes the table does not hold take the library's heuristic
    form).  Call after the warm-up pass that tuned the launches and before anything is captured or compared."""
    from . import ops
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if freeze:
            ops.load_form_table(ops.form_table(), frozen=True)
        return ops.form_table()
    box = [ops.form_table() if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    ops.load_form_table(box[0], frozen=freeze)
    return box[0]


def block_range(n_items, world, rank):
    """Balanced contiguous partition: [lo, hi) of `n_items` owned by `rank`; every rank gets floor or ceil of
    n_items / world items (9 samples on 8 ranks: one rank takes two, nobody idles while another holds a double share)."""
    return rank * n_items // world, (rank + 1) * n_items // world


def owner_of(index, n_items, world):
    """The rank whose `block_range` holds `index`."""
    return ((index + 1) * world - 1) // n_items


class FrameExchange:
    """Preallocated buffers of the one all-gather of the 'frames' layout.

    `send` holds this rank's block of items padded to the largest block (`per` rows, the collective needs equal
    contributions); `recv` holds world * per rows.  Global item i sits in row owner(i) * per + (i - lo(owner(i)))."""

    def __init__(self, n_items, item_shape, device, dtype=torch.float32, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_items = n_items
        self.per = max(hi - lo for lo, hi in (block_range(n_items, self.world, r) for r in range(self.world)))
        self.lo, self.hi = block_range(n_items, self.world, self.rank)
        self.send = torch.zeros((max(self.per, 1),) + tuple(item_shape), dtype=dtype, device=device)
        self.recv = torch.empty((self.world * max(self.per, 1),) + tuple(item_shape), dtype=dtype, device=device)
        rows = []
        for i in range(n_items):
            r = owner_of(i, n_items, self.world)
            rows.append(r * self.per + (i - block_range(n_items, self.world, r)[0]))
        self.row_of = rows
        self._index = {}
        row_bytes = self.send[0].numel() * self.send.element_size()
        self.bytes_received = (n_items - (self.hi - self.lo)) * row_bytes      # rows of the other ranks (padding not counted)

    def local_out(self):
        """Where this rank's items go (rows [0, hi - lo) of the send buffer): hand it to the producer as `out=`."""
        return self.send[:self.hi - self.lo]

    def gather(self):
        if self.world == 1 and not dist.is_initialized():
            self.recv.copy_(self.send)
        else:
            _timed_exchange(lambda: dist.all_gather_into_tensor(self.recv, self.send, group=self.group), self.bytes_received, self.recv.device)
        return self.recv

    def rows(self, lo, hi):
        """Items [lo, hi) of the gathered set, in order: a view when their rows are consecutive, else one gather."""
        rows = self.row_of[lo:hi]
        if all(b == a + 1 for a, b in zip(rows[:-1], rows[1:])):
            return self.recv[rows[0]:rows[0] + len(rows)]
        idx = self._index.get((lo, hi))
        if idx is None:
            idx = self._index[(lo, hi)] = torch.tensor(rows, dtype=torch.long, device=self.recv.device)
        return self.recv.index_select(0, idx)


def sample_owner_ranges(batch, world):
    """[lo, hi) of samples every rank runs the post-pooling stack for in the 'frames' layout with the all-to-all exchange:
    balanced blocks when the batch covers the ranks, else sample r on rank r and nothing on the surplus ranks (they only
    pool frames; with the all-gather exchange they replicate a sample instead)."""
    if batch >= world:
        return [block_range(batch, world, r) for r in range(world)]
    return [(r, r + 1) if r < batch else (batch, batch) for r in range(world)]


class FrameScatter:
    """The exchange of the 'frames' layout that moves only what is needed (the all-to-all-v
    variant): rank r pools frames block_range(n_frames)[r] and sends each to the ONE rank that runs its sample's stack;
    a rank receives exactly the S frames of each sample it owns.  `dist.all_to_all_single` with split sizes: the frames a
    rank pools are consecutive and so are the frames a rank needs, so both buffers are contiguous in global frame order -
    no index gather, no padding to equal contributions.  Bytes per rank at B = 24, world = 8 (72 frames of 10.24 MB): the
    all-gather receives 645 MB and sends 92 MB per rank; this exchange moves 0 (every rank pools its own samples' frames);
    at B = 4, world = 8 (12 frames): all-gather 123 MB received per rank, here 30.7 MB into each of the four owners.
    Buffers are allocated once and reused."""

    def __init__(self, batch, frames_per_sample, item_shape, device, dtype=torch.float32, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        S = frames_per_sample
        n_frames = batch * S
        self.lo, self.hi = block_range(n_frames, self.world, self.rank)
        owners = sample_owner_ranges(batch, self.world)
        self.blo, self.bhi = owners[self.rank]
        need = [(lo * S, hi * S) for lo, hi in owners]
        have = [block_range(n_frames, self.world, r) for r in range(self.world)]
        overlap = lambda a, b: max(0, min(a[1], b[1]) - max(a[0], b[0]))
        self.in_splits = [overlap(have[self.rank], need[r]) for r in range(self.world)]      # what I send to r
        self.out_splits = [overlap(have[r], need[self.rank]) for r in range(self.world)]     # what r sends to me
        assert sum(self.in_splits) == self.hi - self.lo and sum(self.out_splits) == (self.bhi - self.blo) * S
        self.send = torch.zeros((max(self.hi - self.lo, 1),) + tuple(item_shape), dtype=dtype, device=device)
        self.recv = torch.empty((max((self.bhi - self.blo) * S, 1),) + tuple(item_shape), dtype=dtype, device=device)
        self.bytes_received = sum(n for r, n in enumerate(self.out_splits) if r != self.rank) * self.recv[0].numel() * self.recv.element_size()

    def local_out(self):
        return self.send[:self.hi - self.lo]

    def exchange(self):
        """-> the frames of this rank's samples, in order ((bhi - blo) * S rows; empty for a surplus rank)."""
        n_out = sum(self.out_splits)
        if self.world == 1:
            return self.send[:n_out]                                       # everything stays where the kernel wrote it
        _timed_exchange(lambda: dist.all_to_all_single(self.recv[:n_out], self.send[:self.hi - self.lo], self.out_splits, self.in_splits,
                                                       group=self.group), self.bytes_received, self.recv.device)
        return self.recv[:n_out]


def gather_blocks(local, n_items, group=None):
    """All-gather of block-partitioned leading-dim chunks: `local` holds this rank's rows of a global
    (n_items, ...) tensor; returns the full tensor on every rank (one-off form of `FrameExchange`)."""
    ex = FrameExchange(n_items, tuple(local.shape[1:]), local.device, local.dtype, group)
    ex.local_out().copy_(local)
    ex.gather()
    return ex.rows(0, n_items)


class ShardedBevPath:
    """Wraps a `fiery_amd.Fiery` for multi-GPU inference of the hot path.

    `pool_frames(frame_lo, frame_hi, out) -> (n_local, C, X, Y)` (`out`: a preallocated destination or None) and
    `stack(bev (b*S, C, X, Y), batch_lo, batch_hi) -> dict` are supplied by the caller, so the bookkeeping is testable
    without a GPU.  `run` returns this rank's outputs; in the 'frames' layout with fewer samples than ranks, rank r
    computes sample r % batch (the surplus ranks replicate a sample instead of idling: same latency, and every rank
    returns a dict)."""

    def __init__(self, group=None, layout='auto', exchange='all_gather'):
        assert layout in ('auto', 'batch', 'frames'), layout
        assert exchange in ('all_gather', 'all_to_all'), exchange
        self.group = group
        self.requested = layout
        self.exchange_kind = exchange
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._exchange = {}

    def layout(self, batch):
        if self.requested != 'auto':
            assert self.requested == 'frames' or batch >= self.world, 'batch sharding needs a sample per rank'
            return self.requested
        return 'batch' if batch >= self.world else 'frames'

    def exchange(self, n_frames, frame_shape, device, dtype=torch.float32):
        key = (n_frames, tuple(frame_shape), str(device), dtype)
        ex = self._exchange.get(key)
        if ex is None:
            ex = self._exchange[key] = FrameExchange(n_frames, frame_shape, device, dtype, self.group)
        return ex

    def scatter(self, batch, frames_per_sample, frame_shape, device, dtype=torch.float32):
        key = ('a2a', batch, frames_per_sample, tuple(frame_shape), str(device), dtype)
        sc = self._exchange.get(key)
        if sc is None:
            sc = self._exchange[key] = FrameScatter(batch, frames_per_sample, frame_shape, device, dtype, self.group)
        return sc

    def run(self, batch, frames_per_sample, pool_frames, stack, frame_shape=None, device='cpu'):
        S = frames_per_sample
        if self.layout(batch) == 'batch':
            lo, hi = block_range(batch, self.world, self.rank)
            return stack(pool_frames(lo * S, hi * S, None), lo, hi)
        n_frames = batch * S
        flo, fhi = block_range(n_frames, self.world, self.rank)
        if self.exchange_kind == 'all_to_all':
            # frame sharding + one all-to-all-v: every frame goes to the one rank that owns its sample
            if frame_shape is None:                  # (tests) learn the shape from a first result
                local = pool_frames(flo, fhi, None)
                sc = self.scatter(batch, S, tuple(local.shape[1:]), local.device, local.dtype)
                sc.local_out().copy_(local)
            else:
                sc = self.scatter(batch, S, frame_shape, device)
                pool_frames(flo, fhi, sc.local_out())
            mine = sc.exchange()
            if sc.bhi == sc.blo:
                return None                          # a surplus rank (batch < world): it pooled frames, it owns no sample
            return stack(mine, sc.blo, sc.bhi)
        # frame sharding + one all-gather
        if frame_shape is None:                      # (tests) learn the shape from a first result
            local = pool_frames(flo, fhi, None)
            ex = self.exchange(n_frames, tuple(local.shape[1:]), local.device, local.dtype)
            ex.local_out().copy_(local)
        else:
            ex = self.exchange(n_frames, frame_shape, device)
            pool_frames(flo, fhi, ex.local_out())
        ex.gather()
        if batch >= self.world:
            blo, bhi = block_range(batch, self.world, self.rank)
        else:
            blo = self.rank % batch                   # ranks beyond the batch replicate a sample (latency mode)
            bhi = blo + 1
        return stack(ex.rows(blo * S, bhi * S), blo, bhi)


def sharded_bev_forward(model, K, E, ego, lifted=None, depth_logits=None, features=None, group=None, noise=None,
                        layout='auto', exchange='all_gather'):
    """Hot path of `model` over the *global* batch held (replicated) by every rank; returns this rank's
    samples' outputs (dict; None for a rank that owns no sample: exchange='all_to_all' with fewer samples than ranks) and
    the [lo, hi) batch range they cover.  exchange: how the 'frames' layout moves the pooled maps - 'all_gather' (every
    rank receives every frame) or 'all_to_all' (every frame goes to its sample's owner only, `FrameScatter`)."""
    from .model import pack_sequence_dim
    eng = model.engine()
    rf = model.receptive_field
    K, E, ego = K[:, :rf].contiguous(), E[:, :rf].contiguous(), ego[:, :rf].contiguous()
    B = K.shape[0]
    n = K.shape[2]
    Kf, Ef = pack_sequence_dim(K), pack_sequence_dim(E)
    owned = {}
    frame_shape = (model.encoder_out_channels,) + tuple(model.bev_size)

    def pool_frames(lo, hi, out):
        if hi == lo:
            return torch.zeros((0,) + frame_shape, device=K.device)
        geo = eng.geometry(Kf[lo:hi], Ef[lo:hi], model._camera_matrices(Kf[lo:hi], Ef[lo:hi]))
        if lifted is not None:
            x = lifted[:, :rf].reshape(B * rf, *lifted.shape[2:])[lo:hi].permute(0, 1, 3, 4, 5, 2)
            return eng.pool(x, geo, out=out)
        dl = depth_logits[:, :rf].reshape(B * rf, n, *depth_logits.shape[3:])[lo:hi]
        ft = features[:, :rf].reshape(B * rf, n, *features.shape[3:])[lo:hi]
        return eng.pool_fused(dl, ft, geo, out=out)

    def stack(bev, blo, bhi):
        owned['range'] = (blo, bhi)
        nz = noise[blo:bhi] if noise is not None else None
        theta = model._warp_transforms(ego)                  # over the whole batch, like the single-process path
        theta = None if theta is None else theta[blo:bhi]
        if model.sample_streams and bhi - blo > 1 and bev.is_cuda:
            return model._bev_stack_per_sample(bev, ego[blo:bhi], None, nz, theta)
        return eng.bev_stack(bev, ego[blo:bhi], None, nz, theta=theta)

    sharder = getattr(model, '_sharder', None)
    if sharder is None or sharder.group is not group or sharder.requested != layout or sharder.exchange_kind != exchange:
        sharder = model._sharder = ShardedBevPath(group, layout, exchange)
    out = sharder.run(B, rf, pool_frames, stack, frame_shape=frame_shape, device=K.device)
    return out, owned.get('range')


def sharded_bev_forward_graph(model, K, E, ego, lifted=None, depth_logits=None, features=None, group=None, noise=None,
                              layout='auto', exchange='all_gather'):
    """`sharded_bev_forward` replayed from a captured hipGraph: the exchange of the 'frames' layout - the RCCL all-gather /
    all-to-all-v, enqueued by torch.distributed on the capturing stream - is a node of the graph between the pooling kernel
    and the BEV stack, so a step is ONE graph launch per rank like the batch layout's (round 4 measured the eager form of this
    layout 14 % behind the replayed batch layout at one rank: host enqueue of ~130 launches).  Every rank captures and replays
    in step (the collective is part of each rank's graph).  Keyed on the argument buffers like `Fiery.bev_forward_graph`; the
    returned tensors belong to the graph.  reference call site: /root/reference/train.py:34-43 (DDP over the batch)."""
    model._require_eval()
    args = dict(K=K, E=E, ego=ego, lifted=lifted, depth_logits=depth_logits, features=features, noise=noise)
    key = ('sharded', layout, exchange, id(group), model._engine_generation, model.sample_streams, model.camera_matrix_mode,
           model.warp_transform_mode) + tuple((k,) if v is None else (k, v.data_ptr(), tuple(v.shape), tuple(v.stride()), v.dtype)
                                              for k, v in args.items())
    entry = model._graphs.get(key)
    if entry is None:
        call = lambda: sharded_bev_forward(model, K, E, ego, lifted=lifted, depth_logits=depth_logits, features=features, group=group,
                                           noise=noise, layout=layout, exchange=exchange)
        with torch.no_grad():
            call()                                    # eager once: buffers, workspaces, tile choices, the communicator
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # (thread-local capture mode: the process group's watchdog thread may query events while this thread captures)
            from . import ops
            with torch.cuda.graph(graph, stream=ops.prepare_capture(K.device), capture_error_mode='thread_local'):
                out, owned = call()
        while len(model._graphs) >= 4:
            model._graphs.pop(next(iter(model._graphs)))
        entry = model._graphs[key] = (graph, out, owned, args)
    entry[0].replay()
    return entry[1], entry[2]
