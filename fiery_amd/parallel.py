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

    def local_out(self):
        """Where this rank's items go (rows [0, hi - lo) of the send buffer): hand it to the producer as `out=`."""
        return self.send[:self.hi - self.lo]

    def gather(self):
        if self.world == 1 and not dist.is_initialized():
            self.recv.copy_(self.send)
        else:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
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

    def __init__(self, group=None, layout='auto'):
        assert layout in ('auto', 'batch', 'frames'), layout
        self.group = group
        self.requested = layout
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

    def run(self, batch, frames_per_sample, pool_frames, stack, frame_shape=None, device='cpu'):
        S = frames_per_sample
        if self.layout(batch) == 'batch':
            lo, hi = block_range(batch, self.world, self.rank)
            return stack(pool_frames(lo * S, hi * S, None), lo, hi)
        # frame sharding + one all-gather
        n_frames = batch * S
        flo, fhi = block_range(n_frames, self.world, self.rank)
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
                        layout='auto'):
    """Hot path of `model` over the *global* batch held (replicated) by every rank; returns this rank's
    samples' outputs (dict) and the [lo, hi) batch range they cover."""
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
    if sharder is None or sharder.group is not group or sharder.requested != layout:
        sharder = model._sharder = ShardedBevPath(group, layout)
    out = sharder.run(B, rf, pool_frames, stack, frame_shape=frame_shape, device=K.device)
    return out, owned.get('range')
