"""Multi-GPU execution of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU tests).

The path shards naturally: every (batch, frame) lift-splat is independent, and
everything after pooling is independent per batch element.  Two layouts:

* batch sharding (default whenever the batch covers the ranks): rank r owns a contiguous block of samples
  end to end; no data-path collective at all - what `bench.py --gpus N` measures (weak scaling).
* frame sharding for small batches (B < world, e.g. single-sample latency on 8 GPUs): the B*S frames are
  split across ranks for geometry + lift-splat, ONE all-gather moves the pooled BEV maps
  ((C, X, Y) fp32 = 10.24 MB per frame at baseline.yml) so that every rank holds all frames, then each rank
  runs the temporal / future-prediction / decoder stack for the samples it owns.  On the 8-GPU fully
  connected xGMI mesh an all-gather drives all seven links of a GPU at once, so the exchange is
  bandwidth-bound per link on 1/7 of the data rather than a 7-step ring.
"""
import math

import torch
import torch.distributed as dist


def block_range(n_items, world, rank):
    """Contiguous block partition: [lo, hi) of `n_items` owned by `rank` (ceil-sized blocks, tail may be empty)."""
    per = math.ceil(n_items / world)
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def owner_of(index, n_items, world):
    return index // math.ceil(n_items / world)


def gather_blocks(local, n_items, group=None):
    """All-gather of block-partitioned leading-dim chunks: `local` holds this rank's rows of a global
    (n_items, ...) tensor; returns the full tensor on every rank.  Blocks are padded to equal size for the
    collective (one all_gather_into_tensor call) and trimmed afterwards."""
    world = dist.get_world_size(group)
    per = math.ceil(n_items / world)
    padded = local.new_zeros((per,) + tuple(local.shape[1:]))
    padded[:local.shape[0]] = local
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    return out[:n_items]


class ShardedBevPath:
    """Wraps a `fiery_amd.Fiery` for multi-GPU inference of the hot path.

    `pool_frames(frame_lo, frame_hi) -> (n_local, C, X, Y)` and `stack(bev (b*S, C, X, Y), batch_lo, batch_hi)
    -> dict` are supplied by the caller, so the bookkeeping is testable without a GPU."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def layout(self, batch):
        return 'batch' if batch >= self.world else 'frames'

    def run(self, batch, frames_per_sample, pool_frames, stack):
        S = frames_per_sample
        if self.world == 1:
            return stack(pool_frames(0, batch * S), 0, batch)
        if self.layout(batch) == 'batch':
            lo, hi = block_range(batch, self.world, self.rank)
            if lo == hi:
                return None
            return stack(pool_frames(lo * S, hi * S), lo, hi)
        # frame sharding + one all-gather
        n_frames = batch * S
        flo, fhi = block_range(n_frames, self.world, self.rank)
        local = pool_frames(flo, fhi)
        bev = gather_blocks(local, n_frames, self.group)
        b = self.rank % batch                       # ranks beyond the batch replicate a sample (latency mode)
        return stack(bev[b * S:(b + 1) * S], b, b + 1)


def sharded_bev_forward(model, K, E, ego, lifted=None, depth_logits=None, features=None, group=None, noise=None):
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

    def pool_frames(lo, hi):
        if hi == lo:
            return torch.zeros(0, model.encoder_out_channels, *model.bev_size, device=K.device)
        geo = eng.geometry(Kf[lo:hi], Ef[lo:hi])
        if lifted is not None:
            x = lifted[:, :rf].reshape(B * rf, *lifted.shape[2:])[lo:hi].permute(0, 1, 3, 4, 5, 2)
            return eng.pool(x, geo)
        dl = depth_logits[:, :rf].reshape(B * rf, n, *depth_logits.shape[3:])[lo:hi]
        ft = features[:, :rf].reshape(B * rf, n, *features.shape[3:])[lo:hi]
        return eng.pool_fused(dl, ft, geo)

    def stack(bev, blo, bhi):
        owned['range'] = (blo, bhi)
        nz = noise[blo:bhi] if noise is not None else None
        return eng.bev_stack(bev.contiguous(), ego[blo:bhi], None, nz)

    out = ShardedBevPath(group).run(B, rf, pool_frames, stack)
    return out, owned.get('range')
