"""Host-side operator objects over the C ABI: pixel-major (NHWC) buffers and convolution ops.

This is the layer `fiery_amd.engine` composes the BEV stack from.  It only prepares arguments
(weight packing, BatchNorm folding, descriptors); all arithmetic runs in libfiery_hip.so.
"""
import ctypes as C
import os

import torch

from . import native

UNIT = 8          # input channels are consumed in units of 8 floats

# When a list is installed here (bench.py does, for its instrumented step), every launch appends
# (kind, start_event, end_event, algorithmic_work) - flops for convolutions, bytes for pooling.
PROFILE_SINK = None
# matrix-core precision of the convolutions built from here on: native.PRECISION_F32 (the reference's arithmetic, the parity
# configuration) or native.PRECISION_BF16 (BASELINE.json configs[3] / [4]); `BevEngine` sets it from `model.conv_precision`
DEFAULT_PRECISION = native.PRECISION_F32
AUTOTUNE = os.environ.get('FIERY_CONV_AUTOTUNE', '1') != '0'     # time both tile heights once per conv shape (GPU only)


def profiled(kind, work, stream_tensor, fn, detail=None):
    if PROFILE_SINK is None or not stream_tensor.is_cuda:
        return fn()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    out = fn()
    end.record()
    PROFILE_SINK.append((kind, start, end, work, detail))
    return out


def pool_algorithmic_bytes(detail):
    """Algorithmic HBM bytes of one pooling call recorded by `profiled('voxel_pool', ...)`:
    every in-grid point's C features once, every point's geometry once, the dense output once.  N_kept is counted with
    `fiery_voxel_index` on the geometry the call read (inference calls leave no voxel ranks in their workspace since round 6:
    FIERY_POOL_NO_RANKS) - call it after the step, outside any timed region."""
    n_points = detail['points']
    rank, _ = detail['lib'].voxel_index(detail['geometry'], detail['grid'], want_idx=False)
    n_kept = int((rank >= 0).sum().item())
    c = detail['channels']
    return 4.0 * c * n_kept + 12.0 * n_points + 4.0 * c * detail['voxels'] * detail['frames'], n_kept


def _chain_split_image(w, blocks_outer):
    """The split image of a chained 1 x 1's weights for the split tile kernels (`fiery_conv_desc.weights2_split / weights3_split`):
    w (rows, K) fp32 with rows a multiple of 32 and K a multiple of 16 -> bf16 [3 terms][a][b][64 lanes][8], every value as three
    bf16 terms that add up to it exactly; lane = (row % 32, hi), element j = column 16 h + (j < 4 ? 4 hi + j : 8 + 4 hi + j - 4).
    blocks_outer = False: [a][b] = [k half h][row block] (weights2: 64 rows x 32 mid channels);
    blocks_outer = True:  [a][b] = [k block of 32][half inside it] with ONE row block (weights3: 32 rows x 64 channels)."""
    w = w.detach().float().cpu()
    rows, K = w.shape
    t1 = w.bfloat16().float()
    r1 = w - t1
    t2 = r1.bfloat16().float()
    terms = torch.stack([t1, t2, r1 - t2])                                         # (3, rows, K); the last is rounded below
    hi = torch.arange(2).view(2, 1)
    j = torch.arange(8).view(1, 8)
    col_in_half = torch.where(j < 4, 4 * hi + j, 8 + 4 * hi + j - 4)               # (hi, j) -> column inside a 16-column half
    halves = K // 16
    cols = (16 * torch.arange(halves).view(halves, 1, 1) + col_in_half.view(1, 2, 8))          # (half, hi, j)
    g = terms[:, :, cols]                                                          # (3, rows, half, hi, j)
    g = g.view(3, rows // 32, 32, halves, 2, 8)                                    # (term, row block, m, half, hi, j)
    if blocks_outer:
        assert rows == 32
        img = g.permute(0, 3, 4, 2, 5, 1).reshape(3, halves // 2, 2, 2, 32, 8)     # (term, k block, half in block, hi, m, j)
    else:
        img = g.permute(0, 3, 1, 4, 2, 5)                                          # (term, half, row block, hi, m, j)
    return img.contiguous().bfloat16().reshape(-1)


def round_up(v, m):
    return (v + m - 1) // m * m


class Buf:
    """A pixel-major activation view: `n_img` images of H x W pixels, `C` channels wide inside rows of
    `ld` floats.  Element (img, y, x, c) lives at base + img*img_stride + (y*W + x)*ld + c.  Views can
    select a channel slice (writers of a concatenation target its slices) or a strided image subset
    (one time step of a (batch, time) sequence)."""

    def __init__(self, tensor, n_img, H, W, C, ld=None, img_stride=None, base_off=0):
        self.tensor = tensor                 # keeps the storage alive
        self.n_img, self.H, self.W, self.C = n_img, H, W, C
        self.ld = ld if ld is not None else tensor.shape[-1]
        self.img_stride = img_stride if img_stride is not None else H * W * self.ld
        self.base_off = base_off             # floats from the start of `tensor`

    @staticmethod
    def alloc(n_img, H, W, C, device, zero=True):
        width = round_up(C, UNIT)
        make = torch.zeros if zero else torch.empty
        return Buf(make(n_img, H, W, width, dtype=torch.float32, device=device), n_img, H, W, width)

    @property
    def ptr(self):
        return self.tensor.data_ptr() + 4 * self.base_off

    def slice(self, c_off, C):
        assert c_off % 4 == 0 and c_off + C <= self.C
        return Buf(self.tensor, self.n_img, self.H, self.W, C, self.ld, self.img_stride, self.base_off + c_off)

    def images(self, first, count, step=1):
        """Images first, first+step, ... (count of them)."""
        assert first + (count - 1) * step < self.n_img
        return Buf(self.tensor, count, self.H, self.W, self.C, self.ld, self.img_stride * step,
                   self.base_off + first * self.img_stride)

    def nhwc(self):
        """(n_img, H, W, C) torch view of the data."""
        return torch.as_strided(self.tensor, (self.n_img, self.H, self.W, self.C),
                                (self.img_stride, self.W * self.ld, self.ld, 1),
                                self.tensor.storage_offset() + self.base_off)

    def to_nchw(self):
        return self.nhwc().permute(0, 3, 1, 2).contiguous()

    def as_nhwc_struct(self):
        s = native.Nhwc()
        s.ptr, s.ld, s.img_stride = self.ptr, self.ld, self.img_stride
        return s


def _null_nhwc():
    s = native.Nhwc()
    s.ptr, s.ld, s.img_stride = None, 0, 0
    return s


def fold_bn(bn, cout, conv_bias=None, eps=None):
    """(scale, shift) of `BN(conv + bias)` in inference mode; identity scale when bn is None."""
    if bn is None:
        scale = torch.ones(cout, dtype=torch.float32)
        shift = torch.zeros(cout, dtype=torch.float32) if conv_bias is None else conv_bias.detach().float().cpu().clone()
        return scale, shift
    w, b = bn.weight.detach().float().cpu(), bn.bias.detach().float().cpu()
    mean, var = bn.running_mean.detach().float().cpu(), bn.running_var.detach().float().cpu()
    scale = w / torch.sqrt(var + (bn.eps if eps is None else eps))
    shift = b - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().float().cpu() * scale
    return scale, shift


class ConvOp:
    """One convolution (+ folded BN / bias, activation, residual, GRU epilogues) ready to launch.

    weight   : (Cout, Cin_total, kH, kW) or (Cout, Cin_total, kT, kH, kW) dense tensor (any device)
    chan_map : for every logical input channel its position in the padded, unit-aligned channel
               space formed by source 0 followed by source 1
    units    : (units0, units1) 8-channel units read from each source
    """

    def __init__(self, lib, weight, chan_map, units, scale, shift, device, stride=1, pad=None,
                 act=native.ACT_NONE, epi=native.EPI_PLAIN, res_before_act=False, precision=None, tune=True, forms=None):
        """tune: time both tile heights the first time a shape is launched (`_pick_tile`); False for one-shot ops.
        precision: native.PRECISION_F32 / PRECISION_BF16 (None: `ops.DEFAULT_PRECISION`) - bf16 rounds the matrix-core
        operands (weights here, activations on chip), accumulates in fp32; launches the bf16 kernel does not cover run in fp32."""
        self.lib = lib
        self.precision = DEFAULT_PRECISION if precision is None else precision
        # the optional forms whose weight images are packed (each is one device kernel per op): all of them for ops that time their
        # candidates, the fp32 Winograd image alone for one-shot ops (the training graph packs per call and names what it wants)
        if forms is None:
            forms = ('wino', 'wsplit', 'split') if tune else ('wino',)
        w = weight.detach().to(device=device, dtype=torch.float32).contiguous()
        self.cout, self.cin_total = w.shape[0], w.shape[1]
        kernel = tuple(w.shape[2:])
        self.kT, self.kH, self.kW = (1,) * (3 - len(kernel)) + kernel
        self.stride = stride
        self.padH, self.padW = ((self.kH - 1) // 2, (self.kW - 1) // 2) if pad is None else pad
        self.units = tuple(units)
        cin_units = sum(self.units)
        taps = self.kT * self.kH * self.kW
        assert len(chan_map) == self.cin_total
        self.packed = lib.conv_pack_weights(w.view(self.cout, self.cin_total, taps), self.cout, self.cin_total, taps,
                                            list(chan_map), cin_units)
        self.packed_bf16 = None
        if self.precision == native.PRECISION_BF16:
            self.packed_bf16 = lib.conv_pack_weights_bf16(w.view(self.cout, self.cin_total, taps), self.cout, self.cin_total,
                                                          taps, list(chan_map), cin_units)
        self.cout_pad = round_up(self.cout, 32)
        # the split image (fp32 accuracy on the bf16 matrix cores: three bf16 terms per operand) for the layers the library's
        # split tile kernels cover - whole 32-channel stages, 32- or 64-wide cout tiles: a candidate form of `_pick_tile` ('split')
        self.packed_split = None
        if ('split' in forms and SPLIT_TILES and self.precision == native.PRECISION_F32 and cin_units % 4 == 0 and self.units[0] % 4 == 0 and
                self.cout_pad % 128 != 0):
            self.packed_split = lib.conv_pack_weights_split(w.view(self.cout, self.cin_total, taps), self.cout, self.cin_total, taps,
                                                            list(chan_map), cin_units)
        # Winograd F(2x2, 3x3) image of the weights for the layers that form covers (3 x 3 / stride 1 / 'same', whole 16-channel
        # stages per source, 64-cout tiles; fp32 only): a candidate form of `_pick_tile`
        self.packed_winograd = self.packed_winograd_split = None
        if (WINOGRAD and ('wino' in forms or 'wsplit' in forms) and self.precision == native.PRECISION_F32 and (self.kT, self.kH, self.kW) == (1, 3, 3) and stride == 1 and
                (self.padH, self.padW) == (1, 1) and self.cout_pad % 64 == 0 and all(u % 2 == 0 for u in self.units) and
                # (what the library's scalar-addressed loop - the only one with a Winograd form - asks of the channel layout:
                # whole 32-channel stages per tap and in source 0; a launch can still fall back on extents, see `_winograd_taken`)
                cin_units % 4 == 0 and self.units[0] % 4 == 0 and cin_units >= 4):
            if 'wino' in forms:
                self.packed_winograd = lib.conv_pack_weights_winograd(w.view(self.cout, self.cin_total, taps), self.cout, self.cin_total,
                                                                      list(chan_map), cin_units)
            # the same form on the bf16 matrix cores, every operand as three bf16 terms (fp32 accuracy; 'wsplit')
            if WINOGRAD_SPLIT and 'wsplit' in forms:
                self.packed_winograd_split = lib.conv_pack_weights_winograd_split(
                    w.view(self.cout, self.cin_total, taps), self.cout, self.cin_total, list(chan_map), cin_units)
        if torch.is_tensor(scale) and scale.device == w.device and scale.numel() == self.cout_pad:
            self.scale, self.shift = scale, shift          # already padded, already resident
        else:
            sc = torch.zeros(self.cout_pad, dtype=torch.float32)
            sh = torch.zeros(self.cout_pad, dtype=torch.float32)
            sc[:self.cout] = scale
            sh[:self.cout] = shift
            self.scale, self.shift = sc.to(device), sh.to(device)
        self.act, self.epi, self.res_before_act = act, epi, res_before_act
        self.tune = tune
        self.chain = None
        self.chain3 = None
        self.heads = None
        self.force_form = None       # tests / A-B runs: 64, 128, 'sk', 'wino', 'wsplit' or 'split' instead of the measured choice
        self.last_form = None


    def chain_pointwise(self, weight, scale, shift, act):
        """Fuse a following 1x1 convolution (Cin <= 32 = this op's padded outputs, Cout <= 64) into this kernel:
        its input tile never leaves the chip.  `weight` (Cout2, Cin2[, 1, 1])."""
        assert self.cout_pad == 32 and self.epi == native.EPI_PLAIN, 'chaining needs a 32-channel plain convolution'
        w = weight.detach().float().reshape(weight.shape[0], weight.shape[1])
        cout2, cin2 = w.shape
        assert cin2 == self.cout and cout2 <= 64
        device = self.packed.device
        w64 = torch.zeros(64, cin2, dtype=torch.float32, device=device)     # pack as a 64-wide tile (BN = 64 image)
        w64[:cout2] = w.to(device)
        packed = self.lib.conv_pack_weights(w64.contiguous(), 64, cin2, 1, list(range(cin2)), 4)
        sc = torch.zeros(64, dtype=torch.float32)
        sh = torch.zeros(64, dtype=torch.float32)
        sc[:cout2], sh[:cout2] = scale, shift
        self.chain = dict(w=packed, scale=sc.to(device), shift=sh.to(device), act=act, cout=cout2, w_split=None)
        if self.packed_split is not None:              # (the split tile kernel multiplies the chained products split too)
            w32k = torch.zeros(64, 32, dtype=torch.float32)
            w32k[:cout2, :cin2] = w.cpu()
            self.chain['w_split'] = _chain_split_image(w32k, blocks_outer=False).to(device)
        return self

    def chain_next(self, weight, scale, shift, act):
        """After a `chain_pointwise` up-projection: also apply the NEXT block's 1x1 down-projection (Cin = 64 = the chained
        result, Cout <= 32) to the finished tile - residual included - and write it to `out3` of the call.  Returns a new
        op that shares this one's packed weights (the plain op stays usable on its own)."""
        assert self.chain is not None and self.chain['cout'] == 64, 'chain_next follows a 64-channel chained 1x1'
        w = weight.detach().float().reshape(weight.shape[0], weight.shape[1])
        cout3, cin3 = w.shape
        assert cin3 == 64 and cout3 <= 32
        device = self.packed.device
        w32 = torch.zeros(32, 64, dtype=torch.float32, device=device)
        w32[:cout3] = w.to(device)
        import copy
        op = copy.copy(self)
        sc = torch.zeros(32, dtype=torch.float32)
        sh = torch.zeros(32, dtype=torch.float32)
        sc[:cout3], sh[:cout3] = scale, shift
        op.chain3 = dict(w=self.lib.conv_pack_weights(w32.contiguous(), 32, 64, 1, list(range(64)), 8), scale=sc.to(device),
                         shift=sh.to(device), act=act, cout=cout3, w_split=None)
        if self.packed_split is not None:
            op.chain3['w_split'] = _chain_split_image(w32.cpu(), blocks_outer=True).to(device)
        return op

    def attach_heads(self, weight, bias, groups, sigmoids):
        """Turn this convolution into the decoder-heads form (FIERY_EPI_HEADS): its activated output - 64 hidden
        channels per head - stays on chip and only the final 1x1 rows are stored.  weight (n_out, 64), bias (n_out,),
        groups[o] = which 64-channel group row o reads, sigmoids[o] = apply a sigmoid to row o."""
        assert self.cout_pad % 128 == 0 and self.chain is None and self.epi == native.EPI_PLAIN
        n_out = weight.shape[0]
        assert weight.shape == (n_out, 64) and n_out <= native.MAX_HEAD_OUTPUTS and len(groups) == len(sigmoids) == n_out
        device = self.packed.device
        self.heads = dict(w=weight.detach().float().contiguous().to(device), b=bias.detach().float().contiguous().to(device),
                          groups=[int(g) for g in groups], sigmoids=[int(bool(v)) for v in sigmoids], n_out=n_out)
        self.epi = native.EPI_HEADS
        return self

    def out_hw(self, H, W):
        return ((H + 2 * self.padH - self.kH) // self.stride + 1, (W + 2 * self.padW - self.kW) // self.stride + 1)

    def _set_form(self, d, form, sk=None):
        if form == 'split':                            # the split tile kernels: 128-pixel tiles
            d.precision, d.weights_bf16 = native.PRECISION_F32_SPLIT, self.packed_split.data_ptr()
            d.weights2_split, d.weights3_split = self._chain_split_ptrs()
            form = 128
        else:
            d.weights_bf16 = self.packed_bf16.data_ptr() if self.packed_bf16 is not None else None
            d.precision = self.precision if self.packed_bf16 is not None else native.PRECISION_F32
            d.weights2_split = d.weights3_split = None
        d.winograd = 1 if form == 'wino' else native.WINOGRAD_SPLIT_TERMS if form == 'wsplit' else 0
        d.weights_winograd = (self.packed_winograd.data_ptr() if form == 'wino' else
                              self.packed_winograd_split.data_ptr() if form == 'wsplit' else None)
        if form in ('wino', 'wsplit'):
            form = 0
        d.tile_m, d.stream_k = (128, 1) if form == 'sk' else (form, 0)
        if form == 'sk':
            d.sk_workspace, d.sk_workspace_bytes = sk['ws'].data_ptr(), sk['ws'].numel() * 4
            d.sk_counters, d.sk_counters_len = sk['cnt'].data_ptr(), sk['cnt'].numel()
        else:
            d.sk_workspace = d.sk_counters = None
            d.sk_workspace_bytes = d.sk_counters_len = 0

    @property
    def _sig(self):
        """What identifies this layer in the process-wide table of measured forms (`FORM_TABLE`): everything a launch's
        candidates and their timings depend on except the output shape."""
        return (self.kT, self.kH, self.kW, self.stride, self.cin_total, self.cout, self.units, self.act, self.epi,
                int(self.res_before_act), self.precision, self.chain['cout'] if self.chain else 0,
                self.chain3['cout'] if self.chain3 else 0, self.heads['n_out'] if self.heads else 0)

    @property
    def _tile_m(self):
        """This layer's measured forms, {(n_img, H, W) of the output: form} (a view of `FORM_TABLE`)."""
        sig = self._sig
        return {k[1]: v for k, v in FORM_TABLE.items() if k[0] == sig}

    def _winograd_taken(self, d):
        """Whether the library runs THIS launch as Winograd when asked to (the packed image exists for every 3 x 3 / stride 1
        layer with whole 16-channel stages; the launch also needs the aligned, 16-byte addressable variant, an unchained
        epilogue, ... - `fiery_conv_form_used` knows).  Leaves the descriptor's form members as it found them."""
        if self.packed_winograd is None and self.packed_winograd_split is None:
            return False
        keep = (d.winograd, d.weights_winograd)
        if self.packed_winograd is not None:
            d.winograd, d.weights_winograd = 1, self.packed_winograd.data_ptr()
        else:
            d.winograd, d.weights_winograd = native.WINOGRAD_SPLIT_TERMS, self.packed_winograd_split.data_ptr()
        taken = self.lib.conv_form_used(d) in (native.CONV_FORM_WINOGRAD, native.CONV_FORM_WINOGRAD_SPLIT)
        d.winograd, d.weights_winograd = keep
        return taken

    def _split_taken(self, d):
        """Whether the library runs THIS launch in the split tile form when asked to (`fiery_conv_precision_used`)."""
        if self.packed_split is None:
            return False
        keep = (d.precision, d.weights_bf16, d.tile_m, d.winograd, d.stream_k, d.weights2_split, d.weights3_split)
        d.precision, d.weights_bf16, d.tile_m, d.winograd, d.stream_k = native.PRECISION_F32_SPLIT, self.packed_split.data_ptr(), 128, 0, 0
        d.weights2_split, d.weights3_split = self._chain_split_ptrs()
        taken = self.lib.conv_precision_used(d) == native.PRECISION_F32_SPLIT
        d.precision, d.weights_bf16, d.tile_m, d.winograd, d.stream_k, d.weights2_split, d.weights3_split = keep
        return taken

    def _chain_split_ptrs(self):
        w2 = self.chain['w_split'] if self.chain is not None else None
        w3 = self.chain3['w_split'] if self.chain3 is not None else None
        return (w2.data_ptr() if w2 is not None else None, w3.data_ptr() if w3 is not None else None)

    def _pick_tile(self, d, out):
        """The form of this launch: a tile height (64 / 128 output pixels per workgroup, one workgroup per tile), or
        stream-K (the launch's work dealt evenly to one round of workgroups, shared tiles summed through a workspace -
        `fiery_conv_desc.stream_k`).  The same launches repeat every step, so the first time a shape is seen on the GPU the
        candidates are timed (HIP events, on the launch stream) and the fastest is kept - the partly filled last round of
        workgroups makes the better choice shape-dependent.  The convolution is a pure function of its
        inputs, so the extra launches leave the same result behind (up to the last bits between stream-K and the tile
        forms: another summation split)."""
        self._set_form(d, 0)
        self.last_form = 0                             # (what the last launch was asked to run in: tests look here)
        narrow = self.cout_pad % 64 != 0 or self.chain is not None       # one fp32 tile shape: the choice is fp32 / split
        if (narrow and self.packed_split is None) or (not self.tune and self.force_form is None):
            return                                     # one tile shape only (or: library heuristic; a forced form is honoured)
        if self.heads is not None and self.packed_winograd is None and self.packed_winograd_split is None:
            return                                     # (the heads' direct form has one tile shape)
        key = (out.n_img, out.H, out.W)
        choice = self.force_form                       # (a forced form - per op, or FIERY_CONV_FORM for all - goes before the table)
        if choice is None and FORCE_FORM:
            choice = FORCE_FORM if FORCE_FORM in ('sk', 'wino', 'wsplit', 'split') else int(FORCE_FORM)
        if narrow and choice not in (None, 0, 'split'):
            choice = 0
        if choice is None:
            choice = FORM_TABLE.get((self._sig, key))
        if choice == 'wino' and self.packed_winograd is None:
            choice = 0
        if choice == 'wsplit' and self.packed_winograd_split is None:
            choice = 'wino' if self.packed_winograd is not None else 0
        if choice == 'split' and not self._split_taken(d):
            choice = 0
        sk = _stream_k_workspace(self.lib, d, out.tensor) if (choice == 'sk' or choice is None) and self.heads is None and not narrow else None
        if choice is None:
            if FORM_TABLE_FROZEN or not _autotune_enabled(out.tensor):
                return                                 # library heuristic (and nothing cached: tune when possible)
            wino_forms = ((['wino'] if self.packed_winograd is not None else []) +
                          (['wsplit'] if self.packed_winograd_split is not None else [])) if self._winograd_taken(d) else []
            forms = [64, 128] + (['sk'] if sk is not None else []) + wino_forms
            if self.heads is not None:
                forms = [0] + wino_forms               # heads epilogue: the direct form's one tile shape, or Winograd
            if narrow:
                forms = [0]
            if self.heads is None and self._split_taken(d):
                forms.append('split')
            if len(forms) == 1:
                FORM_TABLE[(self._sig, key)] = forms[0]
                return
            times = {f: float('inf') for f in forms}
            for _trial in range(2):                    # alternate the candidates, keep each one's best trial
                for form in forms:
                    self._set_form(d, form, sk)
                    self.lib.conv_fwd(d, out.tensor)   # warm
                    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    start.record()
                    for _ in range(2):
                        self.lib.conv_fwd(d, out.tensor)
                    end.record()
                    end.synchronize()
                    times[form] = min(times[form], start.elapsed_time(end))
            choice = FORM_TABLE[(self._sig, key)] = min(times, key=times.get)
        if choice == 'sk' and sk is None:
            choice = 0                                 # (no workspace for this stream, e.g. first met inside a capture)
        self._set_form(d, choice, sk)
        self.last_form = choice

    def __call__(self, srcs, out, res=None, img_bias=None, out2=None, aux0=None, aux1=None,
                 T_out=1, t_out0=0, t_in_add=0, cout_store=None, img_bias_border=False, head_planes=None, out3=None):
        """srcs: list of (Buf, batch_stride, time_stride) or Buf (plain image batch).  With attached heads `out` is a
        `HeadsOut(n_img, H, W, tensor)` and head_planes[o] = (address of row o's plane of image 0, floats between images)."""
        d = native.ConvDesc()
        first = None
        for i in range(2):
            s = d.src[i]
            if i < len(srcs) and self.units[i] > 0:
                item = srcs[i]
                buf, bstride, tstride = item if isinstance(item, tuple) else (item, item.img_stride, 0)
                first = first or buf
                s.ptr = buf.ptr
                s.ld, s.units, s.batch_stride, s.time_stride = buf.ld, self.units[i], bstride, tstride
            else:
                s.ptr, s.ld, s.units, s.batch_stride, s.time_stride = None, 0, 0, 0, 0
        d.Hin, d.Win = first.H, first.W
        d.Hout, d.Wout = out.H, out.W
        d.n_img_out, d.T_out, d.t_out0, d.t_in_add = out.n_img, T_out, t_out0, t_in_add
        d.kT, d.kH, d.kW, d.stride, d.padH, d.padW = self.kT, self.kH, self.kW, self.stride, self.padH, self.padW
        d.weights, d.cout_pad = self.packed.data_ptr(), self.cout_pad
        d.scale, d.shift = self.scale.data_ptr(), self.shift.data_ptr()
        d.img_bias = img_bias.data_ptr() if img_bias is not None else None
        d.img_bias_border = int(bool(img_bias_border))
        d.act, d.epi, d.res_before_act = self.act, self.epi, int(self.res_before_act)
        d.res = res.as_nhwc_struct() if res is not None else _null_nhwc()
        if self.heads is not None:
            d.out = _null_nhwc()
            hd = d.heads
            hd.w, hd.bias, hd.n_out = self.heads['w'].data_ptr(), self.heads['b'].data_ptr(), self.heads['n_out']
            for o in range(self.heads['n_out']):
                hd.group[o], hd.sigmoid[o] = self.heads['groups'][o], self.heads['sigmoids'][o]
                hd.out[o], hd.img_stride[o] = head_planes[o]
        else:
            d.out = out.as_nhwc_struct()
        if self.chain is not None:
            d.weights2, d.scale2, d.shift2 = self.chain['w'].data_ptr(), self.chain['scale'].data_ptr(), self.chain['shift'].data_ptr()
            d.act2 = self.chain['act']
            d.cout_store = cout_store if cout_store is not None else min(64, round_up(self.chain['cout'], UNIT), out.C)
        else:
            d.weights2 = d.scale2 = d.shift2 = None
            d.act2 = 0
            d.cout_store = cout_store if cout_store is not None else min(self.cout_pad, round_up(self.cout, UNIT),
                                                                         getattr(out, 'C', self.cout_pad))
        if self.chain3 is not None and out3 is not None:
            c3 = self.chain3
            d.weights3, d.scale3, d.shift3, d.act3 = c3['w'].data_ptr(), c3['scale'].data_ptr(), c3['shift'].data_ptr(), c3['act']
            d.out3 = out3.as_nhwc_struct()
        else:
            d.weights3 = d.scale3 = d.shift3 = None
            d.act3 = 0
            d.out3 = _null_nhwc()
        d.out2 = out2.as_nhwc_struct() if out2 is not None else _null_nhwc()
        d.aux0 = aux0.as_nhwc_struct() if aux0 is not None else _null_nhwc()
        d.aux1 = aux1.as_nhwc_struct() if aux1 is not None else _null_nhwc()
        self._keep = (srcs, out, res, img_bias, out2, aux0, aux1, out3)
        self._pick_tile(d, out)                        # (sets the form's members: tile, stream-K, Winograd, precision + bf16 / split image)
        flops = 2.0 * out.n_img * out.H * out.W * self.cin_total * self.kT * self.kH * self.kW * self.cout
        if self.chain is not None:
            flops += 2.0 * out.n_img * out.H * out.W * self.cout * self.chain['cout']
        if self.chain3 is not None and out3 is not None:
            # (the third stage IS the next block's 1x1 down-projection, whose own launch is skipped: its flops are executed here)
            flops += 2.0 * out.n_img * out.H * out.W * 64 * self.chain3['cout']
        if self.heads is not None:
            flops += 2.0 * out.n_img * out.H * out.W * 64 * self.heads['n_out']
        used = 'f32'
        if PROFILE_SINK is not None and d.precision == native.PRECISION_BF16:
            used = 'bf16' if self.lib.conv_precision_used(d) == native.PRECISION_BF16 else 'f32'
        if PROFILE_SINK is not None and d.precision == native.PRECISION_F32_SPLIT:
            if self.lib.conv_precision_used(d) == native.PRECISION_F32_SPLIT:
                used = 'f32 split' if self.chain is None else 'f32 split + chain'
        if PROFILE_SINK is not None and (d.winograd or d.stream_k):
            form = self.lib.conv_form_used(d)          # (what ran, not what was asked for: the flops accounting hangs on it)
            used = {native.CONV_FORM_WINOGRAD: 'f32 winograd', native.CONV_FORM_WINOGRAD_SPLIT: 'f32 winograd split',
                    native.CONV_FORM_STREAM_K: 'f32 stream-K'}.get(form, used)
        profiled('conv_igemm', flops, out.tensor, lambda: self.lib.conv_fwd(d, out.tensor),
                 detail=(self.kT, self.kH, self.kW, self.stride, self.cin_total, self.cout, out.n_img, out.H, out.W, used))


# Stream-K workspaces: one per (device, stream) - launches on different streams may be in flight together -, allocated once at
# a fixed size (captured graphs keep their addresses) and never freed.  Partial tiles: 64 MiB covers one round of workgroups
# of either stream-K kernel (512 x 2 x 128 x 128 or 768 x 2 x 128 x 64 floats); the ticket counters start at zero and every
# launch leaves them at zero.
_SK_WORKSPACES = {}
SK_WORKSPACE_BYTES = 64 << 20
SK_COUNTERS = 1 << 17
STREAM_K = os.environ.get('FIERY_STREAM_K', '1') != '0'
FORCE_FORM = os.environ.get('FIERY_CONV_FORM')      # A/B runs: '64', '128', 'sk' or 'wino' for every launch that has the form, no timing
WINOGRAD = os.environ.get('FIERY_CONV_WINOGRAD', '1') != '0'
SPLIT_TILES = os.environ.get('FIERY_CONV_SPLIT', '1') != '0'      # the split tile kernels (bf16 matrix cores, three-term operands, fp32 accuracy) as a candidate
WINOGRAD_SPLIT = os.environ.get('FIERY_CONV_WINOGRAD_SPLIT', '1') != '0'      # the split form (bf16 matrix cores, three-term operands) as a candidate


# Measured forms, one table per process: (layer signature, (n_img, H, W) of the output) -> 64 / 128 / 'sk' / 'wino'.  The forms
# differ in fp32 rounding (another summation split, Winograd's transforms), and which one wins a timing can differ between runs
# and between ranks - so the table can be taken out (`form_table`), put back (`load_form_table`: a frozen table makes every
# launch reproducible - shapes it does not hold take the library's heuristic form, nothing is timed) and shared between the
# ranks of a job (`fiery_amd.parallel.share_conv_forms`: every rank runs rank 0's arithmetic).  FIERY_CONV_FORM_TABLE=<file>:
# loaded frozen at import when the file exists; `save_form_table(path)` writes it.
FORM_TABLE = {}
FORM_TABLE_FROZEN = False


def form_table():
    return dict(FORM_TABLE)


def load_form_table(table, frozen=True):
    global FORM_TABLE_FROZEN
    FORM_TABLE.clear()
    FORM_TABLE.update(table)
    FORM_TABLE_FROZEN = bool(frozen)


def save_form_table(path):
    import json
    with open(path, 'w') as fh:
        json.dump([[list(sig), list(key), form] for (sig, key), form in sorted(FORM_TABLE.items(), key=repr)], fh)


def read_form_table(path):
    import json

    def tup(v):
        return tuple(tup(x) for x in v) if isinstance(v, list) else v
    with open(path) as fh:
        return {(tup(sig), tup(key)): form for sig, key, form in json.load(fh)}


if os.environ.get('FIERY_CONV_FORM_TABLE') and os.path.exists(os.environ['FIERY_CONV_FORM_TABLE']):
    load_form_table(read_form_table(os.environ['FIERY_CONV_FORM_TABLE']), frozen=True)


def capture_stream(device):
    """The stream this library captures its hipGraphs on: one per device, made once - so that what is keyed on the stream (the
    stream-K workspaces) exists before a capture starts (`prepare_capture`) and the captured launches use the same forms as the
    eager pass in front of them."""
    key = ('capture', torch.device(device).index)
    st = _SK_WORKSPACES.get(key)
    if st is None:
        st = _SK_WORKSPACES[key] = torch.cuda.Stream(device=device)
    return st


def prepare_capture(device):
    """Call before `torch.cuda.graph(graph, stream=capture_stream(device))`: allocates the capture stream's stream-K workspace
    outside the capture."""
    st = capture_stream(device)
    if STREAM_K:
        dev = torch.device(device)
        key = (dev.type, dev.index, st.cuda_stream)
        if key not in _SK_WORKSPACES:
            _SK_WORKSPACES[key] = dict(ws=torch.empty(SK_WORKSPACE_BYTES // 4, dtype=torch.float32, device=dev),
                                       cnt=torch.zeros(SK_COUNTERS, dtype=torch.int32, device=dev))
    return st


def _stream_k_workspace(lib, d, t):
    """The stream-K workspace of the current stream if this launch has a stream-K form that fits it, else None."""
    if not STREAM_K:
        return None
    nbytes, n_cnt, n_wg = lib.conv_stream_k_plan(d)
    if n_wg == 0 or nbytes > SK_WORKSPACE_BYTES or n_cnt > SK_COUNTERS:
        return None
    key = (t.device.type, t.device.index, torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0)
    ws = _SK_WORKSPACES.get(key)
    if ws is None:
        if t.is_cuda and torch.cuda.is_current_stream_capturing():
            return None                                # (allocated outside captures only: the eager pass in front of one)
        ws = _SK_WORKSPACES[key] = dict(ws=torch.empty(SK_WORKSPACE_BYTES // 4, dtype=torch.float32, device=t.device),
                                        cnt=torch.zeros(SK_COUNTERS, dtype=torch.int32, device=t.device))
    return ws


class HeadsOut:
    """Stands in for the output Buf of a heads convolution: its shape, and a tensor that names the device/stream."""

    def __init__(self, n_img, H, W, tensor):
        self.n_img, self.H, self.W, self.tensor = n_img, H, W, tensor


def _autotune_enabled(t):
    return AUTOTUNE and t.is_cuda and PROFILE_SINK is None and not torch.cuda.is_current_stream_capturing()


def identity_chan_map(channels, offset=0):
    return [offset + i for i in range(channels)]


class VoxelPool(torch.autograd.Function):
    """`projection_to_birds_eye_view` as one differentiable operator - the seam the reference fills with
    `VoxelsSumming.apply` (fiery/utils/geometry.py:283-314, called at fiery/models/fiery.py:261).  Forward is
    `fiery_voxel_pool_fwd`; backward hands every in-grid point the gradient of its voxel (`fiery_voxel_pool_bwd`) using
    the voxel ranks the forward call left in its workspace.  Geometry gets no gradient, as in the reference
    (`ctx.mark_non_differentiable(geometry)`, geometry.py:300)."""

    @staticmethod
    def forward(ctx, x, geometry, engine):
        f, n, d, h, w, c = x.shape
        lib = engine.lib
        # a workspace of its own: the ranks must outlive later pooling calls until backward runs
        ws = lib.pool_workspace(f, n, d, h, w, x.device, engine.grid, engine.pool_tile, engine.pool_flags)
        out = lib.voxel_pool(x.detach(), x.stride(), geometry.detach().contiguous(), f, n, d, h, w, c, engine.grid,
                             workspace=ws, tile_voxels=engine.pool_tile, flags=engine.pool_flags)
        ctx.lib, ctx.dims = lib, (f, n, d, h, w, c)
        ctx.save_for_backward(ws[:f * n * d * h * w])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (rank,) = ctx.saved_tensors
        f, n, d, h, w, c = ctx.dims
        # the gradient is laid out like the encoder's output, (F, n, C, D, h, w): the 16-byte streaming-store path
        gx = torch.empty(f, n, c, d, h, w, dtype=torch.float32, device=grad_out.device).permute(0, 1, 3, 4, 5, 2)
        ctx.lib.voxel_pool_bwd(grad_out.float().contiguous(), rank, f, n, d, h, w, c, gx)
        return gx, None, None


class LiftSplat(torch.autograd.Function):
    """Depth softmax + fused lift (x) splat, differentiable in the depth logits and the features (the autograd path of
    fiery/models/encoder.py:99-100 followed by fiery/models/fiery.py:221-273) without the (n, C, D, h, w) outer
    product or its gradient ever existing."""

    @staticmethod
    def forward(ctx, depth_logits, features, geometry, engine):
        f, n, d, h, w = depth_logits.shape
        c = features.shape[2]
        lib = engine.lib
        prob = lib.depth_softmax(depth_logits.detach().reshape(f * n, d, h, w).contiguous())
        feats = features.detach().contiguous()
        ws = lib.pool_workspace(f, n, d, h, w, feats.device, engine.grid, engine.pool_tile, engine.pool_flags)
        out = lib.lift_splat(prob, feats, geometry.detach().contiguous(), f, n, d, h, w, c, engine.grid, workspace=ws,
                             tile_voxels=engine.pool_tile, flags=engine.pool_flags)
        ctx.lib, ctx.dims = lib, (f, n, d, h, w, c)
        ctx.save_for_backward(ws[:f * n * d * h * w], prob, feats)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        rank, prob, feats = ctx.saved_tensors
        f, n, d, h, w, c = ctx.dims
        want_depth, want_feat = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gd, gf = ctx.lib.lift_splat_bwd(grad_out.float().contiguous(), rank, prob, feats, f, n, d, h, w, c,
                                        want_depth=want_depth, want_features=want_feat)
        glogits = ctx.lib.depth_softmax_bwd(prob, gd).view(f, n, d, h, w) if want_depth else None
        return glogits, gf, None, None
