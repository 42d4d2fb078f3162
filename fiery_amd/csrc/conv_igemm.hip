// C ABI of the implicit-GEMM convolution: weight packing, descriptor validation, tile choice and launch
// (kernel: conv_igemm_kernel.h, one translation unit per tile shape), and the final 1x1 heads.
#include "conv_igemm_kernel.h"

namespace fiery {
namespace {

__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ void k_pack_weights(const float* __restrict__ w, int cout, int cin_total, int taps, ChanInverse inv,
                               int cin_units, int bn, int k_chunks, long long total, float* __restrict__ packed) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    // inside a (chunk, cout tile) block the image is [k / 4][cout][k % 4]: a lane's four k of one MFMA
    // group are one 16-byte LDS read
    const int kj = static_cast<int>(i & 3);
    const int nn = static_cast<int>((i >> 2) % bn);
    long long r = (i >> 2) / bn;
    const int kk = static_cast<int>(r % (BK / 4)) * 4 + kj;
    r /= BK / 4;
    const int chunk = static_cast<int>(r % k_chunks);
    const int tile = static_cast<int>(r / k_chunks);
    const int n = tile * bn + nn;
    const int k = chunk * BK + kk;
    const int u = k >> 3, ch = k & 7;
    const int tap = u / cin_units, cc = u - tap * cin_units;
    float v = 0.f;
    if (n < cout && tap < taps) {
        const int ci = inv.ci[cc * 8 + ch];
        if (ci >= 0) v = w[(static_cast<long long>(n) * cin_total + ci) * taps + tap];
    }
    packed[i] = v;
}

// bf16 image of the weights: inside a (chunk, cout tile) block [k / 8][cout][k % 8] - a lane's eight k of one
// v_mfma_f32_32x32x16_bf16 are one 16-byte LDS read
__global__ void k_pack_weights_bf16(const float* __restrict__ w, int cout, int cin_total, int taps, ChanInverse inv,
                                    int cin_units, int bn, int k_chunks, long long total, unsigned short* __restrict__ packed) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kj = static_cast<int>(i & 7);
    const int nn = static_cast<int>((i >> 3) % bn);
    long long r = (i >> 3) / bn;
    const int kk = static_cast<int>(r % (BK / 8)) * 8 + kj;
    r /= BK / 8;
    const int chunk = static_cast<int>(r % k_chunks);
    const int tile = static_cast<int>(r / k_chunks);
    const int n = tile * bn + nn;
    const int k = chunk * BK + kk;
    const int u = k >> 3, ch = k & 7;
    const int tap = u / cin_units, cc = u - tap * cin_units;
    float v = 0.f;
    if (n < cout && tap < taps) {
        const int ci = inv.ci[cc * 8 + ch];
        if (ci >= 0) v = w[(static_cast<long long>(n) * cin_total + ci) * taps + tap];
    }
    packed[i] = bf16_bits(v);
}

// split image of the weights (three bf16 terms per value, x = t1 + t2 + t3 exactly): inside a (chunk, cout tile) block
// [term][k / 8][cout][k % 8]
__global__ void k_pack_weights_split(const float* __restrict__ w, int cout, int cin_total, int taps, ChanInverse inv,
                                     int cin_units, int bn, int k_chunks, long long total, unsigned short* __restrict__ packed) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int kj = static_cast<int>(i & 7);
    const int nn = static_cast<int>((i >> 3) % bn);
    long long r = (i >> 3) / bn;
    const int kk = static_cast<int>(r % (BK / 8)) * 8 + kj;
    r /= BK / 8;
    const int term = static_cast<int>(r % 3);
    r /= 3;
    const int chunk = static_cast<int>(r % k_chunks);
    const int tile = static_cast<int>(r / k_chunks);
    const int n = tile * bn + nn;
    const int k = chunk * BK + kk;
    const int u = k >> 3, ch = k & 7;
    const int tap = u / cin_units, cc = u - tap * cin_units;
    float v = 0.f;
    if (n < cout && tap < taps) {
        const int ci = inv.ci[cc * 8 + ch];
        if (ci >= 0) v = w[(static_cast<long long>(n) * cin_total + ci) * taps + tap];
    }
    const float t1 = bf16_round(v), r1 = v - t1, t2 = bf16_round(r1), r2 = r1 - t2;
    packed[i] = bf16_bits(term == 0 ? t1 : term == 1 ? t2 : r2);
}

struct Geometry {
    int cout_pad, bn, n_tiles, k_chunks, n_units;
};

// cout tile width of a layer: the widest of 128 / 64 / 32 that divides its padded couts.  (FIERY_CONV_BN128_AS_64=1, an
// experiment switch read by packing and launch alike: exactly-128-cout layers as two 64-wide tiles - 3,750 tiles of 64 x 64 on
// 1,024 slots instead of 1,875 of 64 x 128 on 768 for a 120,000-pixel map.)
int conv_bn(int cout_pad) {
    static const bool narrow128 = getenv("FIERY_CONV_BN128_AS_64") != nullptr;
    if (cout_pad == 128 && narrow128) return 64;
    return (cout_pad % 128 == 0) ? 128 : (cout_pad % 64 == 0) ? 64 : 32;
}

Geometry conv_geometry(int cout, int cin_units, int taps) {
    Geometry g;
    g.cout_pad = (cout + 31) / 32 * 32;
    g.bn = conv_bn(g.cout_pad);
    g.n_tiles = g.cout_pad / g.bn;
    g.n_units = cin_units * taps;
    g.k_chunks = (g.n_units + 3) / 4;
    return g;
}

// ---- final 1x1 heads, NCHW result ------------------------------------------------------------------
struct HeadMeta {
    int c_off[8];
    unsigned char sigmoid[8];
};

// 64 pixels per workgroup.  The pixel-major tile is staged once through LDS (unit-stride reads), then
// each lane owns a pixel and each wavefront a subset of the outputs, so the NCHW planes are written
// with unit stride too.
__global__ __launch_bounds__(256) void k_heads_1x1(const float* __restrict__ in, int in_ld, int HW, int C, int head_c,
                                                   int n_out, const float* __restrict__ w, const float* __restrict__ bias,
                                                   HeadMeta meta, float* __restrict__ out) {
    HIP_DYNAMIC_SHARED(float, tile)   // [64][C + 1]
    const int img = blockIdx.y;
    const int p0 = blockIdx.x * 64;
    const int npx = min(64, HW - p0);
    const int row = C + 1;
    const float* src = in + (static_cast<long long>(img) * HW + p0) * in_ld;
    for (int i = threadIdx.x; i < npx * C; i += blockDim.x) {
        const int px = i / C, c = i - px * C;
        tile[px * row + c] = src[static_cast<long long>(px) * in_ld + c];
    }
    __syncthreads();
    const int px = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (px >= npx) return;
    for (int o = wave; o < n_out; o += 4) {
        const float* wr = w + o * head_c;
        const float* t = tile + px * row + meta.c_off[o];
        float acc = bias[o];
        for (int c = 0; c < head_c; ++c) acc = fmaf(wr[c], t[c], acc);
        if (meta.sigmoid[o]) acc = sigmoidf(acc);
        out[(static_cast<long long>(img) * n_out + o) * HW + p0 + px] = acc;
    }
}

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" size_t fiery_conv_packed_floats(int cout, int cin_units, int taps) {
    if (cout <= 0 || cin_units <= 0 || taps <= 0) return 0;
    const Geometry g = conv_geometry(cout, cin_units, taps);
    return static_cast<size_t>(g.n_tiles) * g.k_chunks * BK * g.bn;
}

extern "C" int fiery_conv_pack_weights(const float* w, int cout, int cin_total, int taps, const int32_t* chan_map,
                                       int cin_units, float* packed, fiery_stream_t stream) {
    FIERY_REQUIRE(w && chan_map && packed, "conv_pack_weights: null pointer");
    FIERY_REQUIRE(cout > 0 && cin_total > 0 && taps > 0 && cin_units > 0, "conv_pack_weights: bad shape");
    FIERY_REQUIRE(cin_units <= kMaxPackUnits, "conv_pack_weights: at most %d input channels", kMaxPackUnits * 8);
    ChanInverse inv;
    for (int i = 0; i < kMaxPackUnits * 8; ++i) inv.ci[i] = -1;
    for (int ci = 0; ci < cin_total; ++ci) {
        const int pos = chan_map[ci];
        FIERY_REQUIRE(pos >= 0 && pos < cin_units * 8, "conv_pack_weights: chan_map[%d] = %d out of range", ci, pos);
        FIERY_REQUIRE(inv.ci[pos] < 0, "conv_pack_weights: chan_map maps two channels to position %d", pos);
        inv.ci[pos] = static_cast<short>(ci);
    }
    const Geometry g = conv_geometry(cout, cin_units, taps);
    const long long total = static_cast<long long>(g.n_tiles) * g.k_chunks * BK * g.bn;
    hipLaunchKernelGGL(k_pack_weights, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w, cout, cin_total,
                       taps, inv, cin_units, g.bn, g.k_chunks, total, packed);
    return check_launch("conv_pack_weights");
}

extern "C" int fiery_conv_pack_weights_bf16(const float* w, int cout, int cin_total, int taps, const int32_t* chan_map,
                                            int cin_units, void* packed, fiery_stream_t stream) {
    FIERY_REQUIRE(w && chan_map && packed, "conv_pack_weights_bf16: null pointer");
    FIERY_REQUIRE(cout > 0 && cin_total > 0 && taps > 0 && cin_units > 0, "conv_pack_weights_bf16: bad shape");
    FIERY_REQUIRE(cin_units <= kMaxPackUnits, "conv_pack_weights_bf16: at most %d input channels", kMaxPackUnits * 8);
    ChanInverse inv;
    for (int i = 0; i < kMaxPackUnits * 8; ++i) inv.ci[i] = -1;
    for (int ci = 0; ci < cin_total; ++ci) {
        const int pos = chan_map[ci];
        FIERY_REQUIRE(pos >= 0 && pos < cin_units * 8, "conv_pack_weights_bf16: chan_map[%d] = %d out of range", ci, pos);
        FIERY_REQUIRE(inv.ci[pos] < 0, "conv_pack_weights_bf16: chan_map maps two channels to position %d", pos);
        inv.ci[pos] = static_cast<short>(ci);
    }
    const Geometry g = conv_geometry(cout, cin_units, taps);
    const long long total = static_cast<long long>(g.n_tiles) * g.k_chunks * BK * g.bn;
    hipLaunchKernelGGL(k_pack_weights_bf16, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w, cout, cin_total,
                       taps, inv, cin_units, g.bn, g.k_chunks, total, static_cast<unsigned short*>(packed));
    return check_launch("conv_pack_weights_bf16");
}

extern "C" int fiery_conv_pack_weights_split(const float* w, int cout, int cin_total, int taps, const int32_t* chan_map,
                                             int cin_units, void* packed, fiery_stream_t stream) {
    FIERY_REQUIRE(w && chan_map && packed, "conv_pack_weights_split: null pointer");
    FIERY_REQUIRE(cout > 0 && cin_total > 0 && taps > 0 && cin_units > 0, "conv_pack_weights_split: bad shape");
    FIERY_REQUIRE(cin_units <= kMaxPackUnits, "conv_pack_weights_split: at most %d input channels", kMaxPackUnits * 8);
    ChanInverse inv;
    for (int i = 0; i < kMaxPackUnits * 8; ++i) inv.ci[i] = -1;
    for (int ci = 0; ci < cin_total; ++ci) {
        const int pos = chan_map[ci];
        FIERY_REQUIRE(pos >= 0 && pos < cin_units * 8, "conv_pack_weights_split: chan_map[%d] = %d out of range", ci, pos);
        FIERY_REQUIRE(inv.ci[pos] < 0, "conv_pack_weights_split: chan_map maps two channels to position %d", pos);
        inv.ci[pos] = static_cast<short>(ci);
    }
    const Geometry g = conv_geometry(cout, cin_units, taps);
    const long long total = static_cast<long long>(g.n_tiles) * g.k_chunks * 3 * BK * g.bn;
    hipLaunchKernelGGL(k_pack_weights_split, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), w, cout, cin_total,
                       taps, inv, cin_units, g.bn, g.k_chunks, total, static_cast<unsigned short*>(packed));
    return check_launch("conv_pack_weights_split");
}

extern "C" size_t fiery_conv_winograd_packed_floats(int cout, int cin_units) {
    if (cout <= 0 || cin_units <= 0) return 0;
    return conv_winograd_packed_floats(cout, cin_units);
}

extern "C" int fiery_conv_pack_weights_winograd(const float* w, int cout, int cin_total, const int32_t* chan_map, int cin_units,
                                                float* packed, fiery_stream_t stream) {
    FIERY_REQUIRE(w && chan_map && packed, "conv_pack_weights_winograd: null pointer");
    FIERY_REQUIRE(cout > 0 && cin_total > 0 && cin_units > 0 && cin_units % 2 == 0, "conv_pack_weights_winograd: bad shape (whole 16-channel stages)");
    FIERY_REQUIRE(cin_units <= kMaxPackUnits, "conv_pack_weights_winograd: at most %d input channels", kMaxPackUnits * 8);
    ChanInverse inv;
    for (int i = 0; i < kMaxPackUnits * 8; ++i) inv.ci[i] = -1;
    for (int ci = 0; ci < cin_total; ++ci) {
        const int pos = chan_map[ci];
        FIERY_REQUIRE(pos >= 0 && pos < cin_units * 8, "conv_pack_weights_winograd: chan_map[%d] = %d out of range", ci, pos);
        FIERY_REQUIRE(inv.ci[pos] < 0, "conv_pack_weights_winograd: chan_map maps two channels to position %d", pos);
        inv.ci[pos] = static_cast<short>(ci);
    }
    if (conv_winograd_pack(w, cout, cin_total, inv, cin_units, packed, as_stream(stream)) != 0) return fail(FIERY_EINVAL, "conv_pack_weights_winograd: launch failed");
    return check_launch("conv_pack_weights_winograd");
}

extern "C" size_t fiery_conv_winograd_split_packed_floats(int cout, int cin_units) {
    if (cout <= 0 || cin_units <= 0) return 0;
    return conv_winograd_split_packed_floats(cout, cin_units);
}

extern "C" int fiery_conv_pack_weights_winograd_split(const float* w, int cout, int cin_total, const int32_t* chan_map, int cin_units,
                                                      float* packed, fiery_stream_t stream) {
    FIERY_REQUIRE(w && chan_map && packed, "conv_pack_weights_winograd_split: null pointer");
    FIERY_REQUIRE(cout > 0 && cin_total > 0 && cin_units > 0 && cin_units % 2 == 0, "conv_pack_weights_winograd_split: bad shape (whole 16-channel stages)");
    FIERY_REQUIRE(cin_units <= kMaxPackUnits, "conv_pack_weights_winograd_split: at most %d input channels", kMaxPackUnits * 8);
    ChanInverse inv;
    for (int i = 0; i < kMaxPackUnits * 8; ++i) inv.ci[i] = -1;
    for (int ci = 0; ci < cin_total; ++ci) {
        const int pos = chan_map[ci];
        FIERY_REQUIRE(pos >= 0 && pos < cin_units * 8, "conv_pack_weights_winograd_split: chan_map[%d] = %d out of range", ci, pos);
        FIERY_REQUIRE(inv.ci[pos] < 0, "conv_pack_weights_winograd_split: chan_map maps two channels to position %d", pos);
        inv.ci[pos] = static_cast<short>(ci);
    }
    if (conv_winograd_split_pack(w, cout, cin_total, inv, cin_units, packed, as_stream(stream)) != 0)
        return fail(FIERY_EINVAL, "conv_pack_weights_winograd_split: launch failed");
    return check_launch("conv_pack_weights_winograd_split");
}

namespace {
// does the bf16 form take this launch?  (the scalar-addressed loop's conditions, a tile shape that has a bf16 kernel)
bool conv_takes_bf16(const fiery_conv_desc* d, bool aligned, int bm, int bn, int cin_units) {
    if (d->precision != FIERY_PRECISION_BF16 || !d->weights_bf16 || !aligned || cin_units < 4) return false;
    if (getenv("FIERY_CONV_CLKPROBE") || getenv("FIERY_CONV_PRIO")) return false;
    return (bm == 128 && (bn == 32 || bn == 64 || bn == 128)) || (bm == 64 && (bn == 64 || bn == 128));
}
struct StreamKPlan {
    long long workspace_bytes = 0;
    int n_counters = 0, n_workgroups = 0;
};
enum ConvRunMode { kRunLaunch, kRunPrecision, kRunStreamKPlan, kRunForm };
int conv_run(const fiery_conv_desc* d, fiery_stream_t stream, ConvRunMode mode, StreamKPlan* plan = nullptr);
}  // namespace

extern "C" int fiery_conv_fwd(const fiery_conv_desc* d, fiery_stream_t stream) { return conv_run(d, stream, kRunLaunch); }

extern "C" int fiery_conv_precision_used(const fiery_conv_desc* d) { return conv_run(d, nullptr, kRunPrecision); }

extern "C" int fiery_conv_form_used(const fiery_conv_desc* d) { return conv_run(d, nullptr, kRunForm); }

extern "C" int fiery_conv_stream_k_plan(const fiery_conv_desc* d, int64_t* workspace_bytes, int32_t* n_counters, int32_t* n_workgroups) {
    FIERY_REQUIRE(workspace_bytes && n_counters && n_workgroups, "conv_stream_k_plan: null pointer");
    StreamKPlan plan;
    const int rc = conv_run(d, nullptr, kRunStreamKPlan, &plan);
    if (rc != FIERY_OK) return rc;
    *workspace_bytes = plan.workspace_bytes;
    *n_counters = plan.n_counters;
    *n_workgroups = plan.n_workgroups;
    return FIERY_OK;
}

namespace {
// validates the descriptor, plans the launch; kRunPrecision: returns the precision the launch would run in; kRunStreamKPlan:
// fills *plan with what the stream-K form of the launch needs (n_workgroups = 0: not covered)
int conv_run(const fiery_conv_desc* d, fiery_stream_t stream, ConvRunMode mode, StreamKPlan* plan) {
    const bool launch = mode == kRunLaunch;
    FIERY_REQUIRE(d, "conv_fwd: null descriptor");
    FIERY_REQUIRE(d->src[0].ptr && d->src[0].units > 0, "conv_fwd: source 0 missing");
    FIERY_REQUIRE(d->src[1].units == 0 || d->src[1].ptr, "conv_fwd: source 1 missing");
    FIERY_REQUIRE(d->weights && d->scale && d->shift && (d->out.ptr || d->epi == FIERY_EPI_HEADS), "conv_fwd: null pointer");
    FIERY_REQUIRE(d->kT >= 1 && d->kH >= 1 && d->kW >= 1 && d->stride >= 1, "conv_fwd: bad kernel shape");
    FIERY_REQUIRE(d->n_img_out > 0 && d->T_out > 0 && d->n_img_out % d->T_out == 0, "conv_fwd: bad image counts");
    FIERY_REQUIRE(d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0, "conv_fwd: bad spatial shape");
    FIERY_REQUIRE(d->cout_pad > 0 && d->cout_pad % 32 == 0, "conv_fwd: cout_pad must be a multiple of 32");
    FIERY_REQUIRE(d->act != FIERY_ACT_SWISH || (d->epi == FIERY_EPI_PLAIN && !d->weights2),
                  "conv_fwd: the swish activation exists for the plain, unchained epilogue only");
    FIERY_REQUIRE(d->act >= FIERY_ACT_NONE && d->act <= FIERY_ACT_SWISH, "conv_fwd: unknown activation %d", d->act);
    FIERY_REQUIRE(d->cout_store > 0 && (d->weights2 || d->cout_store <= d->cout_pad), "conv_fwd: bad cout_store");
    FIERY_REQUIRE(static_cast<long long>(d->n_img_out) * d->Hout * d->Wout < (1ll << 31) - 256, "conv_fwd: more than 2^31 output pixels");
    for (int s = 0; s < 2; ++s) {
        if (d->src[s].units == 0) continue;
        FIERY_REQUIRE(aligned16(d->src[s].ptr) && d->src[s].ld % 4 == 0 && d->src[s].batch_stride % 4 == 0 &&
                          d->src[s].time_stride % 4 == 0,
                      "conv_fwd: source %d must be 16-byte aligned with strides in multiples of 4 floats", s);
        FIERY_REQUIRE(d->src[s].ld >= d->src[s].units * 8, "conv_fwd: source %d narrower than its channel units", s);
    }
    const int cin_units = d->src[0].units + d->src[1].units;
    FIERY_REQUIRE(cin_units <= kMaxPackUnits, "conv_fwd: too many input channels");
    for (int s = 0; s < 2; ++s) {
        if (d->src[s].units == 0) continue;
        // the kernel addresses each source with 32-bit element offsets
        const long long n_batch = d->n_img_out / d->T_out;
        const long long span = (n_batch - 1) * llabs(d->src[s].batch_stride) +
                               (static_cast<long long>(d->T_out) + d->kT + llabs(static_cast<long long>(d->t_in_add))) * llabs(d->src[s].time_stride) +
                               (static_cast<long long>(d->Hin) + d->kH) * (d->Win + d->kW) * d->src[s].ld;
        FIERY_REQUIRE(span < (1ll << 31), "conv_fwd: source %d spans more than 2^31 floats", s);
    }
    if (d->epi == FIERY_EPI_GRU_GATES) {
        FIERY_REQUIRE(d->out2.ptr && d->aux0.ptr, "conv_fwd: GRU gate epilogue needs out2 and aux0");
    } else if (d->epi == FIERY_EPI_GRU_OUT) {
        FIERY_REQUIRE(d->aux0.ptr && d->aux1.ptr, "conv_fwd: GRU output epilogue needs aux0 and aux1");
    } else {
        FIERY_REQUIRE(d->epi == FIERY_EPI_PLAIN || d->epi == FIERY_EPI_HEADS, "conv_fwd: unknown epilogue %d", d->epi);
    }
    const int taps = d->kT * d->kH * d->kW;
    ConvP p;
    p.sk_tiles = 0;
    p.sk_ws = nullptr;
    p.sk_cnt = nullptr;
    p.wmg_img = p.wmg_tw = 0;
    p.wsh_img = p.wsh_tw = 0;
    for (int s = 0; s < 2; ++s) {
        // bytes from ptr to the end of the last row the launch may read: last batch element, last frame (output frame
        // T_out - 1 reads source frame T_out - 1 + t_in_add at its latest tap), last pixel, the source's channel units.
        // Meaningful (and used) for the scalar-addressed loop only, which requires non-negative strides and < 2^29 floats.
        const long long n_batch = d->n_img_out / d->T_out;
        const long long ext = (n_batch - 1) * d->src[s].batch_stride +
                              (static_cast<long long>(d->T_out) - 1 + d->t_in_add) * d->src[s].time_stride +
                              (static_cast<long long>(d->Hin) * d->Win - 1) * d->src[s].ld + d->src[s].units * 8;
        const int ext_bytes = (d->src[s].units > 0 && ext > 0 && ext < (1ll << 29)) ? static_cast<int>(4 * ext) : 0;
        p.src[s] = SrcP{d->src[s].ptr, d->src[s].ld, d->src[s].units, d->src[s].batch_stride, d->src[s].time_stride, ext_bytes};
    }
    p.Hin = d->Hin; p.Win = d->Win; p.Hout = d->Hout; p.Wout = d->Wout;
    {
        // (m, s) with g / div == (g * m) >> s for every 0 <= g < 2^31: m = floor(2^(31 + l) / div) + 1, l = ceil(log2 div)
        auto conv_magic = [](long long div, unsigned* m, int* sh) {
            int l = 0;
            while ((1ll << l) < div) ++l;
            *m = static_cast<unsigned>((1ull << (31 + l)) / static_cast<unsigned long long>(div) + 1ull);
            *sh = 31 + l;
        };
        conv_magic(static_cast<long long>(d->Hout) * d->Wout, &p.mg_hw, &p.sh_hw);
        conv_magic(d->Wout, &p.mg_w, &p.sh_w);
        conv_magic(d->T_out, &p.mg_t, &p.sh_t);
    }
    p.n_img = d->n_img_out; p.Tout = d->T_out; p.tout0 = d->t_out0; p.tinadd = d->t_in_add;
    p.kT = d->kT; p.kH = d->kH; p.kW = d->kW; p.stride = d->stride; p.padH = d->padH; p.padW = d->padW;
    p.w = d->weights;
    p.cout_pad = d->cout_pad;
    p.cin_units = cin_units;
    p.n_units = cin_units * taps;
    p.k_chunks = (p.n_units + 3) / 4;
    p.scale = d->scale; p.shift = d->shift; p.img_bias = d->img_bias;
    p.bias_border = d->img_bias_border != 0;
    p.heads.w = nullptr;
    p.heads.bias = nullptr;
    p.heads.n_out = 0;
    if (d->epi == FIERY_EPI_HEADS) {
        FIERY_REQUIRE(d->cout_pad % 128 == 0 && !d->weights2 && !d->res.ptr, "conv_fwd: heads epilogue needs cout_pad % 128 == 0, no chain, no residual");
        FIERY_REQUIRE(d->heads.w && d->heads.bias && d->heads.n_out > 0 && d->heads.n_out <= FIERY_MAX_HEAD_OUTPUTS,
                      "conv_fwd: heads epilogue needs 1..%d output rows", FIERY_MAX_HEAD_OUTPUTS);
        p.heads.w = d->heads.w;
        p.heads.bias = d->heads.bias;
        p.heads.n_out = d->heads.n_out;
        int per_tile[64] = {0};
        for (int o = 0; o < d->heads.n_out; ++o) {
            const int g = d->heads.group[o];
            FIERY_REQUIRE(g >= 0 && g * 64 < d->cout_pad && d->heads.out[o], "conv_fwd: heads output %d: bad group or null plane", o);
            FIERY_REQUIRE(++per_tile[(g >> 1) & 63] <= 4, "conv_fwd: more than four head outputs read one 128-channel tile");
            p.heads.group[o] = g;
            p.heads.sigmoid[o] = d->heads.sigmoid[o];
            p.heads.out[o] = d->heads.out[o];
            p.heads.istride[o] = d->heads.img_stride[o];
        }
    }
    FIERY_REQUIRE(!p.bias_border || (d->img_bias && d->Hout >= 2 && d->Wout >= 2),
                  "conv_fwd: img_bias_border needs img_bias and an output of at least 2x2 pixels");
    p.act = d->act; p.epi = d->epi; p.res_pre = d->res_before_act;
    p.res = TensP{d->res.ptr, d->res.ld, d->res.img_stride};
    p.out = TensP{d->out.ptr, d->out.ld, d->out.img_stride};
    p.out2 = TensP{d->out2.ptr, d->out2.ld, d->out2.img_stride};
    p.aux0 = TensP{d->aux0.ptr, d->aux0.ld, d->aux0.img_stride};
    p.aux1 = TensP{d->aux1.ptr, d->aux1.ld, d->aux1.img_stride};
    p.cout_store = d->cout_store;
    p.M = static_cast<long long>(d->n_img_out) * d->Hout * d->Wout;
    // 16-byte rows need 16-byte aligned bases and strides in multiples of 4 floats everywhere the epilogue touches
    auto rows_ok = [](const fiery_nhwc& t) {
        return !t.ptr || (aligned16(t.ptr) && t.ld % 4 == 0 && t.img_stride % 4 == 0);
    };
    p.vec_epilogue = (rows_ok(d->out) && rows_ok(d->res) && rows_ok(d->out2) && rows_ok(d->aux0) && rows_ok(d->aux1) &&
                      d->cout_store % 4 == 0 &&
                      (!d->img_bias || aligned16(d->img_bias)) && aligned16(d->scale) && aligned16(d->shift) &&
                      (!d->weights2 || (aligned16(d->scale2) && aligned16(d->shift2))))
                         ? 1 : 0;
    if (const char* forced = getenv("FIERY_CONV_VEC_EPILOGUE")) p.vec_epilogue = p.vec_epilogue && atoi(forced) != 0;
    // bit 1: every tensor of the epilogue is dense over its images and addressable with 31-bit byte offsets (the row epilogue
    // then walks buffer descriptors with one 32-bit add per row, conv_igemm_kernel.h store_rows)
    if (p.vec_epilogue) {
        const long long hw = static_cast<long long>(d->Hout) * d->Wout;
        auto dense = [&](const fiery_nhwc& t) {
            return !t.ptr || (t.img_stride == hw * t.ld && p.M * t.ld * 4 < (1ll << 31));
        };
        bool tensors_dense = dense(d->out) && dense(d->res) && dense(d->out2) && dense(d->aux0) && dense(d->aux1) && (!d->weights3 || dense(d->out3));
        if (const char* forced = getenv("FIERY_CONV_DENSE_EPILOGUE")) tensors_dense = tensors_dense && atoi(forced) != 0;     // A/B runs
        if (tensors_dense && !p.bias_border) p.vec_epilogue |= 2;
        // bit 2 (the Winograd kernels' lean epilogue): every tensor addressable with ONE 31-bit byte offset per thread -
        // image * img_stride + pixel * ld, images at any non-negative stride (the SpatialGRU's time-step views of sequence buffers)
        auto small = [&](const fiery_nhwc& t) {
            return !t.ptr || (t.img_stride >= 0 && ((static_cast<long long>(d->n_img_out) - 1) * t.img_stride + hw * t.ld) * 4 < (1ll << 31));
        };
        if (small(d->out) && small(d->res) && small(d->out2) && small(d->aux0) && small(d->aux1)) p.vec_epilogue |= 4;
    }
    p.w2 = d->weights2;
    p.scale2 = d->scale2;
    p.shift2 = d->shift2;
    p.act2 = d->act2;
    if (d->weights2) {
        FIERY_REQUIRE(d->cout_pad == 32 && d->epi == FIERY_EPI_PLAIN, "conv_fwd: a chained 1x1 needs cout_pad == 32 and the plain epilogue");
        FIERY_REQUIRE(d->scale2 && d->shift2 && aligned16(d->weights2), "conv_fwd: chained 1x1 operands missing or misaligned");
        FIERY_REQUIRE(d->cout_store <= 64, "conv_fwd: a chained 1x1 produces at most 64 channels");
        FIERY_REQUIRE(!d->res.ptr || !d->res_before_act, "conv_fwd: chained 1x1 adds the residual after the activation");
    }
    if (d->weights3) {
        FIERY_REQUIRE(d->weights2 && d->cout_store == 64 && (p.vec_epilogue & 1), "conv_fwd: the third stage needs the chained 1x1, 64 stored channels and 16-byte addressable tensors");
        FIERY_REQUIRE(d->scale3 && d->shift3 && d->out3.ptr && aligned16(d->weights3) && aligned16(d->scale3) && aligned16(d->shift3) &&
                          aligned16(d->out3.ptr) && d->out3.ld % 4 == 0 && d->out3.img_stride % 4 == 0 && d->out3.ld >= 32,
                      "conv_fwd: third-stage operands missing or misaligned");
        FIERY_REQUIRE(d->act3 == FIERY_ACT_NONE || d->act3 == FIERY_ACT_RELU, "conv_fwd: third-stage activation must be none or ReLU");
        FIERY_REQUIRE(!d->out2.ptr && !d->aux0.ptr && !d->aux1.ptr, "conv_fwd: the third stage and the GRU operands exclude each other");
        // the kernel's argument block stays as it is: the operands ride in members the chained mode does not use
        p.heads.w = d->weights3;
        p.heads.n_out = d->act3;
        p.aux0.ptr = const_cast<float*>(d->scale3);
        p.aux1.ptr = const_cast<float*>(d->shift3);
        p.out2 = TensP{d->out3.ptr, d->out3.ld, d->out3.img_stride};
    }
    const int bn = conv_bn(d->cout_pad);
    FIERY_REQUIRE(d->epi != FIERY_EPI_HEADS || bn == 128, "conv_fwd: the heads epilogue needs 128-wide cout tiles");
    const int n_tiles = d->cout_pad / bn;
    // Tile height: workgroups run in rounds of (256 CUs x resident workgroups per CU); the default picks the height
    // whose last round is better filled.  How a partly filled round really behaves depends on the launch (a lone
    // workgroup is latency-bound, not three times faster), so callers that repeat a launch can time both heights
    // and pass the winner in desc->tile_m (fiery_amd/ops.py does, once per shape).
    bool half_tiles = false;
    if (bn >= 64 && !d->weights2) {
        auto fill = [&](int bm, int per_cu) {
            const double tiles = static_cast<double>(ceil_div(p.M, bm)) * n_tiles;
            const double slots = 256.0 * per_cu;
            return tiles / (ceil(tiles / slots) * slots);
        };
        const double e128 = fill(128, bn == 128 ? 2 : 3);
        const double e64 = (bn == 128 ? 1.0 : 0.8) * fill(64, bn == 128 ? 3 : 5);
        half_tiles = bn == 128 || e64 > e128;
        if (d->tile_m == 64 || d->tile_m == 128) half_tiles = d->tile_m == 64;                   // the caller measured
        if (d->epi == FIERY_EPI_HEADS) half_tiles = true;          // its LDS plan (tile + 1x1 rows) is the 64-pixel one
        if (const char* forced = getenv("FIERY_CONV_TILE_M"))                                    // tuning / tests
            if (d->epi != FIERY_EPI_HEADS) half_tiles = atoi(forced) == 64;
    }
    if (cin_units < 4 && d->epi != FIERY_EPI_HEADS) half_tiles = false;      // that variant exists for 128-pixel tiles only
    dim3 grid(ceil_div(p.M, half_tiles ? 64 : 128), n_tiles);
    hipStream_t hs = as_stream(stream);
    unsigned long long* clk = nullptr;                                                          // tuning builds only
    if (const char* probe = getenv("FIERY_CONV_CLKPROBE")) clk = reinterpret_cast<unsigned long long*>(strtoull(probe, nullptr, 0));
    const bool prio = getenv("FIERY_CONV_PRIO") != nullptr;                                    // tuning experiment
    // The scalar-addressed K loop (ALIGNED): every tap must hold a whole number of 32-channel stages from one source,
    // and each source must be addressable with non-negative 31-bit byte offsets from (a little before) its base.
    bool aligned = cin_units % 4 == 0 && d->src[0].units % 4 == 0 && d->kT <= 8 && d->kH <= 8 && d->kW <= 8 && d->t_in_add >= 0;
    for (int s = 0; s < 2 && aligned; ++s) {
        if (d->src[s].units == 0) continue;
        const long long n_batch = d->n_img_out / d->T_out;
        const long long span = (n_batch - 1) * d->src[s].batch_stride +
                               (static_cast<long long>(d->T_out) + d->kT + d->t_in_add) * d->src[s].time_stride +
                               (static_cast<long long>(d->Hin) + d->kH) * (d->Win + d->kW) * d->src[s].ld;
        // (the descriptor size of the scalar-addressed and halo loops is p.src[s].ext_bytes, computed above from the exact
        // extents; 0 there means "out of range" and would make every load return zeros - tie the two conditions together)
        aligned = d->src[s].batch_stride >= 0 && d->src[s].time_stride >= 0 && span < (1ll << 29) && p.src[s].ext_bytes > 0;
    }
    if (const char* forced = getenv("FIERY_CONV_ALIGNED")) aligned = aligned && atoi(forced) != 0;       // tuning / tests
    int variant = aligned ? kConvAligned : kConvGeneric;
    if (cin_units < 4) variant = kConvSmallCin;       // the loop whose unit advance may carry several times per stage
    else if (clk) variant = aligned ? kConvClockAligned : kConvClock;
    else if (prio) variant = kConvPrio;
    int bm = half_tiles ? 64 : 128;
    // Stream-K form (conv_igemm_kernel.h, k_conv_igemm<SK>): fp32, scalar-addressed loop, 128-pixel tiles of 64 or 128 couts, no
    // chained 1x1, no heads; one round of workgroups - as many as the chip holds - each with at least eight chunks of work
    StreamKPlan sk;
    if (variant == kConvAligned && (bn == 64 || bn == 128) && !d->weights2 && d->epi != FIERY_EPI_HEADS &&
        !(d->precision == FIERY_PRECISION_BF16 && conv_takes_bf16(d, aligned, 128, bn, cin_units))) {
        static const int n_cu = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            return n;
        }();
        const long long tiles = static_cast<long long>(ceil_div(p.M, 128)) * n_tiles;
        const long long total = tiles * p.k_chunks;
        long long nwg = static_cast<long long>(n_cu) * conv_stream_k_per_cu(bn);
        if (const char* forced = getenv("FIERY_CONV_SK_WGS")) nwg = atoll(forced);                  // tuning / tests
        if (nwg > total / 8) nwg = total / 8;
        nwg &= ~7ll;
        if (nwg >= 8 && tiles < (1ll << 30)) {
            sk.n_workgroups = static_cast<int>(nwg);
            sk.n_counters = static_cast<int>(tiles);
            sk.workspace_bytes = nwg * 2 * 128 * bn * 4;
        }
    }
    if (mode == kRunStreamKPlan) {
        *plan = sk;
        return FIERY_OK;
    }
    bool stream_k = d->stream_k != 0 && sk.n_workgroups > 0;
    if (const char* forced = getenv("FIERY_CONV_STREAM_K")) stream_k = sk.n_workgroups > 0 && atoi(forced) != 0 && d->sk_workspace != nullptr;   // tuning / tests
    if (stream_k) {
        FIERY_REQUIRE(d->sk_workspace && d->sk_counters && d->sk_workspace_bytes >= sk.workspace_bytes && d->sk_counters_len >= sk.n_counters &&
                          aligned16(d->sk_workspace),
                      "conv_fwd: stream-K needs %lld workspace bytes and %d counters (fiery_conv_stream_k_plan)", sk.workspace_bytes, sk.n_counters);
        bm = 128;
    }
    // Winograd F(2x2, 3x3) form (conv_winograd.hip): the caller packed the transformed weights and asks for it; taken by fp32
    // launches of 3 x 3 / stride 1 / 'same' layers with whole 16-channel stages per source, 64-cout tiles, 16-byte addressable
    // tensors, the plain, GRU or heads epilogues - everything else ignores the request
    const bool winograd = d->winograd != 0 && d->weights_winograd && variant == kConvAligned && d->kT == 1 && d->kH == 3 && d->kW == 3 &&
                          d->stride == 1 && d->padH == 1 && d->padW == 1 && d->Hin == d->Hout && d->Win == d->Wout && d->cout_pad % 64 == 0 &&
                          cin_units % 2 == 0 && d->src[0].units % 2 == 0 && !d->weights2 && (p.vec_epilogue & 1) &&
                          d->precision != FIERY_PRECISION_BF16 && aligned16(d->weights_winograd) &&
                          static_cast<long long>(d->n_img_out) * ((d->Hout + 1) / 2) * ((d->Wout + 1) / 2) < (1ll << 30);
    const bool bf16 = !stream_k && !winograd && conv_takes_bf16(d, aligned, bm, bn, cin_units);
    // split form of the scalar-addressed loop (FIERY_PRECISION_F32_SPLIT with the image of fiery_conv_pack_weights_split in
    // weights_bf16): 128-pixel tiles of 32 (the chained tails too) or 64 couts; everything else runs the fp32 kernels
    const bool split_tile = !stream_k && !winograd && d->precision == FIERY_PRECISION_F32_SPLIT && d->weights_bf16 && aligned && cin_units >= 4 &&
                            bm == 128 && (bn == 32 || bn == 64) && d->epi != FIERY_EPI_HEADS && !getenv("FIERY_CONV_CLKPROBE") && !getenv("FIERY_CONV_PRIO") &&
                            (!d->weights2 || (d->weights2_split && aligned16(d->weights2_split))) &&
                            (!d->weights3 || (d->weights3_split && aligned16(d->weights3_split)));
    const bool split = winograd && d->winograd == FIERY_WINOGRAD_SPLIT_TERMS;      // weights_winograd is the split image then
    if (mode == kRunForm) return winograd ? (split ? FIERY_CONV_FORM_WINOGRAD_SPLIT : FIERY_CONV_FORM_WINOGRAD) : (stream_k ? FIERY_CONV_FORM_STREAM_K : FIERY_CONV_FORM_TILE);
    if (!launch) return bf16 ? FIERY_PRECISION_BF16 : split_tile ? FIERY_PRECISION_F32_SPLIT : FIERY_PRECISION_F32;
    if (winograd) {
        {
            auto magic = [](long long div, unsigned* m, int* sh) {       // (as conv_magic above: exact for 0 <= g < 2^31)
                int l = 0;
                while ((1ll << l) < div) ++l;
                *m = static_cast<unsigned>((1ull << (31 + l)) / static_cast<unsigned long long>(div) + 1ull);
                *sh = 31 + l;
            };
            const int TH = (d->Hout + 1) / 2, TW = (d->Wout + 1) / 2;
            magic(static_cast<long long>(TH) * TW, &p.wmg_img, &p.wsh_img);
            magic(TW, &p.wmg_tw, &p.wsh_tw);
        }
        p.w = d->weights_winograd;
        if (const char* t = getenv("FIERY_WINOGRAD_TRACE")) p.sk_ws = reinterpret_cast<float*>(strtoull(t, nullptr, 0));      // tuning builds (W_TRACE)
        if (!(split ? conv_launch_winograd_split(p, hs) : conv_launch_winograd(p, hs))) return fail(FIERY_EINVAL, "conv_fwd: Winograd launch failed");
        return check_launch("conv_fwd (Winograd)");
    }
    if (stream_k) {
        p.sk_tiles = sk.n_counters;
        p.sk_ws = static_cast<float*>(d->sk_workspace);
        p.sk_cnt = d->sk_counters;
        p.tiles_m = 0;
        if (!conv_launch_stream_k(p, bn, dim3(sk.n_workgroups), hs)) return fail(FIERY_EINVAL, "conv_fwd: no stream-K kernel for %d-wide cout tiles", bn);
        return check_launch("conv_fwd (stream-K)");
    }
    if (split_tile) {
        FIERY_REQUIRE(aligned16(d->weights_bf16), "conv_fwd: split weights must be 16-byte aligned");
        p.w = static_cast<const float*>(d->weights_bf16);
        if (d->weights2) p.w2 = static_cast<const float*>(d->weights2_split);          // (the chained products run split too)
        if (d->weights3) p.heads.w = static_cast<const float*>(d->weights3_split);
        if (!conv_launch_split(p, bn, grid, hs)) return fail(FIERY_EINVAL, "conv_fwd: no split kernel for %d-wide cout tiles", bn);
        return check_launch("conv_fwd (split)");
    }
    if (bf16) {
        FIERY_REQUIRE(aligned16(d->weights_bf16), "conv_fwd: bf16 weights must be 16-byte aligned");
        p.w = static_cast<const float*>(d->weights_bf16);
        // 3 x 3 / stride 1 / 'same' layers with 64 couts or more take the halo loop (conv_igemm_kernel.h); FIERY_CONV_HALO=0: A/B runs
        bool halo = bn >= 64 && !d->weights2 && d->kT == 1 && d->kH == 3 && d->kW == 3 && d->stride == 1 && d->padH == 1 && d->padW == 1 &&
                    d->Hin == d->Hout && d->Win == d->Wout && p.M + 2ll * d->Wout + 2 < (1ll << 31);
        if (const char* forced = getenv("FIERY_CONV_HALO")) halo = halo && atoi(forced) != 0;
        if (!conv_launch_bf16(p, bm, bn, grid, hs, halo)) return fail(FIERY_EINVAL, "conv_fwd: no bf16 kernel for the %d x %d tile", bm, bn);
        return check_launch("conv_fwd (bf16)");
    }
    bool launched;
    // fp32 halo loop (FIERY_CONV_HALO_F32=1; A/B switch while it is being measured): the same layers as the bf16 one
    bool halo_f32 = false;
    if (const char* on = getenv("FIERY_CONV_HALO_F32"))
        halo_f32 = atoi(on) != 0 && variant == kConvAligned && bm == 64 && bn >= 64 && d->kT == 1 && d->kH == 3 && d->kW == 3 && d->stride == 1 &&
                   d->padH == 1 && d->padW == 1 && d->Hin == d->Hout && d->Win == d->Wout && p.M + 2ll * d->Wout + 2 < (1ll << 31);
    if (halo_f32) launched = conv_launch_f32_halo(p, bn, grid, hs);
    else if (half_tiles) launched = bn == 128 ? conv_launch_64x128(p, grid, hs, variant, clk) : conv_launch_64x64(p, grid, hs, variant, clk);
    else if (bn == 128) launched = conv_launch_128x128(p, grid, hs, variant, clk);
    else if (bn == 64) launched = conv_launch_128x64(p, grid, hs, variant, clk);
    else launched = conv_launch_128x32(p, grid, hs, variant, clk);
    if (!launched)
        return fail(FIERY_EINVAL, "conv_fwd: kernel variant %d of the %d x %d tile is not in this build (the probes need "
                                  "FIERY_CONV_TUNING=1 at build time)", variant, half_tiles ? 64 : 128, bn);
    return check_launch("conv_fwd");
}
}  // namespace

extern "C" int fiery_heads_1x1_nchw(const float* in, int in_ld, int n_img, int HW, int C, int head_c, int n_out,
                                    const float* w, const float* bias, const int32_t* c_off, const uint8_t* sigmoid,
                                    float* out, fiery_stream_t stream) {
    FIERY_REQUIRE(in && w && bias && c_off && sigmoid && out, "heads: null pointer");
    FIERY_REQUIRE(n_out > 0 && n_out <= 8 && n_img > 0 && HW > 0 && C > 0 && in_ld >= C, "heads: bad shape");
    FIERY_REQUIRE(head_c > 0 && head_c <= C, "heads: bad head width");
    FIERY_REQUIRE(static_cast<size_t>(64) * (C + 1) * sizeof(float) <= 160 * 1024, "heads: too many channels");
    HeadMeta meta;
    for (int o = 0; o < 8; ++o) {
        meta.c_off[o] = 0;
        meta.sigmoid[o] = 0;
    }
    for (int o = 0; o < n_out; ++o) {
        FIERY_REQUIRE(c_off[o] >= 0 && c_off[o] + head_c <= C, "heads: c_off[%d] out of range", o);
        meta.c_off[o] = c_off[o];
        meta.sigmoid[o] = sigmoid[o];
    }
    hipLaunchKernelGGL(k_heads_1x1, dim3(ceil_div(HW, 64), n_img), dim3(256), static_cast<size_t>(64) * (C + 1) * sizeof(float),
                       as_stream(stream), in, in_ld, HW, C, head_c, n_out, w, bias, meta, out);
    return check_launch("heads_1x1");
}
