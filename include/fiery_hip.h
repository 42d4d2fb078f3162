/*
 * fiery_hip.h - C ABI of libfiery_hip.so, the MI355X (gfx950) kernels behind the FIERY camera-to-BEV
 * hot path.
 *
 * The reference (wayveai/fiery) has no native layer: this path is Python calling ATen operators.  Each
 * entry point below replaces the ATen call sequence of the cited reference lines and is what a Python
 * (ctypes) binding for that path binds - see INTEGRATION.md for the binding a maintainer would add.
 *
 * Conventions (all functions):
 *   - plain pointers and sizes; device pointers unless the parameter is documented "host";
 *   - return 0 on success, a negative FIERY_E* code otherwise; fiery_last_error() (host string,
 *     thread-local) describes the last failure; nothing throws;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no call synchronises, allocates
 *     or frees device memory - the caller owns every buffer including workspaces, so calls are
 *     hipGraph-capturable and re-entrant across streams given distinct workspaces;
 *   - fp32 tensors; "NHWC" tensors are addressed as base + image*img_stride + (y*W + x)*ld + channel.
 */
#ifndef FIERY_HIP_H
#define FIERY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIERY_ABI_VERSION 25      /* 25 (round 6): fiery_conv_desc grew (weights2_split, weights3_split); 24 (round 6): the split Winograd form (fiery_conv_pack_weights_winograd_split, FIERY_CONV_FORM_WINOGRAD_SPLIT); 23 (round 6): fiery_conv_form_used, FIERY_POOL_NO_RANKS, the prepass quantises without divisions where that is exact; 22 (round 5): fiery_conv_desc grew (weights_winograd, winograd, stream_k, sk_*); fiery_conv_pack_weights_winograd, fiery_conv_winograd_packed_floats, fiery_conv_stream_k_plan */

#define FIERY_OK 0
#define FIERY_EINVAL (-22)      /* bad argument (shape, alignment, null pointer) */
#define FIERY_ENOMEM (-12)      /* workspace too small */
#define FIERY_ELAUNCH (-5)      /* the HIP runtime rejected the launch */

typedef void* fiery_stream_t;   /* hipStream_t */

int fiery_abi_version(void);
const char* fiery_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Lift: camera geometry            (reference: fiery/models/fiery.py:193-208 `get_geometry`)
 * ---------------------------------------------------------------------------------------------- */

/* Per camera: M = R . K^-1 (3x3) and t (3) -> cam[n][12] = {M row-major, t}.
 * Replaces `rotation.matmul(torch.inverse(intrinsics))`, fiery.py:195,203.
 * intrinsics [n][3][3], extrinsics [n][4][4] (camera->ego). */
int fiery_camera_matrices(const float* intrinsics, const float* extrinsics, int n_cameras,
                          float* cam, fiery_stream_t stream);

/* The same matrices served from a calibration table (round 4, ABI 21): `torch.inverse` in the reference's CPU path is LAPACK;
 * the closed form above equals it bit for bit for zero-skew pinhole intrinsics only.  The caller computes cam[12] for the
 * calibrations of its rig ON THE HOST with the reference's own operators and files them in `table` under the 21 words that enter
 * the computation (K's nine, then the three rows of [R | t]); this entry point looks every camera's words up (open addressing,
 * FNV-1a over the words folded once: h ^= h >> 15; at most FIERY_CALIB_PROBES slots from h & (table_slots - 1), an empty slot
 * ends the search) and copies the twelve numbers on a hit.  A miss evaluates the device form (as fiery_camera_matrices) and
 * appends the camera's words to `misses`, which every call rewrites:
 *   misses[0] entries written (<= miss_capacity), [1] misses of the call, [2] call number, [3] unused,
 *   then miss_capacity rows of FIERY_CALIB_MISS_WORDS words {21 key words, camera index, 2 unused}, then the call number again
 *   (a host that copies the list asynchronously compares the two call numbers to see that its copy is whole).
 * table: table_slots (a power of two >= 64) rows of FIERY_CALIB_ENTRY_WORDS words {occupied, 21 key words, 12 value words, 2 unused}.
 * No host interaction: capturable in a hipGraph.  The host side is fiery_amd/calibration.py. */
#define FIERY_CALIB_KEY_WORDS 21
#define FIERY_CALIB_ENTRY_WORDS 36
#define FIERY_CALIB_MISS_WORDS 24
#define FIERY_CALIB_MISS_HEADER 4
#define FIERY_CALIB_PROBES 16
int fiery_camera_matrices_cached(const float* intrinsics, const float* extrinsics, int n_cameras,
                                 const uint32_t* table, int table_slots, uint32_t* misses, int miss_capacity,
                                 float* cam, fiery_stream_t stream);

/* geometry[n][D][H][W][3] = M_n . (u*d, v*d, d) + t_n for every frustum point.
 * Replaces fiery.py:199-205.  frustum [D][H][W][3] = (u, v, depth) (fiery.py:109-128). */
int fiery_lift_geometry(const float* frustum, const float* cam, int n_cameras, int D, int H, int W,
                        float* geometry, fiery_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Splat: voxel pooling             (reference: fiery/models/fiery.py:221-273
 *                                   `projection_to_birds_eye_view` + fiery/utils/geometry.py:283-302
 *                                   `VoxelsSumming.forward`; LSS names voxel_pooling / QuickCumsum)
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    float origin[3];      /* bev_start_position - bev_resolution / 2, evaluated in fp32 (fiery.py:236) */
    float resolution[3];  /* bev_resolution   (geometry.py:53)                                        */
    int32_t dim[3];       /* bev_dimension    (geometry.py:55-56); dim[2] must be 1 for pooling       */
} fiery_bev_grid;

/* Integer path only: for each of n_points positions, the voxel index triple (trunc toward zero, as
 * `.long()`), and rank = ix*(Y*Z) + iy*Z + iz, or -1 when the point is outside the grid
 * (fiery.py:236-256).  idx may be NULL.  Bit-exact against the reference by construction. */
int fiery_voxel_index(const float* geometry, int64_t n_points, const fiery_bev_grid* grid /* host */,
                      int32_t* rank, int32_t* idx /* [n_points][3] or NULL */, fiery_stream_t stream);

#define FIERY_POOL_DETERMINISTIC 1u  /* flags: bit-reproducible sums (order-independent fixed point), slower */
/* flags: the caller vouches that the workspace is either a zero-filled allocation or was last used by a pooling call of
 * this library that returned FIERY_OK (every call leaves the region it clears as it found it): the call then skips its
 * memset dispatch.  Without the flag any workspace contents are accepted. */
#define FIERY_POOL_WORKSPACE_CLEAN 2u
/* flags: the caller does not need the voxel ranks (int32 [frames][n_cameras][D][H][W], -1 = outside the grid) that a
 * pooling call otherwise leaves at the head of its workspace for fiery_voxel_pool_bwd: an inference call in the
 * compact-plane form then does not write them (17 MB of the prepass's 27 MB of stores at baseline.yml; the other
 * forms read the ranks themselves and ignore the flag).  Results are unchanged. */
#define FIERY_POOL_NO_RANKS 4u

/* Scratch needed by fiery_voxel_pool_fwd / fiery_lift_splat_fwd for this problem (n_voxels = X*Y;
 * tile_voxels and flags as passed to the pooling call); 0 if the arguments are unusable. */
size_t fiery_voxel_pool_workspace_bytes(int frames, int n_cameras, int D, int H, int W, int n_voxels,
                                        int tile_voxels, uint32_t flags);

/* Byte offset, inside that workspace, of `frames` int32 counters: after a fiery_voxel_pool_fwd call that took the
 * compact-plane form, counter f holds the number of voxels of frame f that at least one point falls in.  A caller that
 * repeats a call on the same rig reads them once (after the stream has finished) and passes a little more than their
 * maximum as `tile_voxels`: the kernel's LDS plane then has one cell per occupied voxel and more workgroups share a CU.
 * Same arguments as fiery_voxel_pool_workspace_bytes; 0 if they are unusable. */
size_t fiery_voxel_pool_occupied_offset(int frames, int n_cameras, int D, int H, int W, int n_voxels,
                                        int tile_voxels, uint32_t flags);

/* out[f][c][ix][iy] = sum of x over the points of frame f that fall in voxel (ix, iy, 0); voxels no
 * point reaches are 0.
 *   x        : logical [frames][n_cameras][D][H][W][C]; element (f,n,d,h,w,c) lives at
 *              x + f*xs[0] + n*xs[1] + d*xs[2] + h*xs[3] + w*xs[4] + c*xs[5]   (xs = x_strides, host).
 *              The encoder's native layout (n, C, D, H, W) - the permuted view fiery.py:214-219
 *              returns - is the fast case (xs[4] == 1).
 *   geometry : [frames][n_cameras][D][H][W][3] contiguous
 *   out      : [frames][C][X][Y]
 *   tile_voxels : 0 = choose.  Compact-plane form (the default for 16-byte addressable rows with H <= 28 and
 *              W <= 64): number of LDS cells = occupied voxels the plane can hold in one pass (frames with more take
 *              several passes over their rows: slower, same result).  Other forms: voxels per LDS tile.
 * Kernel forms (chosen per call): compact plane (cells for occupied voxels only, two or three
 * workgroups per CU, any grid up to 2^24 voxels), tiled (several LDS tiles per plane with ordered work lists: the
 * bit-reproducible mode, the fused form, scalar strides), whole dense plane (FIERY_POOL_COMPACT=0).          */
int fiery_voxel_pool_fwd(const float* x, const int64_t* x_strides /* host [6] */, const float* geometry,
                         int frames, int n_cameras, int D, int H, int W, int C,
                         const fiery_bev_grid* grid /* host */, float* out,
                         void* workspace, size_t workspace_bytes, int tile_voxels, uint32_t flags,
                         fiery_stream_t stream);

/* Fused lift (x) splat: never materialises the (n, C, D, H, W) outer product.
 * out[f][c][ix][iy] = sum over points of depth_prob[f][n][d][h][w] * features[f][n][c][h][w]
 * (reference: fiery/models/encoder.py:99-100 followed by fiery.py:221-273).
 * depth_prob [frames][n_cameras][D][H][W] (already soft-maxed over D), features [frames][n_cameras][C][H][W]. */
int fiery_lift_splat_fwd(const float* depth_prob, const float* features, const float* geometry,
                         int frames, int n_cameras, int D, int H, int W, int C,
                         const fiery_bev_grid* grid /* host */, float* out,
                         void* workspace, size_t workspace_bytes, int tile_voxels, uint32_t flags,
                         fiery_stream_t stream);

/* Backward of fiery_voxel_pool_fwd with respect to x (training).  Replaces `VoxelsSumming.backward`
 * (fiery/utils/geometry.py:304-314, `grad_out[cumsum(keep) - keep]`) together with what autograd does around it to
 * undo the argsort, the bounds mask and the reshape (fiery.py:233-261): every in-grid point receives the gradient
 * of the voxel it was added to, every other point zero.  A copy: bit-exact.  Geometry gets no gradient
 * (`ctx.mark_non_differentiable`, geometry.py:300).
 *   grad_out : [frames][C][n_voxels]            (n_voxels = X*Y)
 *   rank     : int32 [frames][n_cameras][D][H][W], voxel rank or -1 - the first frames*n_cameras*D*H*W ints of the
 *              workspace the forward call filled, or the output of fiery_voxel_index
 *   grad_x   : logical [frames][n_cameras][D][H][W][C] addressed through gx_strides (host [6]) like x in the forward
 *              call; every element is written. */
int fiery_voxel_pool_bwd(const float* grad_out, const int32_t* rank, int frames, int n_cameras, int D, int H, int W,
                         int C, int n_voxels, float* grad_x, const int64_t* gx_strides /* host [6] */,
                         fiery_stream_t stream);

/* Backward of fiery_lift_splat_fwd (autograd through fiery/models/encoder.py:99-100 and the pooling) without the
 * (n, C, D, H, W) gradient of the outer product:
 *   grad_depth[f][n][d][h][w]    = sum_c features[f][n][c][h][w]   * grad_out[f][c][rank[f][n][d][h][w]]
 *   grad_features[f][n][c][h][w] = sum_d depth_prob[f][n][d][h][w] * grad_out[f][c][rank[f][n][d][h][w]]
 * (points with rank -1 contribute nothing).  Either output may be NULL.  workspace: 16-byte aligned,
 * fiery_lift_splat_bwd_workspace_bytes(frames, C, n_voxels) bytes (the voxel-major copy of grad_out). */
size_t fiery_lift_splat_bwd_workspace_bytes(int frames, int C, int n_voxels);
int fiery_lift_splat_bwd(const float* grad_out, const int32_t* rank, const float* depth_prob, const float* features,
                         int frames, int n_cameras, int D, int H, int W, int C, int n_voxels,
                         float* grad_depth, float* grad_features, void* workspace, size_t workspace_bytes,
                         fiery_stream_t stream);

/* softmax over the depth axis: logits [n][D][HW] -> prob (reference: fiery/models/encoder.py:99). */
int fiery_depth_softmax(const float* logits, int n, int D, int HW, float* prob, fiery_stream_t stream);

/* its backward: grad_logits = prob * (grad_prob - sum_D prob * grad_prob)   (autograd of encoder.py:99) */
int fiery_depth_softmax_bwd(const float* prob, const float* grad_prob, int n, int D, int HW, float* grad_logits,
                            fiery_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Ego-motion warp                  (reference: fiery/utils/geometry.py:181-253
 *                                   `cumulative_warp_features` / `warp_features`, :82-157 pose algebra)
 * ---------------------------------------------------------------------------------------------- */

/* theta[b][s][6]: the 2x3 sampling transform warp_features builds (geometry.py:192-215) for frame s of
 * batch b, i.e. flow[s] @ ... @ flow[S-2] reduced to (cos, -sin, ty/extent_y, sin, cos, -tx/extent_x);
 * entries for s = S-1 are the identity.  future_egomotion [B][S][6].
 * ego_shifted (optional): the temporal model's ego-pose input, row (b, 0) = 0 and row (b, s) = future_egomotion[b][s-1]
 * (fiery.py:152-154: `cat([zeros_like(ego[:, :1]), ego[:, :-1]])`), written by the same launch. */
int fiery_warp_params(const float* future_egomotion, int B, int S, float extent_x, float extent_y,
                      float* theta, float* ego_shifted /* [B][S][6] or NULL */, fiery_stream_t stream);

/* Label warping of the training side (fiery/trainer.py:133-191 -> `cumulative_warp_features_reverse`,
 * fiery/utils/geometry.py:256-280): theta[b][0] = identity, theta[b][i] = the sampling transform of
 * inverse(flow[0]) @ ... @ inverse(flow[i-1]), same six numbers per frame as fiery_warp_params. */
int fiery_warp_params_reverse(const float* future_egomotion, int B, int S, float extent_x, float extent_y,
                              float* theta, fiery_stream_t stream);

/* `flags` of the two resampling entry points.  The sampling positions are evaluated in the rounding order of ATen's CPU
 * affine_grid + grid_sample (csrc/warp.hip), whose one host-dependent step is the BLAS product base_grid . theta^T: MKL
 * fuses it (k ascending) on Intel hosts and rounds products and sums separately on AMD hosts.  Set
 * FIERY_WARP_FUSED_GRID_PRODUCT for the fused form; with the right form for the host the reference's CPU path was
 * timed on, and transforms computed by that host, the resampled maps equal the reference's bit for bit. */
#define FIERY_WARP_FUSED_GRID_PRODUCT 1

/* out[img][c][y][x] = in[img][c][nearest source pixel of (y, x) under theta[img]] or 0 outside the map:
 * grid_sample(mode='nearest', padding_mode='zeros', align_corners=False) on channel planes (the label tensors are
 * NCHW with 1 .. 6 channels).  in and out [n_img][C][H][W], theta [n_img][6]; not in place. */
int fiery_bev_warp_nearest_nchw(const float* in, const float* theta, int n_img, int C, int H, int W, float* out,
                                int flags, fiery_stream_t stream);

/* Bilinear grid-sample with zero padding, align_corners=False (geometry.py:219-220), reading NCHW
 * [n_img][C][H][W] and writing NHWC (ld, img_stride as given).  Images whose `identity[i]` (host
 * array, may be NULL) is non-zero are copied exactly (the present frame is never resampled,
 * geometry.py:245). */
int fiery_bev_warp_nchw_to_nhwc(const float* in, const float* theta /* [n_img][6] */,
                                const uint8_t* identity /* host */, int n_img, int C, int H, int W,
                                float* out, int out_ld, int64_t out_img_stride, int flags, fiery_stream_t stream);

/* Adjoint of fiery_bev_warp_nchw_to_nhwc in its input (training; the sampling positions depend on the ego-motion only, which
 * carries no gradient - fiery/utils/geometry.py:181-222 under autograd): grad_in[img][c] receives, for every output pixel, its
 * gradient times the four bilinear weights at the four source pixels that lie inside the map (atomic fp32 additions: the order of
 * the few contributions per source pixel is not fixed, as in ATen's own backward); images with identity[i] != 0 are copied.
 * grad_out NHWC (ld, img_stride as given), grad_in [n_img][C][H][W], overwritten. */
int fiery_bev_warp_bwd_nhwc_to_nchw(const float* grad_out, int g_ld, int64_t g_img_stride, const float* theta /* [n_img][6] */,
                                    const uint8_t* identity /* host */, int n_img, int C, int H, int W, float* grad_in, int flags,
                                    fiery_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BEV convolution stack            (reference: fiery/layers/convolutions.py, fiery/layers/temporal.py,
 *                                   fiery/models/{temporal_model,future_prediction,distributions,decoder}.py;
 *                                   on CUDA these are cuDNN conv2d/conv3d + batch_norm + activation calls)
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
    const float* ptr;
    int32_t ld;              /* floats between consecutive pixels                                  */
    int32_t units;           /* 8-channel units taken from this source (0 = unused)               */
    int64_t batch_stride;    /* floats between consecutive batch elements (output image / T_out)  */
    int64_t time_stride;     /* floats between consecutive frames of one batch element            */
} fiery_conv_src;

typedef struct {
    float* ptr;
    int32_t ld;
    int64_t img_stride;      /* floats between consecutive output images */
} fiery_nhwc;

#define FIERY_ACT_NONE 0
#define FIERY_ACT_RELU 1
#define FIERY_ACT_SIGMOID 2
#define FIERY_ACT_SWISH 3        /* v * sigmoid(v): the image trunk's MBConv blocks; plain epilogue only */

#define FIERY_EPI_PLAIN 0      /* out = act(acc*scale+shift [+res before act]) [+res after act]          */
#define FIERY_EPI_GRU_GATES 1  /* channels [0,C/2): out=sigmoid -> update gate; [C/2,C): out2=(1-sigmoid)*aux0 */
#define FIERY_EPI_GRU_OUT 2    /* h~=relu(..); out(=out2) = (1-aux0)*aux1 + aux0*h~   (layers/temporal.py:49-62) */
#define FIERY_EPI_HEADS 3      /* hidden = act(acc*scale+shift) stays on chip; heads_out = final 1x1s of it (see heads) */
#define FIERY_MAX_HEAD_OUTPUTS 8

/* One implicit-GEMM convolution with a fused epilogue, NHWC fp32, fp32 MFMA accumulate.
 * Input channels are the virtual concatenation of src[0] and src[1] (8-channel units); the kernel
 * taps are (kT, kH, kW) with causal zero padding in time (layers/temporal.py:65-85) and symmetric
 * zero padding (padH, padW) in space.  Weights come pre-packed by fiery_conv_pack_weights. */
typedef struct {
    fiery_conv_src src[2];
    int32_t Hin, Win, Hout, Wout;
    int32_t n_img_out;                 /* = batch * T_out                                               */
    int32_t T_out;                     /* frames per batch element in the output (1 for 2-D convs)      */
    int32_t t_out0;                    /* absolute time of output frame 0                               */
    int32_t t_in_add;                  /* t_out0 - (absolute time of frame 0 of the sources)            */
    int32_t kT, kH, kW, stride, padH, padW;
    const float* weights;              /* packed [n_tiles][k_chunks][32][BN]                            */
    int32_t cout_pad;                  /* multiple of 32                                                */
    const float* scale;                /* [cout_pad]                                                    */
    const float* shift;                /* [cout_pad]                                                    */
    const float* img_bias;             /* [n_img_out][cout_pad] added before scale, or NULL (see img_bias_border) */
    int32_t act;
    int32_t epi;
    int32_t res_before_act;
    fiery_nhwc res;                    /* residual, ptr NULL = none (const in practice)                */
    fiery_nhwc out;
    int32_t cout_store;                /* channels written to out (<= cout_pad); FIERY_EPI_GRU_GATES: channels of EACH
                                        * half of the result (the update gate to out, (1 - reset) * state to out2)     */
    fiery_nhwc out2;                   /* second destination (GRU modes), ptr NULL = none               */
    fiery_nhwc aux0, aux1;             /* GRU operands                                                   */
    /* Optional chained 1x1 convolution on the result tile (the Bottleneck's up-projection,
     * layers/convolutions.py:128-133): when weights2 != NULL (requires cout_pad == 32 and FIERY_EPI_PLAIN) the
     * activation act(acc*scale+shift) is not stored but multiplied, still on chip, by weights2
     * (packed with fiery_conv_pack_weights for 4 input units and cout2 <= 64 outputs), then
     * out = act2(.*scale2 + shift2) [+ res after the activation]; cout_store counts channels of that result. */
    const float* weights2;
    const float* scale2;               /* [64] */
    const float* shift2;               /* [64] */
    int32_t act2;
    /* Optional third stage on the chained result (the NEXT Bottleneck's 1x1 down-projection, 64 -> 32 + BN + ReLU):
     * when weights3 != NULL (requires weights2, 16-byte addressable tensors and cout_store == 64) the finished 64-channel
     * tile - residual included - is multiplied, still on chip, by weights3 (packed for 8 input units and 32 outputs) and
     * out3 = act3(.*scale3 + shift3) receives 32 channels.  act3: FIERY_ACT_NONE or FIERY_ACT_RELU. */
    const float* weights3;
    const float* scale3;               /* [32] */
    const float* shift3;               /* [32] */
    int32_t act3;
    fiery_nhwc out3;
    /* Output pixels per workgroup tile: 0 = let the library choose, 64 or 128 = the caller's choice (a caller that
     * issues the same launch every step can time both once and keep the faster; 64 needs cout_pad % 64 == 0 and
     * no chained 1x1, otherwise the value is ignored).  Results do not depend on it. */
    int32_t tile_m;
    /* != 0: img_bias is [n_img_out][9][cout_pad]; output pixel (y, x) takes row 3*cy + cx with
     * cy = 0 / 1 / 2 for y == 0 / interior / y == Hout-1 and cx likewise.  This is how a spatially constant group
     * of input channels (the broadcast latent sample of the first SpatialGRU, fiery.py:316-330 + temporal.py:36-62)
     * is folded out of a zero-padded 3x3 convolution: its contribution is one of nine per-image vectors,
     * depending on which taps fall inside the image. */
    int32_t img_bias_border;
    /* FIERY_EPI_HEADS: the decoder heads (models/decoder.py:30-51) are Conv3x3 -> BN -> ReLU -> Conv1x1(+bias)
     * [-> Sigmoid].  All heads' 3x3 convolutions run as this one GEMM (cout_pad = 64 hidden channels per head,
     * a multiple of 128); the hidden tile never leaves the chip: output o is
     *   heads_out[o][image*heads_img_stride[o] + pixel] = [sigmoid](bias[o] + sum_c w[o][c] * hidden[64*group[o] + c])
     * i.e. pixel-contiguous planes (NCHW).  `out` is not written and may be null. */
    struct {
        const float* w;                              /* [n_out][64]                                     */
        const float* bias;                           /* [n_out]                                         */
        int32_t n_out;                               /* <= FIERY_MAX_HEAD_OUTPUTS                       */
        int32_t group[FIERY_MAX_HEAD_OUTPUTS];       /* 64-channel group of the hidden tensor it reads  */
        int32_t sigmoid[FIERY_MAX_HEAD_OUTPUTS];
        float* out[FIERY_MAX_HEAD_OUTPUTS];          /* plane of image 0                                */
        int64_t img_stride[FIERY_MAX_HEAD_OUTPUTS];  /* floats between the planes of consecutive images */
    } heads;
    /* Matrix-core precision.  FIERY_PRECISION_BF16 with weights_bf16 != NULL: operands rounded to bf16 (round to nearest
     * even) at the matrix-core inputs - v_mfma_f32_32x32x16_bf16, an eighth of the fp32 matrix-pipe time - with fp32
     * accumulation; activations stay fp32 in memory and every epilogue (BatchNorm, activation, residual, GRU gates,
     * chained 1x1s, heads) is the fp32 one.  Launches the bf16 form does not cover (channel layouts the scalar-addressed
     * loop cannot take: fewer than 32 channels per tap or source, 35- / 70-channel temporal layers) run the fp32 kernel
     * with `weights`, which therefore must always be set; fiery_conv_precision_used tells which form a descriptor gets. */
    const void* weights_bf16;          /* packed by fiery_conv_pack_weights_bf16, or NULL                  */
    int32_t precision;                 /* FIERY_PRECISION_F32 (0) or FIERY_PRECISION_BF16                  */
    /* STREAM-K (round 5).  The tile forms above run one workgroup per output tile, in rounds of as many workgroups as the chip
     * holds; a launch whose last round is partly filled pays for a full one (938 tiles of 128 x 128 on 512 slots: 8 % idle;
     * 294 tiles on 768 slots: 62 %).  stream_k != 0 asks for the form that deals the launch's (tile, K chunk) units out evenly
     * to exactly one round of workgroups instead; tiles whose chunks end up shared are summed through `sk_workspace` (partial
     * accumulators, written and read past the L2s, which are not coherent across XCDs) by the last holder to arrive, in part
     * order - results are deterministic for a given device, and differ from the tile forms' in the last bits (another
     * summation split).  Taken only by launches the form covers (fiery_conv_stream_k_plan says which, and what they need:
     * fp32, scalar-addressed loop, 64- or 128-wide cout tiles, no chained 1x1, no heads); others ignore the request.
     * sk_counters: zero before the first launch; every launch leaves them zero.  One workspace per stream in flight. */
    /* WINOGRAD F(2x2, 3x3) (round 5): winograd != 0 with weights_winograd (fiery_conv_pack_weights_winograd) asks for the form
     * that computes every 2 x 2 output block from a 4 x 4 input block with 16 multiplies per (cin, cout) instead of 36; taken
     * by fp32 launches of 3 x 3 / stride 1 / 'same' layers (kT = 1) with whole 16-channel stages per source, cout_pad % 64 == 0,
     * 16-byte addressable tensors, the plain, GRU or heads epilogues, no chained 1x1 - others ignore the request.  Results differ
     * from the direct form by fp32 rounding (another order of additions; measured 8e-6 on the hot path's outputs).
     * winograd == FIERY_WINOGRAD_SPLIT_TERMS (round 6) with the image of fiery_conv_pack_weights_winograd_split asks for the
     * same form on the bf16 matrix cores with every fp32 operand as three bf16 terms and six partial products per product -
     * fp32 accuracy (rms error against fp64 not above the fp32 matrix instruction's, tools/probe/split_bf16_probe.hip); same
     * conditions. */
    const float* weights_winograd;
    int32_t winograd;
    int32_t stream_k;
    void* sk_workspace;
    int64_t sk_workspace_bytes;
    int32_t* sk_counters;
    int32_t sk_counters_len;
    /* FIERY_PRECISION_F32_SPLIT with a chained 1x1 (weights2) [and a third stage (weights3)]: the split images of those weights -
     * three bf16 terms per value, as the kernel's 16-byte matrix operands: weights2_split [3 terms][2 k halves][2 cout blocks][64
     * lanes][8], lane = (cout % 32, hi), element j = mid channel 16 half + (j < 4 ? 4 hi + j : 8 + 4 hi + j - 4); weights3_split
     * [3 terms][2 k blocks of 32][2 halves][64 lanes][8], lane = (cout, hi), same order inside a half (fiery_amd/ops.py:
     * `_chain_split_image` builds both on the host).  A chained launch without them runs the fp32 kernel. */
    const void* weights2_split;
    const void* weights3_split;
} fiery_conv_desc;

#define FIERY_PRECISION_F32 0
#define FIERY_PRECISION_BF16 1
/* fp32 ACCURACY on the bf16 matrix cores (round 6): weights_bf16 = the image of fiery_conv_pack_weights_split - every operand as
 * three bf16 terms (x = t1 + t2 + t3 exactly), six partial products per product, fp32 accumulation; not less accurate than the
 * fp32 matrix instruction (tools/probe/split_bf16_probe.hip).  Taken by launches of the scalar-addressed loop on 128-pixel tiles
 * with 32- or 64-wide cout tiles (the chained Bottleneck tails included when weights2_split / weights3_split are given);
 * everything else runs the fp32 kernels with `weights`.  fiery_conv_precision_used tells which. */
#define FIERY_PRECISION_F32_SPLIT 2

/* Packs a dense weight W[cout][cin_total][taps] (taps = kT*kH*kW, row-major as PyTorch stores conv
 * weights) into the kernel's layout.  Input channel ci of the logical concat maps to padded position
 * chan_map[ci] (host array, length cin_total); units0+units1 eight-channel units in total. */
size_t fiery_conv_packed_floats(int cout, int cin_units, int taps);
int fiery_conv_pack_weights(const float* w, int cout, int cin_total, int taps,
                            const int32_t* chan_map /* host */, int cin_units,
                            float* packed, fiery_stream_t stream);

/* The split packing (FIERY_PRECISION_F32_SPLIT): per 32-k stage and cout tile [3 terms][k / 8][cout][k % 8] bf16;
 * 3 * fiery_conv_packed_floats(...) / 2 floats' worth of bytes. */
int fiery_conv_pack_weights_split(const float* w, int cout, int cin_total, int taps,
                                  const int32_t* chan_map /* host */, int cin_units,
                                  void* packed, fiery_stream_t stream);

/* The bf16 packing of the same weights ([k / 8][cout][k % 8] bf16 per 32-k stage and cout tile; values rounded to nearest
 * even): fiery_conv_packed_floats(...) / 2 floats' worth of bytes. */
int fiery_conv_pack_weights_bf16(const float* w, int cout, int cin_total, int taps,
                                 const int32_t* chan_map /* host */, int cin_units,
                                 void* packed, fiery_stream_t stream);

/* The Winograd F(2x2, 3x3) image of a 3 x 3 convolution's weights W[cout][cin_total][9]: U = G g G^T per (cout, cin) in fp64,
 * rounded once, packed [cout / 64][16 transform points][cin_pad / 4][64][4]; fiery_conv_winograd_packed_floats(...) floats. */
size_t fiery_conv_winograd_packed_floats(int cout, int cin_units);
int fiery_conv_pack_weights_winograd(const float* w, int cout, int cin_total, const int32_t* chan_map /* host */, int cin_units,
                                     float* packed, fiery_stream_t stream);
/* The split image (fiery_conv_desc.winograd == FIERY_WINOGRAD_SPLIT_TERMS): U as above, each value as three bf16 terms that add
 * up to it exactly, packed as the kernel's 16-byte operands [cout / 64][16 points][cin_pad / 16][2 cout blocks][3 terms][64 lanes][8];
 * fiery_conv_winograd_split_packed_floats(...) floats' worth of bytes (1.5x the fp32 image). */
#define FIERY_WINOGRAD_SPLIT_TERMS 3
size_t fiery_conv_winograd_split_packed_floats(int cout, int cin_units);
int fiery_conv_pack_weights_winograd_split(const float* w, int cout, int cin_total, const int32_t* chan_map /* host */, int cin_units,
                                           float* packed, fiery_stream_t stream);

int fiery_conv_fwd(const fiery_conv_desc* desc /* host */, fiery_stream_t stream);

/* Weight gradient of a 2-D convolution of this library (training: what autograd computes for `conv2d`'s weight in
 * fiery/layers/convolutions.py:9-168, layers/temporal.py:10-62, models/decoder.py:53-91):
 *   dw[cout][tap][c] += sum over output pixels p of grad_out[p][cout] * in[p * stride + tap - pad][c]
 * in: pixel-major [n_img][Hin][Win] rows of in_ld floats, cin_units * 8 (padded) channels; grad_out: pixel-major
 * [n_img][Hout][Wout] rows of g_ld floats; *_img_stride: floats between images (0 = contiguous).  dw is
 * [cout][kH * kW][cin_units * 8] and must be ZERO on entry (partial sums arrive by fp32 atomics: the last bits depend on
 * their order).  The data gradient needs no entry point of its own: it is fiery_conv_fwd with the weights transposed and
 * mirrored (and the gradient zero-stuffed for stride 2), see fiery_amd/train_graph.py. */
int fiery_conv_wgrad(const float* in, int in_ld, int64_t in_img_stride, int cin_units, const float* grad_out, int g_ld,
                     int64_t g_img_stride, int cout, int n_img, int Hin, int Win, int Hout, int Wout, int kH, int kW,
                     int stride, int padH, int padW, float* dw, fiery_stream_t stream);

/* The same with a matrix-core precision: FIERY_PRECISION_BF16 rounds both operands (grad_out and in) to bf16 on chip and
 * accumulates in fp32 - the weight gradient of mixed-precision training; layers its bf16 kernel does not cover (anything but
 * 3 x 3 / stride 1 / pad 1 on 16-byte addressable rows) run the fp32 kernels. */
int fiery_conv_wgrad_prec(const float* in, int in_ld, int64_t in_img_stride, int cin_units, const float* grad_out, int g_ld,
                          int64_t g_img_stride, int cout, int n_img, int Hin, int Win, int Hout, int Wout, int kH, int kW,
                          int stride, int padH, int padW, int precision, float* dw, fiery_stream_t stream);

/* FIERY_PRECISION_F32 or FIERY_PRECISION_BF16: the matrix-core form fiery_conv_fwd runs this descriptor in (negative:
 * an error code - the descriptor is invalid). */
int fiery_conv_precision_used(const fiery_conv_desc* desc /* host */);

/* The form fiery_conv_fwd runs this descriptor in: a request for Winograd (`winograd`) or stream-K (`stream_k`) is taken only
 * where the form covers the launch and silently falls back to the tile form otherwise - callers that account flops or time
 * candidate forms ask here (negative: an error code - the descriptor is invalid). */
#define FIERY_CONV_FORM_TILE 0
#define FIERY_CONV_FORM_STREAM_K 1
#define FIERY_CONV_FORM_WINOGRAD 2
#define FIERY_CONV_FORM_WINOGRAD_SPLIT 3
int fiery_conv_form_used(const fiery_conv_desc* desc /* host */);

/* What the stream-K form of this descriptor needs (the descriptor's own stream_k / sk_* members are not looked at):
 * *n_workgroups = 0 when the form does not cover the launch, else its grid, with *workspace_bytes (partial tiles) and
 * *n_counters (int32, zero-initialised) the caller must provide in sk_workspace / sk_counters.  Returns 0, or an error code
 * for an invalid descriptor. */
int fiery_conv_stream_k_plan(const fiery_conv_desc* desc /* host */, int64_t* workspace_bytes, int32_t* n_counters,
                             int32_t* n_workgroups);

/* Final 1x1 heads: out_nchw[img][o][y][x] = act_o(bias[o] + sum_{c<head_c} w[o][c] * in[img][y][x][c_off[o] + c])
 * (fiery/models/decoder.py:30-51, the last conv (+ Sigmoid) of each head), NCHW result.
 * `in` has C channels per pixel; n_out <= 8; c_off/sigmoid are host arrays of length n_out;
 * w is [n_out][head_c] and bias [n_out] on device. */
int fiery_heads_1x1_nchw(const float* in, int in_ld, int n_img, int HW, int C, int head_c, int n_out,
                         const float* w, const float* bias, const int32_t* c_off /* host */,
                         const uint8_t* sigmoid /* host */, float* out, fiery_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Small dense / pooling / resampling helpers of the stack
 * ---------------------------------------------------------------------------------------------- */

/* mean over `n_pixels` consecutive pixels starting at in + o*outer_stride + i*inner_stride, for
 * o < n_outer, i < n_inner: out[o*n_inner + i][C].  AdaptiveAvgPool2d(1) (models/distributions.py:24)
 * is n_pixels = H*W; the (2,H,W) pyramid average pool (layers/temporal.py:186-191) is
 * n_pixels = 2*H*W over two adjacent frames of a (batch, time) buffer with inner_stride = one frame.
 * workspace >= n_outer*n_inner*C*64 floats. */
int fiery_spatial_mean(const float* in, int in_ld, int64_t outer_stride, int n_outer, int64_t inner_stride,
                       int n_inner, int n_pixels, int C, float* out, float* workspace, fiery_stream_t stream);

/* y[r][o] = clamp(act(scale[o] * (w_mul * sum_j W[o][w_col0 + j] * v[r][j] + (accumulate ? y[r][o] : 0))
 *                     + shift[o]), lo, hi)   for small matrices (1x1 convs on per-image vectors: pyramid-pool
 * branch, ego-pose channels, distribution head and its log-sigma clamp, models/distributions.py:37).
 * W is [n_out][w_ld]; scale/shift may be NULL (1 / 0); lo/hi = -inf/+inf for no clamp. */
int fiery_rowwise_dense(const float* v, int v_ld, int rows, int n_in, const float* W, int w_ld, int w_col0,
                        int n_out, float w_mul, const float* scale, const float* shift, int act, int accumulate,
                        float lo, float hi, float* y, int y_ld, fiery_stream_t stream);

/* Average pooling of spatially CONSTANT channels exactly as ATen computes it: avg_pool3d adds the window's
 * elements one by one in fp32 (t, then y, then x), so 2*H*W additions of a constant round in the same
 * direction every time and the reference's pooled ego-pose channels (layers/temporal.py:186-191 applied to
 * the channels fiery.py:148-155 appends) are ~5e-4 off their true mean.  This reproduces that sum bit for bit:
 * out[r][j] = (count_each additions of prev[r][j], then count_each additions of cur[r][j]) / total count;
 * prev may be NULL (window clipped at t = 0). */
int fiery_sequential_window_mean(const float* prev, const float* cur, int rows, int n, int count_each,
                                 float* out, int out_ld, fiery_stream_t stream);

/* sample[r][j] = mu[r][j] + exp(log_sigma[r][j]) * noise[r][j]   (noise NULL = zeros)
 * (fiery/models/fiery.py:316-327, inference branch). */
int fiery_latent_sample(const float* mu, const float* log_sigma, const float* noise, int ld, int rows, int n,
                        float* sample, int sample_ld, fiery_stream_t stream);

/* 2x2 stride-2 max pooling, NHWC, odd sizes padded with one zero row/column first
 * (layers/convolutions.py:150,166). */
int fiery_maxpool2x2_nhwc(const float* in, int in_ld, int64_t in_img_stride /* floats between images; 0 = H*W*in_ld */,
                          int n_img, int H, int W, int C, float* out, int out_ld, fiery_stream_t stream);

/* Gradient of fiery_maxpool2x2_nhwc in its input: grad_in (n_img, H, W, C in rows of gi_ld) gets grad_out at the first maximum
 * of each window in row-major order (ATen's max_pool2d_with_indices tie rule; the zero column / row that pads an odd size takes
 * part in the maximum and its share of the gradient is dropped) and zeros elsewhere. */
int fiery_maxpool2x2_bwd_nhwc(const float* in, int in_ld, int64_t in_img_stride /* 0 = H*W*in_ld */, const float* grad_out, int g_ld,
                              int n_img, int H, int W, int C, float* grad_in, int gi_ld, fiery_stream_t stream);

/* Depthwise k x k convolution, NHWC, + folded BatchNorm (scale, shift; may be NULL) + activation: the MBConv blocks of the
 * image trunk (efficientnet-pytorch `MBConvBlock._depthwise_conv` + `_bn1` + swish, behind fiery/models/encoder.py:58-86).
 * w: tap-major [k*k][w_ld >= C].  Zero padding: pad_top / pad_left rows / columns before the image, after it whatever
 * Hout / Wout ask for (the trunk's "static same" padding is asymmetric).  C and the leading dimensions multiples of 4. */
int fiery_depthwise_conv_nhwc(const float* in, int in_ld, int n_img, int H, int W, int C, const float* w, int w_ld,
                              int k, int stride, int pad_top, int pad_left, int Hout, int Wout, const float* scale,
                              const float* shift, int act, float* out, int out_ld, fiery_stream_t stream);

/* Weight gradient of the depthwise convolution (training of the image trunk; what autograd computes for the weight of
 * `MBConvBlock._depthwise_conv`): dw[tap][c] += sum over images and output pixels of grad_out . in at the tap; dw is tap-major
 * [k*k][dw_ld >= C] like the forward's weights and must be ZERO on entry (partial sums arrive by fp32 atomics).  The input
 * gradient needs no entry point of its own: it is fiery_depthwise_conv_nhwc on the output gradient (zero-stuffed for stride
 * 2) with the taps mirrored and the padding k - 1 - pad (fiery_amd/train_graph.py:HipDepthwiseConv2d).  k in {1, 3, 5, 7}. */
int fiery_depthwise_conv_wgrad_nhwc(const float* in, int in_ld, int n_img, int H, int W, int C, const float* grad_out, int g_ld,
                                    int Hout, int Wout, int k, int stride, int pad_top, int pad_left, float* dw, int dw_ld,
                                    fiery_stream_t stream);

/* Squeeze-and-excite gate: gate[img][c] = sigmoid(w2[c][:] . swish(w1 . mean[img] + b1) + b2[c])
 * (efficientnet-pytorch `MBConvBlock._se_reduce` / `_se_expand`).  w1 [hidden][C], w2 [C][hidden]; C <= 1024,
 * hidden <= 64. */
int fiery_se_gate(const float* mean, int mean_ld, int n_img, int C, const float* w1, const float* b1, int hidden,
                  const float* w2, const float* b2, float* gate, int gate_ld, fiery_stream_t stream);

/* The same gate straight from the feature map: channel means of x (n_pixels pixels per image, NHWC) and the two dense
 * layers in two launches.  workspace >= n_img * C * 64 floats. */
int fiery_se_gate_nhwc(const float* x, int ld, int64_t img_stride, int n_img, int n_pixels, int C, const float* w1,
                       const float* b1, int hidden, const float* w2, const float* b2, float* gate, int gate_ld,
                       float* workspace, fiery_stream_t stream);

/* x[img][pixel][c] *= gate[img][c], in place (squeeze-and-excite: `torch.sigmoid(x_squeezed) * x`). */
int fiery_scale_channels_nhwc(float* x, int ld, int n_img, int HW, int C, const float* gate, int gate_ld,
                              fiery_stream_t stream);

/* out = bilinear_x2(in) + shift[c] + skip  (align_corners=False; layers/convolutions.py:203-214 with
 * the 1x1 conv and BN scale already applied at low resolution - both commute with the interpolation).
 * shift and skip may be NULL: the plain x2 interpolation of `UpsamplingConcat` (layers/convolutions.py:171-200). */
int fiery_upsample2x_add_nhwc(const float* in, int in_ld, int n_img, int H, int W, int C,
                              const float* shift, const float* skip, int skip_ld,
                              float* out, int out_ld, fiery_stream_t stream);

/* BatchNorm with batch statistics (+ fused ReLU) on pixel-major rows - what `nn.BatchNorm2d / BatchNorm3d (+ nn.ReLU)` of the
 * BEV stack compute in train() mode (layers/convolutions.py:27-34, 85-105; layers/temporal.py:77-84, 107-117; autograd's
 * backward of the pair).  x: [n_pixels] rows of ld floats, C channels (all images and frames of the batch together).
 * forward: batch_stats != 0: mean / biased variance over the rows, running_mean / running_var (may be NULL) updated with
 *   `momentum` (unbiased variance), as torch does; batch_stats == 0: the running statistics are used.
 *   y = [relu] ((x - mean) * invstd * gamma + beta), gamma / beta may be NULL (1 / 0); channels C .. C_store of y are written
 *   as zeros.  mean, invstd: [C] outputs the backward pass takes.
 * backward: y = the forward's output when it applied the ReLU (gates the gradient), else NULL.  grad_in, dgamma[C], dbeta[C].
 * workspace: fiery_bn_workspace_floats(C) floats, 16-byte aligned; results are deterministic (fixed-order sums). */
int64_t fiery_bn_workspace_floats(int C);
int fiery_bn_train_fwd(const float* x, int ld, int64_t n_pixels, int C, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int batch_stats, float momentum, float eps, int relu,
                       float* y, int y_ld, int C_store, float* mean, float* invstd, float* workspace, fiery_stream_t stream);
int fiery_bn_train_bwd(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels,
                       int C, const float* gamma, const float* mean, const float* invstd, int batch_stats, float* grad_in,
                       int gi_ld, int C_store, float* dgamma, float* dbeta, float* workspace, fiery_stream_t stream);

/* The same passes one at a time, for statistics that span several processes (`nn.SyncBatchNorm`, train.py:35-37): the
 * caller combines the local mean / biased variance / pixel count of every process (Chan's formula) between
 * fiery_bn_train_stats and fiery_bn_apply, and all-reduces the local sums between fiery_bn_train_bwd_sums and
 * fiery_bn_train_bwd_dx (whose total_pixels is the pixel count of all processes). */
int fiery_bn_train_stats(const float* x, int ld, int64_t n_pixels, int C, float* mean, float* var, float* workspace,
                         fiery_stream_t stream);
int fiery_bn_apply(const float* x, int ld, int64_t n_pixels, int C, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, int relu, float* y, int y_ld, int C_store, fiery_stream_t stream);
int fiery_bn_train_bwd_sums(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels,
                            int C, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* workspace,
                            fiery_stream_t stream);
int fiery_bn_train_bwd_dx(const float* grad_out, int g_ld, const float* x, int ld, const float* y, int y_ld, int64_t n_pixels,
                          int C, const float* gamma, const float* mean, const float* invstd, const float* dgamma,
                          const float* dbeta, int64_t total_pixels, float* grad_in, int gi_ld, int C_store, fiery_stream_t stream);

/* The element-wise half of `SpatialGRU.gru_cell` (layers/temporal.py:49-62) for training, one streaming pass per direction:
 *   reset:  r = sigmoid(pre + bias),  rh = (1 - r) * h                      backward: d_pre = -d_rh * h * r (1 - r),  dh = d_rh * (1 - r)
 *   out:    u = sigmoid(pre + bias),  h_new = (1 - u) * h + u * cand        backward: d_pre = d_hn (cand - h) u (1 - u),  dh = d_hn (1 - u),
 *                                                                                     dcand = d_hn * u
 * pre = the gate convolution's raw output (no bias), bias[C] = conv bias + gru_bias_init.  Inputs are pixel-major rows with their
 * own leading dimensions; every output is dense [n_pixels][C_store] with channels C .. C_store zero; C, C_store multiples of 4. */
int fiery_gru_reset_fwd(const float* pre, int pre_ld, const float* bias, const float* h, int h_ld, int64_t n_pixels, int C,
                        float* r, float* rh, int C_store, fiery_stream_t stream);
int fiery_gru_reset_bwd(const float* d_rh, int g_ld, const float* r, const float* h, int h_ld, int64_t n_pixels, int C,
                        float* d_pre, float* dh, int C_store, fiery_stream_t stream);
int fiery_gru_out_fwd(const float* pre, int pre_ld, const float* bias, const float* h, int h_ld, const float* cand, int cand_ld,
                      int64_t n_pixels, int C, float* u, float* h_new, int C_store, fiery_stream_t stream);
int fiery_gru_out_bwd(const float* d_hn, int g_ld, const float* u, const float* h, int h_ld, const float* cand, int cand_ld,
                      int64_t n_pixels, int C, float* d_pre, float* dh, float* dcand, int C_store, fiery_stream_t stream);

/* Instance labels of the input pipeline - `convert_instance_mask_to_center_and_offset_label` (fiery/utils/instance.py:12-77,
 * called per sample by fiery/data.py): ids[T][H][W] instance-id maps (0 = background, 1 .. n_instances), warped_ids[T][H][W]
 * the same maps resampled (nearest) into the previous frame's ego frame (frame 0 unused) ->
 * centerness[T][1][H][W] = max over the frame's instances of exp(-d^2 / sigma^2) around their rounded centres of mass,
 * offset[T][2][H][W] = (centre - pixel) for the pixel's own instance, flow[T][2][H][W] = (warped centre in t + 1) - (centre in t)
 * for instances present in both frames; `ignore_index` where undefined.  workspace: fiery_instance_labels_workspace_ints ints. */
int64_t fiery_instance_labels_workspace_ints(int T, int n_instances);
int fiery_instance_labels(const int32_t* ids, const int32_t* warped_ids, int T, int H, int W, int n_instances, float sigma,
                          float ignore_index, float* centerness, float* offset, float* flow, int32_t* workspace,
                          fiery_stream_t stream);

/* Camera-image preparation of the input pipeline: `resize_and_crop_image` (fiery/utils/geometry.py:8-12: PIL resize BILINEAR +
 * crop) followed by `normalise_image` (fiery/data.py:53-57: ToTensor + Normalize).  images: [n][in_h][in_w][3] uint8 (decoded RGB).
 * The resized image (res_h x res_w) is Pillow's 8-bit resampling byte for byte: two separable passes with the coefficient tables
 * of Pillow's precompute_coeffs / normalize_coeffs_8bpc, computed by the caller on the host (bounds_*[size][2] = first tap, tap
 * count; kk_*[size][ksize_*] = 22-bit fixed-point weights).  Only the crop window (crop_left, crop_top, crop_w, crop_h; parts
 * outside the resized image are black, as Image.crop pads) is produced: out[n][3][crop_h][crop_w] = ((v / 255) - mean) / std.
 * tmp: n * tmp_h * crop_w * 3 bytes for rows y_first .. y_first + tmp_h - 1 of the horizontal pass (the input rows the window's
 * vertical taps read).  mean3 / std3 are HOST arrays. */
int fiery_image_resize_crop_normalise(const uint8_t* images, int n, int in_h, int in_w, int res_h, int res_w,
                                      const int32_t* bounds_h, const int32_t* kk_h, int ksize_h, const int32_t* bounds_v,
                                      const int32_t* kk_v, int ksize_v, int y_first, int tmp_h, int crop_left, int crop_top,
                                      int crop_w, int crop_h, const float* mean3 /* host */, const float* std3 /* host */,
                                      uint8_t* tmp, float* out, fiery_stream_t stream);

/* Gradient of the plain x2 interpolation (fiery_upsample2x_add_nhwc without shift / skip) with respect to its input -
 * what autograd computes for `nn.Upsample(scale_factor=2, mode='bilinear')` in layers/convolutions.py:203-214 (training).
 * grad_out: [n_img][2H][2W] rows of g_ld floats; grad_in: [n_img][H][W] rows of gi_ld floats; C a multiple of 4. */
int fiery_upsample2x_bwd_nhwc(const float* grad_out, int g_ld, int n_img, int H, int W, int C, float* grad_in, int gi_ld,
                              fiery_stream_t stream);

/* out[img][p][c0 + c] = v[img][c] for every pixel (spatial broadcast of the latent sample,
 * fiery/models/fiery.py:329-330). */
int fiery_broadcast_nhwc(const float* v, int v_ld, int n_img, int HW, int C, float* out, int out_ld,
                         int64_t out_img_stride, fiery_stream_t stream);

/* layout changes at the API seams */
int fiery_nchw_to_nhwc(const float* in, int n_img, int C, int HW, float* out, int out_ld,
                       int64_t out_img_stride, fiery_stream_t stream);
int fiery_nhwc_to_nchw(const float* in, int in_ld, int64_t in_img_stride, int n_img, int C, int HW,
                       float* out, fiery_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The step after the path (evaluation): per-frame instance segmentation
 *                                  (reference: fiery/utils/instance.py:80-144
 *                                   `get_instance_segmentation_and_centers`, called per frame from
 *                                   `predict_instance_segmentation_and_trajectories`, evaluate.py:62)
 * ---------------------------------------------------------------------------------------------- */

/* center [n_frames][H][W] (centerness), offset [n_frames][2][H][W], foreground uint8 [n_frames][H][W] (1 = vehicle).
 * Centres = local 3x3 maxima above conf_threshold, row-major order, the first max_centers (<= 256) kept:
 * centers [n_frames][max_centers][2] = (row, column), -1 past n_centers[f].  instance_seg [n_frames][H][W]: 0 =
 * background, else the consecutive id of the nearest centre to (pixel + offset), ids renumbered in ascending order
 * over those that occur (a frame without background pixels starts at 0, as `torch.unique` makes it). */
int fiery_instance_segmentation(const float* center, const float* offset, const uint8_t* foreground, int n_frames,
                                int H, int W, float conf_threshold, int max_centers, int32_t* instance_seg,
                                int32_t* centers, int32_t* n_centers, fiery_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FIERY_HIP_H */
