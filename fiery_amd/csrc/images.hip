// Camera-image preparation of the input pipeline: bilinear resize (antialiased, as Pillow does it), crop, ToTensor, Normalize
// (SURVEY.md section 8f rank 4).
//
// Replaces, per camera image, `resize_and_crop_image` (fiery/utils/geometry.py:8-12: PIL `Image.resize(.., BILINEAR)` +
// `Image.crop`) and `normalise_image` (fiery/data.py:53-57, 216-219: torchvision `ToTensor` + `Normalize`) - 42 images of
// 1600 x 900 per sample in every dataloader worker.  Pillow's 8-bit resampling (libImaging/Resample.c) is two separable passes
// with fixed-point coefficients and an 8-bit intermediate image:
//   horizontal: tmp[y][xx][c] = clip8((2^21 + sum_x in[y][xmin(xx) + x][c] * kh[xx][x]) >> 22)
//   vertical:   res[yy][xx][c] = clip8((2^21 + sum_y tmp[ymin(yy) + y][xx][c] * kv[yy][y]) >> 22)
// with coefficient tables (bounds + 22-bit integers) that depend only on the sizes; the caller computes them on the host in
// double precision exactly as Pillow does (fiery_amd/images.py) and hands them over, so the result is Pillow's byte for byte.
// Only the cropped window is computed; the crop's part outside the resized image is black (Pillow pads with zeros).
// Then out[c][y][x] = ((res / 255) - mean[c]) / std[c] in fp32, each operation rounded on its own (as torch evaluates it).
#include "common.h"

namespace fiery {
namespace {

__device__ __forceinline__ int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

struct ResizeP {
    const uint8_t* in;       // [n][in_h][in_w][3]
    uint8_t* tmp;            // [n][tmp_h][crop_w][3]: rows y_first .. y_first + tmp_h - 1 of the horizontally resized image
    float* out;              // [n][3][crop_h][crop_w]
    const int* bounds_h;     // [res_w][2] (xmin, count)
    const int* kk_h;         // [res_w][ksize_h]
    const int* bounds_v;     // [res_h][2] (ymin, count), ymin relative to the input image
    const int* kk_v;         // [res_h][ksize_v]
    int n, in_h, in_w, res_h, res_w, ksize_h, ksize_v;
    int crop_left, crop_top, crop_w, crop_h;
    int y_first, tmp_h;
    float mean[3], stdv[3];
};

// one thread = one pixel of the intermediate image
__global__ __launch_bounds__(256) void k_resize_horizontal(ResizeP p) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long per_img = static_cast<long long>(p.tmp_h) * p.crop_w;
    if (i >= per_img * p.n) return;
    const int img = static_cast<int>(i / per_img);
    const int r = static_cast<int>(i - img * per_img);
    const int ty = r / p.crop_w, cx = r - ty * p.crop_w;
    const int xx = p.crop_left + cx;                             // column of the resized image
    uint8_t* dst = p.tmp + (i * 3);
    if (xx < 0 || xx >= p.res_w) {
        dst[0] = dst[1] = dst[2] = 0;
        return;
    }
    const int xmin = p.bounds_h[2 * xx], count = p.bounds_h[2 * xx + 1];
    const int* k = p.kk_h + static_cast<long long>(xx) * p.ksize_h;
    const uint8_t* row = p.in + ((static_cast<long long>(img) * p.in_h + (p.y_first + ty)) * p.in_w + xmin) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int x = 0; x < count; ++x) {
        const int w = k[x];
        s0 += row[3 * x] * w;
        s1 += row[3 * x + 1] * w;
        s2 += row[3 * x + 2] * w;
    }
    dst[0] = static_cast<uint8_t>(clip8(s0 >> 22));
    dst[1] = static_cast<uint8_t>(clip8(s1 >> 22));
    dst[2] = static_cast<uint8_t>(clip8(s2 >> 22));
}

// one thread = one pixel of the output window, three channels
__global__ __launch_bounds__(256) void k_resize_vertical_normalise(ResizeP p) {
    const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long per_img = static_cast<long long>(p.crop_h) * p.crop_w;
    if (i >= per_img * p.n) return;
    const int img = static_cast<int>(i / per_img);
    const int r = static_cast<int>(i - img * per_img);
    const int cy = r / p.crop_w, cx = r - cy * p.crop_w;
    const int yy = p.crop_top + cy;                              // row of the resized image
    int v0 = 0, v1 = 0, v2 = 0;                                  // outside the resized image: black
    if (yy >= 0 && yy < p.res_h && p.crop_left + cx >= 0 && p.crop_left + cx < p.res_w) {
        const int ymin = p.bounds_v[2 * yy], count = p.bounds_v[2 * yy + 1];
        const int* k = p.kk_v + static_cast<long long>(yy) * p.ksize_v;
        const uint8_t* col = p.tmp + ((static_cast<long long>(img) * p.tmp_h + (ymin - p.y_first)) * p.crop_w + cx) * 3;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        for (int y = 0; y < count; ++y) {
            const int w = k[y];
            const uint8_t* px = col + static_cast<long long>(y) * p.crop_w * 3;
            s0 += px[0] * w;
            s1 += px[1] * w;
            s2 += px[2] * w;
        }
        v0 = clip8(s0 >> 22);
        v1 = clip8(s1 >> 22);
        v2 = clip8(s2 >> 22);
    }
    const int v[3] = {v0, v1, v2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float unit = __fdiv_rn(static_cast<float>(v[c]), 255.0f);               // ToTensor
        p.out[((static_cast<long long>(img) * 3 + c) * p.crop_h + cy) * p.crop_w + cx] = __fdiv_rn(__fsub_rn(unit, p.mean[c]), p.stdv[c]);
    }
}

}  // namespace
}  // namespace fiery

using namespace fiery;

extern "C" int fiery_image_resize_crop_normalise(const uint8_t* images, int n, int in_h, int in_w, int res_h, int res_w,
                                                 const int32_t* bounds_h, const int32_t* kk_h, int ksize_h, const int32_t* bounds_v,
                                                 const int32_t* kk_v, int ksize_v, int y_first, int tmp_h, int crop_left, int crop_top,
                                                 int crop_w, int crop_h, const float* mean3, const float* std3, uint8_t* tmp,
                                                 float* out, fiery_stream_t stream) {
    FIERY_REQUIRE(images && bounds_h && kk_h && bounds_v && kk_v && mean3 && std3 && tmp && out, "image_resize: null pointer");
    FIERY_REQUIRE(n > 0 && in_h > 0 && in_w > 0 && res_h > 0 && res_w > 0 && ksize_h > 0 && ksize_v > 0 && crop_w > 0 && crop_h > 0,
                  "image_resize: bad shape");
    FIERY_REQUIRE(y_first >= 0 && tmp_h > 0 && y_first + tmp_h <= in_h, "image_resize: intermediate rows outside the input image");
    ResizeP p;
    p.in = images;  p.tmp = tmp;  p.out = out;
    p.bounds_h = bounds_h;  p.kk_h = kk_h;  p.bounds_v = bounds_v;  p.kk_v = kk_v;
    p.n = n;  p.in_h = in_h;  p.in_w = in_w;  p.res_h = res_h;  p.res_w = res_w;  p.ksize_h = ksize_h;  p.ksize_v = ksize_v;
    p.crop_left = crop_left;  p.crop_top = crop_top;  p.crop_w = crop_w;  p.crop_h = crop_h;
    p.y_first = y_first;  p.tmp_h = tmp_h;
    for (int c = 0; c < 3; ++c) {
        p.mean[c] = mean3[c];                                     // host arrays
        p.stdv[c] = std3[c];
    }
    const long long n_tmp = static_cast<long long>(n) * tmp_h * crop_w, n_out = static_cast<long long>(n) * crop_h * crop_w;
    hipLaunchKernelGGL(k_resize_horizontal, dim3(ceil_div(n_tmp, 256)), dim3(256), 0, as_stream(stream), p);
    int rc = check_launch("image_resize (horizontal)");
    if (rc) return rc;
    hipLaunchKernelGGL(k_resize_vertical_normalise, dim3(ceil_div(n_out, 256)), dim3(256), 0, as_stream(stream), p);
    return check_launch("image_resize (vertical)");
}
