"""Camera-image preparation of the input pipeline (reference: fiery/data.py:126-148, 214-228 and fiery/utils/geometry.py:8-36).

Per camera image the reference's dataset worker runs `resize_and_crop_image` (PIL `Image.resize(resize_dims, BILINEAR)` - an
antialiased, fixed-point, two-pass resampling for 8-bit images - and `Image.crop`) and `normalise_image` (torchvision `ToTensor` +
`Normalize`): 42 images of 1600 x 900 per sample.  `resize_crop_normalise` does a whole stack of decoded images in two kernel
launches (`fiery_image_resize_crop_normalise`) with Pillow's result byte for byte: the coefficient tables depend only on the sizes
and are computed here on the host in double precision exactly as libImaging/Resample.c computes them (`precompute_coeffs`,
`normalize_coeffs_8bpc`), the device does the integer arithmetic.  JPEG decoding stays with the caller.
"""
import math

import torch

from . import native

IMAGENET_MEAN = (0.485, 0.456, 0.406)            # fiery/data.py:55
IMAGENET_STD = (0.229, 0.224, 0.225)
PRECISION_BITS = 32 - 8 - 2                      # libImaging/Resample.c


def get_resizing_and_cropping_parameters(cfg):
    """fiery/data.py:126-148."""
    original_height, original_width = cfg.IMAGE.ORIGINAL_HEIGHT, cfg.IMAGE.ORIGINAL_WIDTH
    final_height, final_width = cfg.IMAGE.FINAL_DIM
    resize_scale = cfg.IMAGE.RESIZE_SCALE
    resize_dims = (int(original_width * resize_scale), int(original_height * resize_scale))
    resized_width, resized_height = resize_dims
    crop_h = cfg.IMAGE.TOP_CROP
    crop_w = int(max(0, (resized_width - final_width) / 2))
    crop = (crop_w, crop_h, crop_w + final_width, crop_h + final_height)
    return {'scale_width': resize_scale, 'scale_height': resize_scale, 'resize_dims': resize_dims, 'crop': crop}


def update_intrinsics(intrinsics, top_crop=0.0, left_crop=0.0, scale_width=1.0, scale_height=1.0):
    """fiery/utils/geometry.py:15-36: the 3 x 3 intrinsics after resizing and cropping."""
    updated = intrinsics.clone()
    updated[0, 0] *= scale_width
    updated[0, 2] *= scale_width
    updated[1, 1] *= scale_height
    updated[1, 2] *= scale_height
    updated[0, 2] -= left_crop
    updated[1, 2] -= top_crop
    return updated


def bilinear_coefficients(in_size, out_size):
    """Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` for the BILINEAR filter over the whole axis (box = (0, in_size)):
    -> (bounds [out_size][2] = (first input index, tap count), kk [out_size][ksize] 22-bit fixed point), python ints.
    Double arithmetic in the order of the C source (python floats are C doubles)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                                  # bilinear support = 1
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)                       # C cast: truncation toward zero (the value is >= -0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        weights, total = [], 0.0
        for x in range(xmax):
            arg = (x + xmin - center + 0.5) * ss
            if arg < 0.0:
                arg = -arg
            w = 1.0 - arg if arg < 1.0 else 0.0
            weights.append(w)
            total += w
        if total != 0.0:
            weights = [w / total for w in weights]
        row = [int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS)) for w in weights]
        kk.append(row + [0] * (ksize - xmax))
        bounds.append((xmin, xmax))
    return bounds, kk


_TABLES = {}


def _tables(in_hw, res_hw, device):
    key = (tuple(in_hw), tuple(res_hw), str(device))
    if key not in _TABLES:
        bh, kh = bilinear_coefficients(in_hw[1], res_hw[1])
        bv, kv = bilinear_coefficients(in_hw[0], res_hw[0])
        as_dev = lambda rows: torch.tensor(rows, dtype=torch.int32).to(device).contiguous()
        _TABLES[key] = (as_dev(bh), as_dev(kh), as_dev(bv), as_dev(kv), bv)
    return _TABLES[key]


def resize_crop_normalise(images, resize_dims, crop, mean=IMAGENET_MEAN, std=IMAGENET_STD, lib=None, device=None):
    """images: (n, H, W, 3) uint8 decoded RGB frames (what `np.asarray(PIL.Image.open(..))` gives), host or GPU;
    resize_dims = (width, height) and crop = (left, top, right, bottom) as `get_resizing_and_cropping_parameters` returns them
    -> (n, 3, crop height, crop width) float32 = normalise_image(resize_and_crop_image(img, resize_dims, crop)) per image, on the
    device.  Host inputs are moved to `device` (default: the current HIP device; there is no CPU path)."""
    native.require_usable_gpu_process('resize_crop_normalise')
    lib = lib or native.get()
    where = images.device
    device = torch.device(device) if device is not None else (where if where.type == 'cuda' else torch.device('cuda'))
    n, in_h, in_w, ch = images.shape
    assert ch == 3 and images.dtype == torch.uint8, 'expected (n, H, W, 3) uint8 images'
    res_w, res_h = int(resize_dims[0]), int(resize_dims[1])
    left, top, right, bottom = (int(v) for v in crop)
    crop_w, crop_h = right - left, bottom - top
    assert crop_w > 0 and crop_h > 0
    bounds_h, kk_h, bounds_v, kk_v, bv_host = _tables((in_h, in_w), (res_h, res_w), device)
    # input rows the window's vertical taps read (Pillow computes the horizontal pass for ybox_first .. ybox_last only, too)
    rows = [bv_host[yy] for yy in range(max(top, 0), min(bottom, res_h))]
    if rows:
        y_first = min(r[0] for r in rows)
        y_last = max(r[0] + r[1] for r in rows)
    else:
        y_first, y_last = 0, 1
    return lib.image_resize_crop_normalise(images.to(device).contiguous(), (res_h, res_w), (bounds_h, kk_h, bounds_v, kk_v),
                                           (y_first, y_last - y_first, left, top, crop_w, crop_h), mean, std)
