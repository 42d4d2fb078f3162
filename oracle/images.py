"""TEST INFRASTRUCTURE: the reference's image preparation with the real Pillow.

`resize_and_crop_image` (fiery/utils/geometry.py:8-12) IS two Pillow calls, and Pillow is installed here - so the oracle for the
resampling is the third-party library itself, not a restatement.  `normalise_image` (fiery/data.py:53-57) is torchvision's
ToTensor + Normalize; torchvision is absent, the two transforms are restated from their documented definition
(uint8 HWC -> float CHW / 255; (x - mean) / std per channel, fp32).  Only tests import this module.
"""
import numpy as np
import PIL.Image
import torch


def resize_and_crop_image(img, resize_dims, crop):
    img = img.resize(resize_dims, resample=PIL.Image.BILINEAR)
    return img.crop(crop)


def normalise_image(img, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    tensor = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    mean_t, std_t = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1), torch.tensor(std, dtype=torch.float32).view(3, 1, 1)
    return tensor.sub(mean_t).div(std_t)


def prepare(images_u8, resize_dims, crop):
    """(n, H, W, 3) uint8 array -> (n, 3, h, w) float32 through PIL."""
    return torch.stack([normalise_image(resize_and_crop_image(PIL.Image.fromarray(frame), resize_dims, crop)) for frame in images_u8])
