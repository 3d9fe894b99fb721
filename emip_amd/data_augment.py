"""Training-time augmentation on MI355X, same names / argument order / random-number consumption as the reference's
/root/reference/dataset/data_augment.py (used by `ObjDataset.__getitem__`, dataset/dataset.py:90-103), on decoded 8-bit
frames that already live on the device: uint8 [H,W,3] RGB tensors and a uint8 [H,W] ground-truth mask.

The draws come from Python's `random` and numpy's global generator exactly as in the reference, in the same order, so a
run seeded like the reference's sees the same augmentation parameters; the pixel work is Pillow's arithmetic reproduced bit
for bit by csrc/augment.hip.  `train_sample` strings the pieces together the way `__getitem__` does and finishes with the
device Resize / ToTensor / Normalize of emip_amd/preprocess.py."""
import ctypes
import math
import random

import numpy as np
import torch

from . import _lib
from .preprocess import gray_to_model_input, rgb_to_model_input


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check_rgb(img):
    assert img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[-1] == 3
    return img.contiguous()


def rotate_matrix(angle, w, h):
    """the inverse affine matrix PIL.Image.Image.rotate builds for (angle, expand=False, center=None), Image.py:2541-2568"""
    angle = angle % 360.0
    cx, cy = w / 2, h / 2
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    x, y = -cx, -cy
    m[2], m[5] = m[0] * x + m[1] * y + m[2], m[3] * x + m[4] * y + m[5]
    m[2] += cx
    m[5] += cy
    return m


def rotate(img, angle):
    """PIL `img.rotate(angle, Image.BICUBIC)` for uint8 [H,W,3] or [H,W] on the device"""
    x = img.contiguous()
    H, W = x.shape[:2]
    C = 3 if x.dim() == 3 else 1
    a = angle % 360.0
    if a == 0:                                                      # Image.py:2514-2521 fast paths
        return x.clone()
    if a == 180:
        return torch.flip(x, (0, 1)).contiguous()
    if a in (90, 270) and W == H:
        return torch.rot90(x, 1 if a == 90 else 3, (0, 1)).contiguous()
    out = torch.empty_like(x)
    m = (ctypes.c_double * 6)(*rotate_matrix(angle, W, H))
    _lib.call("emip_rotate_bicubic", x.data_ptr(), out.data_ptr(), H, W, C, ctypes.addressof(m), _stream())
    return out


def randomRotation(img1, img2, label):
    """data_augment.py:12-19"""
    if random.random() > 0.8:
        random_angle = np.random.randint(-15, 15)
        img1 = rotate(img1, random_angle)
        img2 = rotate(img2, random_angle)
        label = rotate(label, random_angle)
    return img1, img2, label


def color_enhance(image, bright, contrast, color, sharp):
    x = _check_rgb(image)
    H, W, _ = x.shape
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    lsum = torch.empty(1, dtype=torch.int64, device=x.device)
    _lib.call("emip_color_enhance", x.data_ptr(), out.data_ptr(), tmp.data_ptr(), lsum.data_ptr(), H, W, float(bright),
              float(contrast), float(color), float(sharp), _stream())
    return out


def colorEnhance(image):
    """data_augment.py:22-31 (the four factors are drawn in the reference's order)"""
    bright_intensity = random.randint(5, 15) / 10.0
    contrast_intensity = random.randint(5, 15) / 10.0
    color_intensity = random.randint(0, 20) / 10.0
    sharp_intensity = random.randint(0, 30) / 10.0
    return color_enhance(image, bright_intensity, contrast_intensity, color_intensity, sharp_intensity)


def randomPeper(img):
    """data_augment.py:34-45 on a uint8 [H,W] mask: the positions and values are drawn on the host in the reference's
    order; a pixel drawn more than once keeps its last value, as the sequential loop leaves it"""
    assert img.is_cuda and img.dtype == torch.uint8 and img.dim() == 2
    out = img.contiguous().clone()
    H, W = out.shape
    noiseNum = int(0.0015 * H * W)
    last = {}
    for _ in range(noiseNum):
        randX = random.randint(0, H - 1)
        randY = random.randint(0, W - 1)
        last[randX * W + randY] = 0 if random.randint(0, 1) == 0 else 255
    if last:
        offs = torch.tensor(list(last.keys()), dtype=torch.int32).to(out.device)
        vals = torch.tensor(list(last.values()), dtype=torch.uint8).to(out.device)
        _lib.call("emip_scatter_u8", out.data_ptr(), offs.data_ptr(), vals.data_ptr(), len(last), _stream())
    return out


def train_sample(image1, image2, gt, trainsize=352):
    """ObjDataset.__getitem__ after decoding (dataset/dataset.py:94-103): augmentation, then the two transforms.
    image1 / image2 uint8 [H,W,3], gt uint8 [H,W] ('L') on the device -> (f32 [3,S,S], f32 [3,S,S], f32 [1,S,S])"""
    image1, image2, gt = randomRotation(image1, image2, gt)
    image1 = colorEnhance(image1)
    image2 = colorEnhance(image2)
    gt = randomPeper(gt)
    return (rgb_to_model_input(image1, trainsize)[0], rgb_to_model_input(image2, trainsize)[0],
            gray_to_model_input(gt, trainsize)[0])
