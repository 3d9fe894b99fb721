"""Input preparation on MI355X: `Resize((352, 352)) -> ToTensor -> Normalize` of the reference's datasets
(/root/reference/dataset/dataset.py:257-260 for inference, :76-79 for training).

The decoded frame stays 8-bit RGB until it is on the device; the resize is Pillow's own two-pass integer resampling
(bit-exact, see csrc/preprocess.hip), so predictions do not change when the host-side PIL resize is replaced.  Decoding
(JPEG/PNG) stays on the host; the training-time augmentations (dataset.py:95-98) are in emip_amd/data_augment.py."""
import math

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 32 - 8 - 2           # Pillow Resample.c
_tables = {}


def pillow_bilinear_coeffs(in_size, out_size):
    """Quantised coefficients and bounds of Pillow's resampling for one axis (Resample.c: precompute_coeffs with the
    bilinear / triangle filter of support 1, then normalize_coeffs_8bpc).  Pure double arithmetic as in C.
    -> (kk int32 [out_size, ksize], bounds int32 [out_size, 2] = (first tap, tap count))"""
    scale = in_size / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            w.append(1.0 - v if v < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << _PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << _PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _device_tables(h0, w0, ho, wo, device):
    key = (h0, w0, ho, wo, str(device))
    t = _tables.get(key)
    if t is None:
        kh, bh = pillow_bilinear_coeffs(w0, wo)
        kv, bv = pillow_bilinear_coeffs(h0, ho)
        t = tuple(torch.from_numpy(a).contiguous().to(device) for a in (kh, bh, kv, bv))
        _tables[key] = t
    return t


def rgb_to_model_input(img_u8, size=352, mean=IMAGENET_MEAN, std=IMAGENET_STD, return_resized=False):
    """img_u8: uint8 [B,H0,W0,3] (or [H0,W0,3]) decoded RGB frames on the device -> normalised f32 [B,3,size,size]
    (what `self.transform(image)` returns in dataset.py:266, stacked); optionally also the resized u8 pixels."""
    assert img_u8.is_cuda and img_u8.dtype == torch.uint8 and img_u8.shape[-1] == 3
    x = img_u8 if img_u8.dim() == 4 else img_u8.unsqueeze(0)
    x = x.contiguous()
    B, H0, W0, _ = x.shape
    kh, bh, kv, bv = _device_tables(H0, W0, size, size, x.device)
    tmp = torch.empty((B, H0, size, 3), dtype=torch.uint8, device=x.device)
    out = torch.empty((B, 3, size, size), dtype=torch.float32, device=x.device)
    u8 = torch.empty((B, size, size, 3), dtype=torch.uint8, device=x.device) if return_resized else None
    import ctypes
    m = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in mean])
    s = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in std])
    _lib.call("emip_preprocess_rgb", x.data_ptr(), H0 * W0 * 3, W0 * 3, B, H0, W0, kh.data_ptr(), bh.data_ptr(),
              kh.shape[1], kv.data_ptr(), bv.data_ptr(), kv.shape[1], tmp.data_ptr(), out.data_ptr(),
              u8.data_ptr() if u8 is not None else None, size, size, ctypes.addressof(m), ctypes.addressof(s),
              torch.cuda.current_stream().cuda_stream)
    return (out, u8) if return_resized else out


def gray_to_model_input(gt_u8, size=352, return_resized=False):
    """gt_u8: uint8 [B,H0,W0] (or [H0,W0]) 'L' masks on the device -> f32 [B,1,size,size] = `gt_transform(gt)` of
    dataset.py:80-82 (Resize + ToTensor), stacked; optionally also the resized u8 pixels."""
    assert gt_u8.is_cuda and gt_u8.dtype == torch.uint8
    x = gt_u8 if gt_u8.dim() == 3 else gt_u8.unsqueeze(0)
    x = x.contiguous()
    B, H0, W0 = x.shape
    kh, bh, kv, bv = _device_tables(H0, W0, size, size, x.device)
    tmp = torch.empty((B, H0, size), dtype=torch.uint8, device=x.device)
    out = torch.empty((B, 1, size, size), dtype=torch.float32, device=x.device)
    u8 = torch.empty((B, size, size), dtype=torch.uint8, device=x.device) if return_resized else None
    _lib.call("emip_preprocess_gray", x.data_ptr(), H0 * W0, W0, B, H0, W0, kh.data_ptr(), bh.data_ptr(), kh.shape[1],
              kv.data_ptr(), bv.data_ptr(), kv.shape[1], tmp.data_ptr(), out.data_ptr(),
              u8.data_ptr() if u8 is not None else None, size, size, torch.cuda.current_stream().cuda_stream)
    return (out, u8) if return_resized else out
