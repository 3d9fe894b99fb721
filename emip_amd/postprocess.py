"""Prediction post-processing on MI355X: the step right after `CoUpdater.forward` in the reference's inference driver
(/root/reference/test.py:28-31).

    output = F.upsample(output[0], size=shape, mode='bilinear', align_corners=False)
    output = output.sigmoid().data.cpu().numpy().squeeze()
    output = (output - output.min()) / (output.max() - output.min() + 1e-8)
    Image.fromarray(output*255).convert('L')

The reference synchronises the device and moves a full-resolution f32 map to the host per frame pair; here the bytes of
the PNG payload are produced on the device (one C-ABI call, no intermediate tensor) and can be copied out asynchronously."""
import torch

from . import _lib


def mask_to_uint8(mask_logits, shape):
    """mask_logits: planar f32 [B,1,H,W] on the device; shape: (Ho, Wo) of the source frame -> uint8 [B,Ho,Wo] on the device"""
    assert mask_logits.is_cuda and mask_logits.dtype == torch.float32 and mask_logits.dim() == 4 and mask_logits.shape[1] == 1
    x = mask_logits.contiguous()
    B, _, H, W = x.shape
    Ho, Wo = int(shape[0]), int(shape[1])
    out = torch.empty((B, Ho, Wo), dtype=torch.uint8, device=x.device)
    ws = torch.empty(2 * B, dtype=torch.int32, device=x.device)
    _lib.call("emip_postprocess_mask", x.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, W, Ho, Wo,
              torch.cuda.current_stream().cuda_stream)
    return out


def mask_to_float(mask_logits, shape):
    """the same resize + sigmoid + per-image min-max as the f32 map in [0, 1] (train.py:125-127: what the validation metrics
    read); planar f32 [B,1,H,W] on the device -> f32 [B,Ho,Wo] on the device, bit for bit the reference's CPU arithmetic"""
    assert mask_logits.is_cuda and mask_logits.dtype == torch.float32 and mask_logits.dim() == 4 and mask_logits.shape[1] == 1
    x = mask_logits.contiguous()
    B, _, H, W = x.shape
    Ho, Wo = int(shape[0]), int(shape[1])
    out = torch.empty((B, Ho, Wo), dtype=torch.float32, device=x.device)
    ws = torch.empty(2 * B, dtype=torch.int32, device=x.device)
    _lib.call("emip_postprocess_mask_f32", x.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, W, Ho, Wo,
              torch.cuda.current_stream().cuda_stream)
    return out


def mask_to_uint8_host(mask_logits, shape, pinned=None):
    """same, plus an asynchronous copy into (reusable) pinned host memory; returns (host tensor, event to wait on)"""
    dev = mask_to_uint8(mask_logits, shape)
    if pinned is None or pinned.shape != dev.shape:
        pinned = torch.empty(dev.shape, dtype=torch.uint8, pin_memory=True)
    pinned.copy_(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return pinned, ev
