#!/usr/bin/env python3
"""The flow-side loss kernels at the training step's shapes (32 pairs, 352 x 352, planar f32): flow_warp alone and the whole
UnflowPairLossFn forward + backward (occlusion masks, two warps, two photometric losses and their gradients).
A/B two builds with EMIP_HIP_LIB."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from emip_amd import _lib, ops
from emip_amd.autograd import UnflowPairLossFn
_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best
g = torch.Generator(device="cuda").manual_seed(1)
im1 = torch.randn(B, 3, 352, 352, device="cuda", generator=g); im2 = torch.randn(B, 3, 352, 352, device="cuda", generator=g)
fw = (torch.randn(B, 2, 352, 352, device="cuda", generator=g) * 3).requires_grad_(True)
bw = (torch.randn(B, 2, 352, 352, device="cuda", generator=g) * 3).requires_grad_(True)
print("flow_warp: %.1f us (%.0f MB moved)" % (timeit(lambda: ops.flow_warp(im2, fw.detach())), (2 * im1.numel() + fw.numel()) * 4 / 1e6))
m1 = ops.occ_mask_backward(bw.detach(), complement=True); m2 = ops.occ_mask_backward(fw.detach(), complement=True)
print("occ_mask_backward: %.1f us" % timeit(lambda: ops.occ_mask_backward(bw.detach(), complement=True)))
def fb():
    fw.grad = bw.grad = None
    l = UnflowPairLossFn.apply(fw, bw, im1, im2, m1, m2)
    l.backward()
    return l
l = fb(); torch.cuda.synchronize()
print("UnflowPairLossFn forward + backward: %.1f us   loss %.6f  |dfw| %.6e |dbw| %.6e" % (
    timeit(fb), float(l), float(fw.grad.abs().sum()), float(bw.grad.abs().sum())))
