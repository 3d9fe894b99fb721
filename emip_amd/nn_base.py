"""Host-side plumbing shared by the nn.Module surfaces.

The modules in emip_amd.model / emip_amd.lib keep the reference's constructor
arguments, attribute names and state_dict keys, but hold
parameters only: their forward passes run hand-written HIP kernels through
emip_amd.ops on channels-last activations.  This file provides

  * the compute-dtype switch (float32 = parity mode, bfloat16 = performance mode),
  * a per-module cache of kernel-ready ("packed") weights, invalidated by the
    parameters' version counters (optimizer steps and load_state_dict bump them),
  * packing helpers (conv weights to [Cout][KH][KW][Cin], BatchNorm folding, ...).
"""
import torch
import torch.nn as nn

_DEFAULT_DTYPE = [torch.float32]


def set_default_dtype(dtype):
    """Compute/storage dtype of activations for modules created or run afterwards."""
    assert dtype in (torch.float32, torch.bfloat16)
    _DEFAULT_DTYPE[0] = dtype


def get_default_dtype():
    return _DEFAULT_DTYPE[0]


class EmipModule(nn.Module):
    """nn.Module with a packed-weight cache.  `self.cdtype` is the activation dtype."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_pack_cache", {})

    @property
    def cdtype(self):
        return _DEFAULT_DTYPE[0]

    def packed(self, key, tensors, builder):
        """builder(*tensors) -> packed object, cached until a tensor's version/device changes."""
        sig = tuple((t.data_ptr(), t._version, t.device) for t in tensors) + (self.cdtype, self.training)
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            val = builder(*tensors)
        self._pack_cache[key] = (sig, val)
        return val

    def _apply(self, fn, *a, **k):  # .to()/.cuda(): drop packed copies living on the old device
        self._pack_cache.clear()
        return super()._apply(fn, *a, **k)


# ---------------------------------------------------------------------------------------------
# packing helpers (weight preprocessing only)


def f32(t):
    return t.detach().float().contiguous()


def pack_linear(w, dtype):
    return w.detach().to(dtype).contiguous()


def pack_conv(w, dtype, cin_pad=None, perm=None):
    """[Cout,Cin,KH,KW] -> [Cout, KH*KW*Cin_pad] with ci fastest (matches emip_conv2d).
    perm: optional input-channel permutation applied before padding."""
    w = w.detach().float()
    if perm is not None:
        w = w[:, perm]
    co, ci, kh, kw = w.shape
    if cin_pad is not None and cin_pad > ci:
        w = torch.cat([w, w.new_zeros(co, cin_pad - ci, kh, kw)], 1)
    return w.permute(0, 2, 3, 1).reshape(co, -1).to(dtype).contiguous()


def fold_bn(w, b, bn, eps=None):
    """Fold an eval-mode BatchNorm2d into the preceding conv: returns (w', b') in f32."""
    eps = bn.eps if eps is None else eps
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + eps)
    w2 = w.detach().float() * scale.view(-1, 1, 1, 1)
    b0 = b.detach().float() if b is not None else torch.zeros_like(scale)
    b2 = (b0 - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
    return w2, b2.contiguous()


def pack_dw(w):
    """depthwise [C,1,3,3] -> f32 [9][C]"""
    return w.detach().float().reshape(w.shape[0], 9).t().contiguous()


def to_cl(x, dtype, cpad=None):
    """module-boundary conversion: planar [B,C,H,W] (any float dtype) -> channels-last"""
    from . import ops
    return ops.planar_to_cl(x.detach().float().contiguous(), dtype, cpad)


def to_planar(x, xc=0, C=None):
    from . import ops
    return ops.cl_to_planar(x, xc, C)
