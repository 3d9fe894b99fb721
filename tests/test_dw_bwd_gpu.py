"""emip_dwconv3x3_bwd_fused (dw_bwd.hip): GELU backward + depthwise input gradient + weight / bias gradient in one pass, against
torch autograd in f32 on the bf16-rounded operands (lib/pvt_v2.py:45-54,316-327) and against the three launches it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel(out, ref):
    return ((out.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-9)).item()


# the PVT maps (22-tiles: 88 / 44 / 22; 11-tile: 11), a map the 16-tile has to mask (20 x 30), channel tails (C % 64 != 0)
@pytest.mark.parametrize("B,H,W,C", [(2, 22, 22, 1280), (1, 44, 44, 512), (1, 88, 88, 64), (3, 11, 11, 2048), (2, 20, 30, 72),
                                     (1, 5, 7, 8)])
@pytest.mark.parametrize("gelu", [True, False])
def test_fused_depthwise_backward(B, H, W, C, gelu):
    from emip_amd import ops
    bf = torch.bfloat16
    x = rnd(B, H, W, C, seed=1).to(bf)
    dy = rnd(B, H, W, C, seed=2).to(bf)
    w = rnd(C, 1, 3, 3, seed=3, scale=0.3)
    b = rnd(C, seed=4, scale=0.1)
    wt = w.view(C, 9).t().contiguous().cuda()
    xd, dyd = x.cuda(), dy.cuda()
    # forward on the device: pre-activation z as the training forward stores it (bf16)
    if gelu:
        _, z = ops.dwconv3x3_dual(xd, wt, b.cuda(), ops.ACT_GELU)
    else:
        z = None
    # reference: autograd through conv (+ gelu applied to the STORED pre-activation, like the device path)
    xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, padding=1, groups=C)
    if gelu:
        zf = z.float().cpu().permute(0, 3, 1, 2)
        zl = zf.clone().requires_grad_(True)
        F.gelu(zl).backward(dy.float().permute(0, 3, 1, 2))
        dpre = zl.grad.to(bf).float()            # the three-launch form rounds dPre to bf16; so does the LDS tile
    else:
        dpre = dy.float().permute(0, 3, 1, 2)
    pre.backward(dpre)
    acc = torch.zeros(10 * C, device="cuda:0")
    dx = ops.dwconv3x3_bwd_fused(xd, z, dyd, wt, acc[:9 * C], acc[9 * C:], gelu)
    torch.cuda.synchronize()
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < 1e-2
    assert rel(acc[:9 * C].view(C, 1, 3, 3), wr.grad) < 5e-3
    assert rel(acc[9 * C:], br.grad) < 5e-3
    # and the launches it replaces, on the same operands
    dz = ops.gelu_bwd(z, dyd) if gelu else dyd
    dx_old = ops.dwconv3x3(dz, w.view(C, 9).flip(1).t().contiguous().cuda())
    dw_old = torch.zeros(9, C, device="cuda:0")
    db_old = torch.zeros(C, device="cuda:0")
    ops.dwconv3x3_wgrad(xd, dz, dw_old, db_old)
    assert rel(dx, dx_old.float().cpu()) < 1e-2
    assert rel(acc[:9 * C].view(C, 9), dw_old.t().cpu()) < 2e-3
    assert rel(acc[9 * C:], db_old.cpu()) < 2e-3


def test_accumulates_and_skips_bias():
    from emip_amd import ops
    B, H, W, C = 1, 22, 22, 64
    bf = torch.bfloat16
    x, dy = rnd(B, H, W, C, seed=1).to(bf).cuda(), rnd(B, H, W, C, seed=2).to(bf).cuda()
    wt = rnd(9, C, seed=3).cuda()
    dw = torch.zeros(9 * C, device="cuda:0")
    ops.dwconv3x3_bwd_fused(x, None, dy, wt, dw, None, False)
    once = dw.clone()
    ops.dwconv3x3_bwd_fused(x, None, dy, wt, dw, None, False)
    assert torch.allclose(dw, 2 * once, rtol=1e-5, atol=1e-5)
