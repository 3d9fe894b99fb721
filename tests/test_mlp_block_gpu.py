"""emip_mlp_block (the whole Mlp half of a stage-3 PVTv2 block in one launch) against a plain PyTorch f32 evaluation of
lib/pvt_v2.py:45-54,165-169 on the same bf16-rounded operands, against the two launches it replaces, and inside Block.run_fused."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(B, H, W, seed, C=320, N=1280):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).cuda()
    x = (r(B, H, W, C) * 1.3 + 0.2).to(torch.bfloat16)
    w1 = r(N, C, k=C ** -0.5).to(torch.bfloat16)
    w2 = r(C, N, k=N ** -0.5).to(torch.bfloat16)
    b1, b2, bd = r(N, k=0.1), r(C, k=0.1), r(N, k=0.1)
    wd = r(9, N, k=0.3)
    xf = x.float().view(-1, C)
    stats = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()
    colsum = w1.float().sum(1).contiguous()
    return x, w1, w2, b1, b2, bd, wd, stats, colsum


def _reference(x, w1, w2, b1, b2, bd, wd, eps):
    """f32 arithmetic with the kernel's rounding points: H and G stored as bf16, exact-erf GELU"""
    B, H, W, C = x.shape
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = (xf * xf).mean(-1, keepdim=True) - mu * mu
    xn = (xf - mu) * torch.rsqrt(var.clamp_min(0) + eps)
    h = (xn @ w1.float().t() + b1).to(torch.bfloat16).float()                      # [B,H,W,N]
    N = h.shape[-1]
    hp = h.permute(0, 3, 1, 2)
    z = F.conv2d(hp, wd.t().reshape(N, 1, 3, 3), bd, padding=1, groups=N).permute(0, 2, 3, 1)
    gact = F.gelu(z).to(torch.bfloat16).float()
    return xf + gact @ w2.float().t() + b2


@pytest.mark.parametrize("B,H,W", [(8, 22, 22), (3, 22, 22), (2, 7, 22), (5, 22, 14)])
def test_against_pytorch_and_the_two_launch_path(B, H, W):
    from emip_amd import ops
    assert ops.mlp_block_eligible(B, H, W, 320, 1280)
    x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, H, W, 1 + B)
    eps = 1e-6
    cst = ops.mlp_block_consts(wd, bd, b1, colsum)
    out = torch.full_like(x, 7.0)
    ost = torch.full((B * H * W, 2), -1.0, device="cuda")
    ops.mlp_block(x, w1, w2, cst, b2, stats.view(-1), eps, out, out_stats=ost)
    ref = _reference(x, w1, w2, b1, b2, bd, wd, eps)
    top = ref.abs().max().item()
    err = (out.float() - ref).abs().max().item()
    # the statistics are those of the STORED (rounded) rows
    of = out.float().view(-1, 320)
    assert torch.allclose(ost[:, 0], of.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(ost[:, 1], (of * of).sum(1), rtol=1e-4, atol=1e-2)
    msg = f"  B={B} {H}x{W}: max |d| vs PyTorch {err:.4f} on values up to {top:.1f}"
    if ops.mlp_fc1dw_eligible(B, H, W, 320, 1280):
        t = ops.mlp_fc1dw(x, w1, b1, colsum, stats.view(-1), eps, wd, bd)
        two = ops.gemm(t, w2, bias=b2, res=x)
        d2 = (out.float() - two.float()).abs().max().item()
        msg += f", vs emip_mlp_fc1dw + GEMM {d2:.4f}"
        assert d2 <= 2.0 ** -7 * top + 1e-3              # the same rounding points: at most an output ulp or two apart
    print(msg)
    assert err < 1.5e-2 * top, (err, top)                 # bf16 output rounding (2^-8 relative) + the polynomial GELU


def test_block_run_fused_with_and_without_the_one_launch_mlp():
    from emip_amd.lib import pvt_v2
    from emip_amd import _lib, nn_base
    torch.manual_seed(5)
    prev = nn_base.get_default_dtype()
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        blk = pvt_v2.Block(dim=320, num_heads=5, mlp_ratio=4, qkv_bias=True, sr_ratio=2).cuda().eval()
        B, H, W, C = 8, 22, 22, 320
        x0 = (torch.randn(B, H, W, C, device="cuda") * 1.2).to(torch.bfloat16)
        outs = []
        keep_band, pvt_v2.MLP_BAND = pvt_v2.MLP_BAND, False      # (round 4's emip_mlp_band would take the launch: tests/test_mlp_band_gpu.py)
        for flag in (True, False):
            pvt_v2.MLP_BLOCK = flag
            try:
                x = x0.clone()
                xf = x.float().view(-1, C)
                stats = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous().view(-1)
                buf = torch.zeros(pvt_v2.Block.scratch_floats(B, H, W, C, 2), device="cuda")
                rec = []
                _lib.profile(rec)
                y, st, _ = blk.run_fused(x, stats, buf, torch.empty_like(x))
                _lib.profile(None)
                torch.cuda.synchronize()
                names = [r[0] for r in rec]
                assert ("emip_mlp_block" in names) == flag and ("emip_mlp_fc1dw" in names) != flag, names
                outs.append((y.float().clone(), st.clone()))
            finally:
                pvt_v2.MLP_BLOCK = False
                _lib.profile(None)
        pvt_v2.MLP_BAND = keep_band
        (a, sa), (b, sb) = outs
        top = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() < 2e-2 * top and (a - b).abs().mean().item() < 1e-3 * top
        assert torch.allclose(sa, sb, rtol=2e-2, atol=2e-2 * top * 320)
    finally:
        nn_base.set_default_dtype(prev)
