"""emip_conv3x3_halo (direct 3 x 3 convolution on an LDS halo tile, InstanceNorm + ReLU of the producer applied on staging,
channel statistics in the epilogue; gmflow/backbone.py:39-69) against plain PyTorch f32 on the same bf16 operands and against
the launches it replaces (implicit-GEMM conv + chan_stats + chan_norm_apply)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, H, W, seed, C=64):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = (torch.randn(B, H, W, C, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    w = (torch.randn(C, C, 3, 3, device="cuda", generator=g) / (3.0 * C ** 0.5))
    wp = w.permute(0, 2, 3, 1).reshape(C, -1).to(torch.bfloat16).contiguous()         # pack_conv layout [Cout][kh][kw][Cin]
    return x, w, wp


def _ref_conv(xf, wp):
    C = wp.shape[0]
    w = wp.float().view(C, 3, 3, C).permute(0, 3, 1, 2)
    return torch.nn.functional.conv2d(xf.permute(0, 3, 1, 2), w, padding=1).permute(0, 2, 3, 1)


# (64 channels, multiples of 16: weights resident; multiples of 11 x 22: the streamed form, also for 96 and 128 channels)
@pytest.mark.parametrize("B,H,W,C", [(2, 176, 176, 64), (3, 32, 48, 64), (1, 16, 16, 64), (2, 33, 44, 64), (2, 88, 88, 96), (3, 22, 44, 96),
                                     (2, 44, 44, 128), (9, 44, 44, 128), (1, 11, 22, 128),
                                     (70, 32, 32, 64), (70, 22, 44, 96), (35, 22, 88, 128)])       # more tiles than workgroups: ranges cross images
def test_plain_conv_and_statistics(B, H, W, C):
    from emip_amd import ops
    assert ops.conv3x3_halo_eligible(B, H, W, C, C) and not ops.conv3x3_halo_eligible(B, H, W + 8, C, C)
    x, w, wp = _setup(B, H, W, 5 + B, C)
    pk = ops.conv3x3_halo_pack(wp)
    ws = ops.conv3x3_halo_ws(B, H, W, x.device, C)
    outs = []
    for _ in range(2):
        sums = torch.full((B, C, 2), -1.0, dtype=torch.float64, device="cuda")
        y = ops.conv3x3_halo(x, pk, out_sums=sums, ws=ws)
        outs.append((y, sums))
    y0 = ops.conv3x3_halo(x, pk)                                     # the instance without statistics
    torch.cuda.synchronize()
    (y, sums), (yb, sumsb) = outs
    assert torch.equal(y, yb) and torch.equal(sums, sumsb) and torch.equal(y, y0)
    assert int(ws[:4 * B].max()) == 0                                 # tickets back at zero
    ref = _ref_conv(x.float(), wp)
    top = ref.abs().max().item()
    err = (y.float() - ref).abs().max().item()
    old = ops.conv2d(x, wp, 3, 3, 1, 1)
    d_old = (y.float() - old.float()).abs().max().item()
    yf = y.double()
    s_ref = torch.stack((yf.sum((1, 2)), (yf * yf).sum((1, 2))), -1)
    es = ((sums - s_ref).abs() / (s_ref.abs() + 1.0)).max().item()
    print(f"  {B}x{H}x{W}x{C}: max |d| vs PyTorch {err:.4f} on values up to {top:.1f}; vs the implicit-GEMM conv {d_old:.4f}; sums rel {es:.1e}")
    assert err < 6e-3 * top and d_old <= 2.0 ** -7 * top and es < 1e-5


@pytest.mark.parametrize("B,H,W,C", [(2, 176, 176, 64), (2, 48, 32, 64), (2, 88, 88, 96), (5, 44, 44, 128), (2, 22, 22, 64),
                                     (70, 32, 32, 64), (70, 22, 44, 96), (35, 22, 88, 128)])
def test_normalise_on_staging_equals_the_separate_passes(B, H, W, C):
    """conv(relu(instance_norm(x))): in_sums from the producer's statistics; the zero padding applies to the NORMALISED tensor"""
    from emip_amd import ops
    x, w, wp = _setup(B, H, W, 11, C)
    pk = ops.conv3x3_halo_pack(wp)
    sums = torch.zeros((B, C, 2), dtype=torch.float64, device="cuda")
    ops.chan_stats(x, B, sums=sums)
    y = ops.conv3x3_halo(x, pk, in_sums=sums, in_eps=1e-5)
    xn = ops.chan_norm_apply(x.clone(), sums, B, 1e-5, relu_inner=True)            # what the separate pass stores (bf16)
    old = ops.conv2d(xn, wp, 3, 3, 1, 1)
    xf = x.float()
    mu = xf.mean((1, 2), keepdim=True)
    var = xf.var((1, 2), unbiased=False, keepdim=True)
    ref = _ref_conv(torch.relu((xf - mu) * torch.rsqrt(var + 1e-5)), wp)
    torch.cuda.synchronize()
    top = ref.abs().max().item()
    err, d_old = (y.float() - ref).abs().max().item(), (y.float() - old.float()).abs().max().item()
    print(f"  {B}x{H}x{W}x{C}: max |d| vs PyTorch {err:.4f}, vs apply + conv {d_old:.4f} on values up to {top:.1f}")
    assert err < 1.2e-2 * top and d_old < 1.2e-2 * top


@pytest.mark.parametrize("B,H,W", [(2, 352, 352), (3, 64, 96), (40, 64, 64)])
def test_stem_conv(B, H, W):
    """emip_conv_stem (7 x 7, stride 2, 3 -> 64 channels with the image stored as 8) against PyTorch f32, the implicit-GEMM conv
    it replaces (same bits) and a statistics pass over its output"""
    from emip_amd import ops
    assert ops.conv_stem_eligible(B, H, W, 8, 64) and not ops.conv_stem_eligible(B, H + 16, W, 8, 64)
    g = torch.Generator(device="cuda").manual_seed(B)
    x = torch.zeros(B, H, W, 8, device="cuda", dtype=torch.bfloat16)
    x[..., :3] = torch.randn(B, H, W, 3, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(64, 3, 7, 7, device="cuda", generator=g) / 12.0
    wp = torch.zeros(64, 7, 7, 8, device="cuda")
    wp[..., :3] = w.permute(0, 2, 3, 1)
    wp = wp.reshape(64, -1).to(torch.bfloat16).contiguous()
    pk = ops.conv_stem_pack(wp)
    ws = ops.conv3x3_halo_ws(B, H // 2, W // 2, x.device, 64)
    outs = []
    for _ in range(2):
        sums = torch.full((B, 64, 2), -1.0, dtype=torch.float64, device="cuda")
        outs.append((ops.conv_stem(x, pk, out_sums=sums, ws=ws), sums))
    y0 = ops.conv_stem(x, pk)
    old = ops.conv2d(x, wp, 7, 7, 2, 3)
    ref = torch.nn.functional.conv2d(x[..., :3].float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), stride=2, padding=3).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    (y, sums), (yb, sumsb) = outs
    assert torch.equal(y, yb) and torch.equal(sums, sumsb) and torch.equal(y, y0) and int(ws[:4 * B].max()) == 0
    top = ref.abs().max().item()
    err, d_old = (y.float() - ref).abs().max().item(), (y.float() - old.float()).abs().max().item()
    yf = y.double()
    s_ref = torch.stack((yf.sum((1, 2)), (yf * yf).sum((1, 2))), -1)
    es = ((sums - s_ref).abs() / (s_ref.abs() + 1.0)).max().item()
    print(f"  {B}x{H}x{W}: max |d| vs PyTorch {err:.4f} on values up to {top:.1f}; vs the implicit-GEMM conv {d_old:.4f}; sums rel {es:.1e}")
    assert err < 6e-3 * top and d_old <= 2.0 ** -7 * top and es < 1e-5
