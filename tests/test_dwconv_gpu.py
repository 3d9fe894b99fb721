"""The LDS-tiled bf16 depthwise 3x3 (dwconv.hip, reached through emip_dwconv3x3) against torch's f32 depthwise conv:
/root/reference/lib/pvt_v2.py:316-327 + the GELU of :50-51, on the four PVT stage shapes, band edges, channel slices."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,C,act", [(4, 22, 22, 1280, 2), (2, 44, 44, 512, 2), (1, 88, 88, 256, 2), (2, 11, 11, 2048, 2),
                                         (1, 5, 7, 64, 0), (2, 44, 44, 128, 0), (1, 30, 100, 64, 1), (3, 1, 1, 64, 2)])
def test_tiled_depthwise_matches_torch(B, H, W, C, act):
    from emip_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(H * W + C)
    wide = torch.randn(B, H, W, C + 64, device=dev, generator=g).to(torch.bfloat16)
    x = wide[..., 32:32 + C]                                  # a channel slice of a wider buffer: ldx > C
    w = torch.randn(C, 1, 3, 3, device=dev, generator=g) * 0.4
    b = torch.randn(C, device=dev, generator=g)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, b, padding=1, groups=C).permute(0, 2, 3, 1)
    if act == 2:
        ref = F.gelu(ref)
    elif act == 1:
        ref = torch.relu(ref)
    out = torch.full((B, H, W, C + 8), 5.0, device=dev, dtype=torch.bfloat16)
    ops.dwconv3x3(x, w.reshape(C, 9).t().contiguous(), b, act=act, out=out[..., :C])
    assert (out[..., :C].float() - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    assert (out[..., C:] == 5.0).all()
