"""emip_mlp_band (the Mlp half of a 22 x 22-stage PVTv2 block per quarter image, three-stage chunk pipeline) against a plain
PyTorch f32 evaluation of lib/pvt_v2.py:45-54,165-169 on the same bf16-rounded operands, against the two launches it replaces,
bit-for-bit against itself (no atomics anywhere), and inside Block.run_fused."""
import pytest
import torch

from test_mlp_block_gpu import _reference, _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bands", [4, 8])
@pytest.mark.parametrize("B", [8, 3, 1, 16])
def test_against_pytorch_and_the_two_launch_path(B, bands):
    from emip_amd import ops
    H = W = 22
    assert ops.mlp_band_eligible(B, H, W, 320, 1280) and not ops.mlp_band_eligible(B, 22, 14, 320, 1280)
    x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, H, W, 11 + B)
    eps = 1e-6
    stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
    outs = []
    for _ in range(2):
        out = torch.full_like(x, 7.0)
        ost = torch.full((B * H * W, 2), -1.0, device="cuda")
        ops.mlp_band(x, stg, taps, b2, stats.view(-1), eps, out, out_stats=ost, bands=bands)
        outs.append((out, ost))
    other = torch.empty_like(x)
    ost_o = torch.empty_like(ost)
    ops.mlp_band(x, stg, taps, b2, stats.view(-1), eps, other, out_stats=ost_o, bands=12 - bands)
    torch.cuda.synchronize()
    (out, ost), (out_b, ost_b) = outs
    assert torch.equal(out, out_b) and torch.equal(ost, ost_b)          # fixed-order reductions only: reproducible bit for bit
    assert torch.equal(out, other) and torch.equal(ost, ost_o)          # ... and the same bits from quarter- and eighth-image workgroups
    ref = _reference(x, w1, w2, b1, b2, bd, wd, eps)
    top = ref.abs().max().item()
    d = (out.float() - ref).abs()
    err = d.max().item()
    # per band: a wrong halo or a wrong tile would show as a band-shaped error
    per_band = d.view(B, 4, 121, 320).amax((0, 2, 3)).tolist()
    of = out.float().view(-1, 320)
    assert torch.allclose(ost[:, 0], of.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(ost[:, 1], (of * of).sum(1), rtol=1e-4, atol=1e-2)
    t = ops.mlp_fc1dw(x, w1, b1, colsum, stats.view(-1), eps, wd, bd)
    two = ops.gemm(t, w2, bias=b2, res=x)
    d2 = (out.float() - two.float()).abs().max().item()
    print(f"  B={B}: max |d| vs PyTorch {err:.4f} on values up to {top:.1f} (per band {['%.4f' % v for v in per_band]}), "
          f"vs emip_mlp_fc1dw + GEMM {d2:.4f}")
    assert d2 <= 2.0 ** -6 * top + 1e-3                  # the same rounding points, another summation order inside the MFMAs
    assert err < 1.5e-2 * top, (err, top)                 # bf16 output rounding (2^-8 relative) + the polynomial GELU


@pytest.mark.parametrize("bands", [4, 8])
def test_strided_rows_and_missing_statistics_output(bands):
    """token rows embedded in a wider buffer (row stride > 320) on both sides; out_stats NULL"""
    from emip_amd import ops
    B, H, W = 2, 22, 22
    x, w1, w2, b1, b2, bd, wd, stats, colsum = _setup(B, H, W, 5)
    stg, taps = ops.mlp_band_packs(w1, b1, colsum, w2, wd, bd)
    wide = torch.zeros((B, H, W, 384), dtype=torch.bfloat16, device="cuda")
    wide[..., 32:352] = x
    xs = wide[..., 32:352]
    owide = torch.full((B, H, W, 336), 3.0, dtype=torch.bfloat16, device="cuda")
    os_ = owide[..., 8:328]
    ops.mlp_band(xs, stg, taps, b2, stats.view(-1), 1e-6, os_, out_stats=None, bands=bands)
    out = torch.empty_like(x)
    ops.mlp_band(x, stg, taps, b2, stats.view(-1), 1e-6, out, out_stats=None, bands=bands)
    torch.cuda.synchronize()
    assert torch.equal(os_, out)
    assert (owide[..., :8] == 3.0).all() and (owide[..., 328:] == 3.0).all()      # nothing written outside the 320 channels


def test_block_run_fused_with_and_without_the_band_launch():
    from emip_amd import _lib, nn_base
    from emip_amd.lib import pvt_v2
    torch.manual_seed(5)
    prev = nn_base.get_default_dtype()
    nn_base.set_default_dtype(torch.bfloat16)
    keep = pvt_v2.MLP_BAND
    try:
        blk = pvt_v2.Block(dim=320, num_heads=5, mlp_ratio=4, qkv_bias=True, sr_ratio=2).cuda().eval()
        B, H, W, C = 8, 22, 22, 320
        x0 = (torch.randn(B, H, W, C, device="cuda") * 1.2).to(torch.bfloat16)
        outs = []
        for flag in (True, False):
            pvt_v2.MLP_BAND = flag
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
            assert ("emip_mlp_band" in names) == flag and ("emip_mlp_fc1dw" in names) != flag, names
            outs.append((y.float().clone(), st.clone()))
        (a, sa), (b, sb) = outs
        top = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() < 2e-2 * top and (a - b).abs().mean().item() < 1e-3 * top
        assert torch.allclose(sa, sb, rtol=2e-2, atol=2e-2 * top * 320)
    finally:
        pvt_v2.MLP_BAND = keep
        _lib.profile(None)
        nn_base.set_default_dtype(prev)
