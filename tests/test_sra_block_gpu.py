"""emip_sra_block -- q projection (norm1 folded, applied from the row statistics) + spatial-reduction attention + proj +
residual in one launch (/root/reference/lib/pvt_v2.py:95-127, 165-168) -- against a plain PyTorch f32 restatement on the
rounded operands and against the three launches it replaces (emip_gemm8 with the output-side LayerNorm, emip_sra_attention,
emip_gemm8 with residual and row statistics), on the three PVT stages that have a spatial reduction and on ragged shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

EPS = 1e-6


def _make(B, H, W, C, Lk, seed):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = (r(B, H, W, C) * 1.3 + 0.2 * r(1, 1, 1, C)).to(torch.bfloat16)
    kv = (r(B, Lk, 2 * C) * 1.2).to(torch.bfloat16)
    kv[0, 0, :64] *= 4.0                                # a dominant key: peaked rows next to flat ones
    wq = (r(C, C) / C ** 0.5 * 1.5).to(torch.bfloat16)  # gamma already folded in
    wp = (r(C, C) / C ** 0.5).to(torch.bfloat16)
    bq, bp = r(C) * 0.1, r(C) * 0.1
    xf = x.float().view(-1, C)
    stats = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()
    return x, kv, wq, wp, bq.contiguous(), bp.contiguous(), stats


def _ref(x, kv, wq, wp, bq, bp, heads, scale):
    B, H, W, C = x.shape
    N, Lk = H * W, kv.shape[1]
    xf = x.float().view(B, N, C)
    mu = xf.mean(-1, keepdim=True)
    var = (xf * xf).mean(-1, keepdim=True) - mu * mu
    xh = (xf - mu) * torch.rsqrt(var.clamp_min(0) + EPS)
    q = (xh @ wq.float().t() + bq).to(torch.bfloat16).float()
    qh = q.view(B, N, heads, 64).permute(0, 2, 1, 3)
    k = kv.float()[..., :C].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    v = kv.float()[..., C:].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    a = torch.softmax(qh @ k.transpose(-1, -2) * scale, -1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B, N, C).to(torch.bfloat16).float()
    return (o @ wp.float().t() + bp + xf).view(B, H, W, C)


@pytest.mark.parametrize("B,H,W,C,Lk", [(4, 22, 22, 320, 121), (8, 22, 22, 320, 121), (2, 44, 44, 128, 121), (1, 88, 88, 64, 121),
                                         (3, 10, 10, 320, 25), (2, 13, 11, 128, 128), (1, 1, 1, 64, 1), (16, 22, 22, 320, 121)])
def test_sra_block_matches_reference_and_the_three_launches(B, H, W, C, Lk):
    from emip_amd import ops
    heads, scale = C // 64, 0.125
    x, kv, wq, wp, bq, bp, stats = _make(B, H, W, C, Lk, 7 * C + H + Lk)
    ref = _ref(x, kv, wq, wp, bq, bp, heads, scale)
    M = B * H * W
    # the three launches
    csq = wq.float().sum(1).contiguous()
    q = ops.gemm(x, wq, bias=bq, ln_stats=stats, ln_eps=EPS, colsum=csq)
    att = torch.empty_like(x)
    ops.sra_attention(q.view(B, H * W, C), kv, att.view(B, H * W, C), B, heads, H * W, Lk, scale)
    st_old = torch.zeros(M, 2, device=x.device)
    old = x.clone()
    ops.gemm(att, wp, bias=bp, res=old, out=old, out_stats=st_old.view(-1))
    # one launch, in place, with the bit-swapped packs
    sw = ops.swap23(C, x.device)
    got = x.clone()
    st_new = torch.zeros(M, 2, device=x.device)
    ops.sra_block(got, stats, EPS, wq[sw].contiguous(), bq, csq, kv, wp[sw][:, sw].contiguous(), bp, heads, scale,
                  out_stats=st_new.view(-1))
    torch.cuda.synchronize()
    top = max(1.0, ref.abs().max().item())
    e_new = (got.float() - ref).abs()
    e_old = (old.float() - ref).abs()
    assert e_new.max().item() < 3e-2 * top, (e_new.max().item(), top)
    # same rounding class as the launches it replaces: mean error within 10 % of theirs (+ one part in 1e4 of the range)
    assert e_new.mean().item() <= 1.1 * e_old.mean().item() + 1e-4 * top, (e_new.mean().item(), e_old.mean().item())
    assert (got.float() - old.float()).abs().max().item() < 3e-2 * top
    # the row statistics are those of the rows it stored
    gf = got.float().view(M, C)
    want = torch.stack([gf.sum(1), (gf * gf).sum(1)], 1)
    assert torch.allclose(st_new, want, rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("B,H,W,Lk", [(4, 22, 22, 121), (16, 22, 22, 121), (3, 10, 10, 25), (1, 1, 1, 128), (2, 13, 11, 77)])
def test_sra_qattn_matches_reference_and_the_two_launches(B, H, W, Lk):
    """emip_sra_qattn (q projection inside the attention launch, one head per workgroup, C = 320) against the f32 restatement
    and against emip_gemm8 (output-side LayerNorm) + emip_sra_attention"""
    from emip_amd import ops
    C, heads, scale = 320, 5, 0.125
    x, kv, wq, wp, bq, bp, stats = _make(B, H, W, C, Lk, H + Lk)
    csq = wq.float().sum(1).contiguous()
    q = ops.gemm(x, wq, bias=bq, ln_stats=stats, ln_eps=EPS, colsum=csq)
    old = torch.empty_like(x)
    ops.sra_attention(q.view(B, H * W, C), kv, old.view(B, H * W, C), B, heads, H * W, Lk, scale)
    got = ops.sra_qattn(x, stats, EPS, wq[ops.swap23(C, x.device)].contiguous(), bq, csq, kv, heads, scale)
    torch.cuda.synchronize()
    xf = x.float().view(B, H * W, C)
    mu = xf.mean(-1, keepdim=True)
    xh = (xf - mu) * torch.rsqrt(((xf * xf).mean(-1, keepdim=True) - mu * mu).clamp_min(0) + EPS)
    qr = (xh @ wq.float().t() + bq).to(torch.bfloat16).float().view(B, H * W, heads, 64).permute(0, 2, 1, 3)
    k = kv.float()[..., :C].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    v = kv.float()[..., C:].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    ref = (torch.softmax(qr @ k.transpose(-1, -2) * scale, -1) @ v).permute(0, 2, 1, 3).reshape(B, H, W, C)
    top = max(1.0, ref.abs().max().item())
    e_new, e_old = (got.float() - ref).abs(), (old.float() - ref).abs()
    assert e_new.max().item() < 3e-2 * top
    assert e_new.mean().item() <= 1.1 * e_old.mean().item() + 1e-4 * top


def test_sra_block_writes_nothing_beyond_its_rows_and_strided_rows():
    """rows of a wider buffer (ldx > C): the columns beside them and the rows behind them stay untouched"""
    from emip_amd import ops
    B, H, W, C, Lk = 2, 9, 5, 128, 30
    x, kv, wq, wp, bq, bp, stats = _make(B, H, W, C, Lk, 5)
    ref = _ref(x, kv, wq, wp, bq, bp, C // 64, 0.125)
    wide = torch.full((B * H * W + 64, C + 64), 9.0, device=x.device, dtype=torch.bfloat16)
    view = wide[:B * H * W, :C]
    view.copy_(x.view(-1, C))
    sw = ops.swap23(C, x.device)
    st = torch.zeros(B * H * W, 2, device=x.device)
    ops.sra_block(view.view(B, H, W, C), stats, EPS, wq[sw].contiguous(), bq, wq.float().sum(1).contiguous(), kv,
                  wp[sw][:, sw].contiguous(), bp, C // 64, 0.125, out_stats=st.view(-1))
    torch.cuda.synchronize()
    assert (wide[:, C:] == 9.0).all() and (wide[B * H * W:] == 9.0).all()
    assert (view.float().view(B, H, W, C) - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())


def test_pvt_block_fused_path_uses_it_and_matches_the_unfused_launches():
    """Block.run_fused with and without emip_sra_block on a stage-3 block: same tokens and statistics up to bf16 rounding"""
    from emip_amd.lib import pvt_v2
    from emip_amd import _lib, nn_base
    torch.manual_seed(3)
    prev = nn_base.get_default_dtype()
    nn_base.set_default_dtype(torch.bfloat16)
    blk = pvt_v2.Block(dim=320, num_heads=5, mlp_ratio=4, qkv_bias=True, sr_ratio=2).cuda().eval()
    B, H, W, C = 4, 22, 22, 320
    x0 = (torch.randn(B, H, W, C, device="cuda") * 1.2).to(torch.bfloat16)
    outs = []
    wide, pvt_v2.SRA_BLOCK_WIDE_ROWS = pvt_v2.SRA_BLOCK_WIDE_ROWS, 10 ** 9      # the q + attention launch form (round 4's default is emip_sra_block)
    for flag in (True, False):
        pvt_v2.SRA_FUSED = flag
        try:
            x = x0.clone()
            xf = x.float().view(-1, C)
            stats = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous().view(-1)
            buf = torch.zeros(pvt_v2.Block.scratch_floats(B, H, W, C, 2), device="cuda")
            rec = []
            _lib.profile(rec)
            y, st, _ = blk.run_fused(x, stats, buf)
            _lib.profile(None)
            torch.cuda.synchronize()
            names = [r[0] for r in rec]
            assert ("emip_sra_qattn" in names) == flag and ("emip_sra_attention" in names) != flag, names
            outs.append((y.float().clone(), st.clone()))
        finally:
            pvt_v2.SRA_FUSED = True
            _lib.profile(None)
    pvt_v2.SRA_BLOCK_WIDE_ROWS = wide
    nn_base.set_default_dtype(prev)
    (a, sa), (b, sb) = outs
    top = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() < 4e-2 * top
    assert (a - b).abs().mean().item() < 2e-3 * top
