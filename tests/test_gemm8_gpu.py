"""emip_gemm8 / emip_conv8 (the 8-wave LDS-DMA body) against plain PyTorch f32 references of the same op, through the
C ABI: every tile configuration on ragged shapes, every epilogue hook, the two-source K loop, implicit-GEMM convs with
channel tails / strides / padding, and the per-tap output-side LayerNorm of the spatial-reduction conv
(/root/reference/lib/pvt_v2.py:101-129: q = Linear(norm1(x)), x_ = sr(norm1(x)))."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
NCFG = 11


def dev():
    return torch.device("cuda:0")


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, device=dev(), generator=g) * scale


@pytest.mark.parametrize("cfg", range(1, NCFG + 1))
@pytest.mark.parametrize("M,N,K", [(333, 72, 64), (1000, 200, 128), (129, 320, 192), (64, 64, 64), (700, 328, 320)])
def test_dense_every_tile_on_ragged_shapes(cfg, M, N, K):
    from emip_amd import ops
    a = _rand(M, K, seed=1).to(torch.bfloat16)
    w = _rand(N, K, scale=K ** -0.5, seed=2).to(torch.bfloat16)
    bias = _rand(N, seed=3)
    out = torch.full((M + 2, N), 7.0, device=dev(), dtype=torch.bfloat16)     # guard rows: nothing beyond M may be written
    ops.gemm8(a, w, bias=bias, out=out[:M], cfg=cfg)
    ref = a.float() @ w.float().t() + bias
    assert (out[:M].float() - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    assert (out[M:] == 7.0).all()


@pytest.mark.parametrize("cfg", [0, 3, 5, 6, 9])
def test_dense_hooks_gelu_residual_stats_and_output_side_layernorm(cfg):
    from emip_amd import ops
    M, N, K = 1500, 320, 256
    x = (_rand(M, K, seed=4) * 2 + 30.0).to(torch.bfloat16)          # row means far from zero: the hard case of the lne form
    w = _rand(N, K, scale=K ** -0.5, seed=5).to(torch.bfloat16)
    bias, res = _rand(N, seed=6), _rand(M, N, seed=7).to(torch.bfloat16)
    xf = x.float()
    stats = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()
    colsum = w.float().sum(1).contiguous()
    eps = 1e-6
    ln = (xf - xf.mean(1, keepdim=True)) * torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + eps)
    ref = F.gelu(ln @ w.float().t() + bias) + res.float()
    out = torch.empty(M, N, device=dev(), dtype=torch.bfloat16)
    ost = torch.zeros(M, 2, device=dev())
    scratch = torch.ones(1000, device=dev())
    ops.gemm8(x, w, bias=bias, res=res, act=ops.ACT_GELU, out=out, ln_stats=stats, ln_eps=eps, colsum=colsum, out_stats=ost,
              zero=scratch, cfg=cfg)
    assert (out.float() - ref).abs().max().item() < 4e-2
    of = out.float()
    assert (ost[:, 0] - of.sum(1)).abs().max().item() < 2e-2 and (ost[:, 1] - (of * of).sum(1)).abs().max().item() < 0.3
    assert (scratch == 0).all()
    # in-place residual update (C aliases R), as the PVT blocks use it
    acc = res.clone()
    ops.gemm8(x, w, bias=bias, res=acc, act=ops.ACT_GELU, out=acc, ln_stats=stats, ln_eps=eps, colsum=colsum, cfg=cfg)
    assert torch.equal(acc, out)


@pytest.mark.parametrize("cfg", [0, 1, 3, 7])
def test_dense_two_source_k_loop(cfg):
    """gmflow/transformer.py:187-189: mlp(cat([source, message])) without materialising the concat"""
    from emip_amd import ops
    M, N, K1, K2 = 2100, 1024, 128, 128
    a1, a2 = _rand(M, K1, seed=8).to(torch.bfloat16), _rand(M, K2, seed=9).to(torch.bfloat16)
    w = _rand(N, K1 + K2, scale=(K1 + K2) ** -0.5, seed=10).to(torch.bfloat16)
    out = ops.gemm8(a1, w, a2=a2, act=ops.ACT_GELU, cfg=cfg)
    ref = F.gelu(torch.cat([a1, a2], 1).float() @ w.float().t())
    assert (out.float() - ref).abs().max().item() < 3e-2


def _conv_ref(x, wp, bias, k, stride, pad):
    Cout, Cin = wp.shape[0], x.shape[-1]
    return F.conv2d(x.float().permute(0, 3, 1, 2), wp.float().view(Cout, k, k, Cin).permute(0, 3, 1, 2), bias, stride=stride,
                    padding=pad).permute(0, 2, 3, 1)


@pytest.mark.parametrize("cfg", [0, 1, 3, 7, 9])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", [(2, 20, 20, 72, 40, 3, 1, 1), (1, 44, 44, 80, 136, 3, 1, 1),
                                                          (3, 22, 22, 128, 320, 3, 2, 1), (2, 16, 24, 64, 64, 2, 2, 0),
                                                          (1, 9, 11, 200, 96, 1, 1, 0)])
def test_conv_implicit_gemm(cfg, B, H, W, Cin, Cout, k, stride, pad):
    from emip_amd import ops
    x = _rand(B, H, W, Cin, seed=11).to(torch.bfloat16)
    wp = _rand(Cout, k * k * Cin, scale=(k * k * Cin) ** -0.5, seed=12).to(torch.bfloat16)
    bias = _rand(Cout, seed=13)
    ref = torch.relu(_conv_ref(x, wp, bias, k, stride, pad))
    out = ops.conv8(x, wp, k, k, stride, pad, bias=bias, act=ops.ACT_RELU, cfg=cfg)
    assert out.shape == ref.shape
    assert (out.float() - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,H,C,sr,cfg", [(4, 22, 320, 2, 0), (3, 44, 128, 4, 0), (2, 88, 64, 8, 0), (1, 22, 320, 2, 0),
                                          (32, 22, 320, 2, 10), (5, 22, 320, 2, 10)])
def test_spatial_reduction_conv_with_per_tap_layernorm(B, H, C, sr, cfg):
    """x_ = sr(norm1(x)) with gamma / beta folded into the weights: raw patches in, LayerNorm applied per tap on the output"""
    from emip_amd import ops
    x = (_rand(B, H, H, C, seed=14) * 1.5 + _rand(B, H, H, 1, seed=15) * 4).to(torch.bfloat16)
    wp = _rand(C, sr * sr * C, scale=(sr * sr * C) ** -0.5, seed=16).to(torch.bfloat16)
    bias = _rand(C, seed=17)
    xf = x.float()
    stats = torch.stack([xf.sum(-1), (xf * xf).sum(-1)], -1).reshape(-1, 2).contiguous()
    tapsum = wp.float().view(C, sr * sr, C).sum(2).t().contiguous()
    eps = 1e-6
    ln = (xf - xf.mean(-1, keepdim=True)) * torch.rsqrt(xf.var(-1, unbiased=False, keepdim=True) + eps)
    ref = _conv_ref(ln, wp, bias, sr, sr, 0)
    ost = torch.zeros(ref.numel() // C, 2, device=dev())
    # cfg 10: N = 320 in one 64 x 320 tile (the token panel read once)
    out = ops.conv8(x, wp, sr, sr, sr, 0, bias=bias, ln_stats=stats, tapsum=tapsum, ln_eps=eps, out_stats=ost, cfg=cfg)
    assert (out.float() - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())
    of = out.float().reshape(-1, C)
    assert (ost[:, 0] - of.sum(1)).abs().max().item() < 2e-2 * max(1.0, of.sum(1).abs().max().item())


def test_dispatch_routes_large_bf16_launches_to_the_same_arithmetic():
    """ops.gemm / ops.conv2d (the entry points the modules call) agree with the explicit 8-wave entry on a large launch and
    keep working on launches the 8-wave body does not take (K % 64 != 0, f32)"""
    from emip_amd import ops
    a = _rand(4096, 320, seed=18).to(torch.bfloat16)
    w = _rand(640, 320, scale=320 ** -0.5, seed=19).to(torch.bfloat16)
    bias = _rand(640, seed=20)
    assert torch.equal(ops.gemm(a, w, bias=bias), ops.gemm8(a, w, bias=bias))
    a2 = _rand(4096, 344, seed=21).to(torch.bfloat16)
    w2 = _rand(128, 344, scale=344 ** -0.5, seed=22).to(torch.bfloat16)
    ref = a2.float() @ w2.float().t()
    assert (ops.gemm(a2, w2).float() - ref).abs().max().item() < 3e-2
    a3, w3 = _rand(4096, 320, seed=23), _rand(64, 320, scale=320 ** -0.5, seed=24)
    assert (ops.gemm(a3, w3) - a3 @ w3.t()).abs().max().item() < 1e-3


@pytest.mark.parametrize("B,rps,K,N", [(8, 484, 320, 320), (5, 1936, 512, 128), (64, 121, 2048, 512)])
def test_gemm8_rowscale_epilogue(B, rps, K, N):
    """emip_gemm8_rs: res + scale[sample] * (a w^T + bias) -- stochastic depth (lib/pvt_v2.py:167-169) in the epilogue of
    the branch's last GEMM -- against torch on the rounded operands, with dropped samples (scale 0) in the batch"""
    from emip_amd import ops
    M = B * rps
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    sc = torch.floor(0.7 + torch.rand(B, generator=g)) / 0.7
    sc[1], sc[2] = 0.0, 1.0 / 0.7                              # at least one dropped and one kept sample
    sc = sc.cuda()
    out = ops.gemm_rowscale(a, w, bias, res, sc, rps)
    assert out is not None
    ref = res.float() + sc.repeat_interleave(rps).view(M, 1) * (a.float() @ w.float().t() + bias)
    err = (out.float() - ref).abs().max().item()
    assert err < 0.02 * ref.abs().max().item(), err
    drop = (sc.repeat_interleave(rps) == 0).nonzero()[:4, 0]
    assert torch.equal(out[drop], res[drop])                  # a dropped sample keeps its skip path bit for bit


@pytest.mark.parametrize("M,N,K,res,bias", [(30976, 128, 128, True, False), (30976, 128, 1024, True, False), (3000, 128, 256, False, True),
                                            (1100, 96, 64, True, True), (70000, 128, 128, True, False)])
def test_gemm8_layernorm_on_output(M, N, K, res, bias):
    """emip_gemm8_lno: R + LayerNorm(A W^T + b) * gamma + beta with the norm in the GEMM epilogue (GMFlow transformer.py:87-113)
    against an f32 restatement and against the two launches it replaces (GEMM, emip_layernorm with residual)"""
    from emip_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn(M, K, device=dev, generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5 * 2).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g) * 0.5 if bias else None
    r = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if res else None
    gamma = 1 + 0.2 * torch.randn(N, device=dev, generator=g)
    beta = 0.1 * torch.randn(N, device=dev, generator=g)
    y = a.float() @ w.float().t() + (b if bias else 0)
    ref = torch.nn.functional.layer_norm(y, (N,), gamma, beta, 1e-5) + (r.float() if res else 0)
    got = ops.gemm_ln_out(a, w, gamma, beta, 1e-5, bias=b, res=r)
    old = ops.layernorm(ops.gemm(a, w, bias=b), gamma, beta, 1e-5, res=r)
    torch.cuda.synchronize()
    top = max(1.0, ref.abs().max().item())
    e_new, e_old = (got.float() - ref).abs(), (old.float() - ref).abs()
    assert e_new.max().item() < 2e-2 * top, e_new.max().item()
    # one rounding (f32 accumulators normalised directly) instead of two: never worse than the two launches
    assert e_new.mean().item() <= 1.05 * e_old.mean().item() + 1e-5 * top
    # in place over the residual, as the transformer uses it
    if res:
        r2 = r.clone()
        ops.gemm_ln_out(a, w, gamma, beta, 1e-5, bias=b, res=r2, out=r2)
        assert torch.equal(r2, got)


@pytest.mark.parametrize("batch,M,N,K,shared_a", [(16, 8712, 128, 1984, True), (16, 1936, 968, 1152, False), (3, 1000, 72, 64, False)])
def test_strided_batched(batch, M, N, K, shared_a):
    """emip_gemm8_batched (the two per-image GEMMs of the factored conv_corr.0) against torch.bmm in f32 and the 4-wave form"""
    from emip_amd import ops
    g = torch.Generator(device="cuda").manual_seed(batch + M)
    a = torch.randn(1 if shared_a else batch, M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(batch, N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.full((batch, M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm8_batched(a, w, out, batch, M, N, K, K, K, N, 0 if shared_a else M * K, N * K, M * N, bias=bias, act=ops.ACT_RELU)
    old = torch.empty_like(out)
    ops.gemm_batched_bias(a, w, old, batch, M, N, K, K, K, N, 0 if shared_a else M * K, N * K, M * N, bias=bias, act=ops.ACT_RELU)
    ref = torch.relu(torch.matmul(a.float(), w.float().transpose(1, 2)) + bias)
    torch.cuda.synchronize()
    top = ref.abs().max().item()
    assert (out.float() - ref).abs().max().item() < 8e-3 * top
    assert (out.float() - old.float()).abs().max().item() <= 2.0 ** -7 * top
