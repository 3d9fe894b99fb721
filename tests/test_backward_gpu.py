"""Backward building blocks against torch autograd (f32 host reference on the rounded operands)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def prep(t, dtype):
    q = t.to(dtype)
    return q.to("cuda:0"), q.float()


def rel(out, ref):
    return ((out.float().cpu() - ref).abs().max() / (ref.abs().max() + 1e-9)).item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(15488, 320, 1280), (7744, 64, 256), (1000, 136, 72), (61952, 128, 128),
                                   (300, 8, 8), (5000, 1280, 320)])
def test_gemm_tn_wgrad(dtype, M, N, K):
    from emip_amd import ops
    dy, dyf = prep(rnd(M, N, seed=1), dtype)
    x, xf = prep(rnd(M, K, seed=2), dtype)
    c = ops.gemm_tn(dy, x)
    ref = dyf.t() @ xf
    assert c.dtype == torch.float32 and c.shape == (N, K)
    assert rel(c, ref) < (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_identity_asymmetric(dtype):
    from emip_amd import ops
    M, N, K = 256, 128, 128
    a = torch.zeros(M, N)
    a[:N] = torch.eye(N)                                     # dY = [I; 0]  ->  C = X[:N]
    b = ((torch.arange(M).view(M, 1) * 3 + torch.arange(K).view(1, K)) % 97).float()
    c = ops.gemm_tn(prep(a, dtype)[0], prep(b, dtype)[0])
    assert torch.equal(c.cpu(), b[:N])


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_dgrad_via_transposed_pack(dtype):
    """dX = dY @ W is the forward GEMM with the weight packed transposed ([K][N])"""
    from emip_amd import ops
    M, N, K = 3000, 320, 1280
    dy, dyf = prep(rnd(M, N, seed=1), dtype)
    w, wf = prep(rnd(N, K, seed=2, scale=1 / math.sqrt(K)), dtype)
    wt = w.t().contiguous()
    dx = ops.gemm(dy, wt)
    assert rel(dx, dyf @ wf) < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(7744, 64), (1936, 128), (485, 320), (121, 512),
                                 # the > 128-channel bf16 form: 16 / 32 lanes per row, 2 / 3 vectors per lane, ragged last vector
                                 (777, 256), (50, 384), (333, 640), (15488, 320), (19, 200)])
def test_layernorm_backward(dtype, M, C):
    from emip_amd import ops
    x, xf = prep(rnd(M, C, seed=1) * 2 + 0.3, dtype)
    dy, dyf = prep(rnd(M, C, seed=2), dtype)
    g = (1 + 0.1 * rnd(C, seed=3))
    b = 0.1 * rnd(C, seed=4)
    xr = xf.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), gr, br, 1e-6).backward(dyf)
    dg = torch.zeros(C, device="cuda:0")
    db = torch.zeros(C, device="cuda:0")
    dx = ops.layernorm_bwd(x, dy, g.cuda(), 1e-6, dg, db)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel(dx, xr.grad) < tol and rel(dg, gr.grad) < tol and rel(db, br.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_backward_many_rows_partial_accumulators(dtype):
    """many-row LayerNorm backward through the autograd entry, and the kernel's partial-accumulator mode (nparts = 32)"""
    from emip_amd import ops
    M, C = 30976, 320
    x, xf = prep(rnd(M, C, seed=11) * 2 + 0.3, dtype)
    dy, dyf = prep(rnd(M, C, seed=12), dtype)
    g = (1 + 0.1 * rnd(C, seed=13))
    xr = xf.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), torch.zeros(C, requires_grad=True)
    F.layer_norm(xr, (C,), gr, br, 1e-6).backward(dyf)
    dx, dg, db = ops.layernorm_bwd_fresh(x, dy, g.cuda(), 1e-6)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    assert rel(dx, xr.grad) < tol and rel(dg, gr.grad) < tol and rel(db, br.grad) < tol
    from emip_amd import _lib
    acc = torch.zeros(32, 2, C, device="cuda:0")
    dx2 = torch.empty_like(x)
    _lib.call("emip_layernorm_bwd", x.data_ptr(), C, dy.data_ptr(), C, dx2.data_ptr(), C, g.cuda().data_ptr(),
              acc.data_ptr(), acc[0, 1].data_ptr(), 32, 2 * C, M, C, 1e-6, 0 if dtype == torch.float32 else 1,
              torch.cuda.current_stream().cuda_stream)
    s = acc.sum(0)
    assert rel(dx2, xr.grad) < tol and rel(s[0], gr.grad) < tol and rel(s[1], br.grad) < tol


def _pack_conv_w(w):  # [Cout,Cin,kh,kw] -> [Cout, kh*kw*Cin]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [(2, 44, 44, 64, 96, 3, 1, 1), (2, 30, 26, 96, 128, 3, 2, 1),
                                                  (2, 88, 88, 64, 64, 8, 8, 0), (1, 44, 44, 1936, 72, 3, 1, 1),
                                                  (2, 64, 48, 8, 64, 7, 4, 3),
                                                  # >= 2048 output pixels: the LDS-DMA ring body (gemm_tn8.hip, CONV)
                                                  (4, 44, 44, 1936, 72, 3, 1, 1), (32, 22, 22, 320, 320, 2, 2, 0),
                                                  (6, 50, 37, 40, 24, 3, 1, 1), (5, 61, 47, 16, 136, 3, 2, 1)])
def test_conv_wgrad_and_dgrad(dtype, B, H, W, Cin, Cout, k, s, p):
    from emip_amd import ops
    x, xf = prep(rnd(B, H, W, Cin, seed=1), dtype)
    w4 = rnd(Cout, Cin, k, k, seed=2, scale=1 / math.sqrt(Cin * k * k)).to(dtype).float()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy, dyf = prep(rnd(B, Ho, Wo, Cout, seed=3), dtype)
    xr = xf.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w4.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, stride=s, padding=p).backward(dyf.permute(0, 3, 1, 2))
    dw = ops.conv2d_wgrad(dy, x, k, k, s, p)[0]
    tol = 1e-4 if dtype == torch.float32 else 4e-3
    assert rel(dw, _pack_conv_w(wr.grad)) < tol
    if s == 1:   # dgrad of a stride-1 conv = the forward conv with flipped, transposed weights
        wflip = w4.flip(2, 3).permute(1, 0, 2, 3).contiguous()            # [Cin, Cout, k, k]
        wd, _ = prep(_pack_conv_w(wflip), dtype)
        dx = ops.conv2d(dy, wd, k, k, 1, k - 1 - p)
        assert rel(dx.permute(0, 3, 1, 2), xr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_batched(dtype):
    from emip_amd import ops
    Bz, M, N, K = 6, 484, 128, 64
    a, af = prep(rnd(Bz, M, N, seed=1), dtype)
    b, bf = prep(rnd(Bz, M, K, seed=2), dtype)
    c = ops.gemm_tn_batched(a, b, Bz, M, N, K, N, K, M * N, M * K)
    assert rel(c, af.transpose(1, 2) @ bf) < (1e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_rows_fwd_bwd(dtype):
    from emip_amd import ops
    M, L, ld = 3000, 121, 128
    x, xf = prep(rnd(M, ld, seed=1) * 3, dtype)
    dp, dpf = prep(rnd(M, ld, seed=2), dtype)
    p = ops.softmax_rows(x, L, scale=0.125)
    ref = (xf[:, :L] * 0.125).softmax(-1)
    assert rel(p[:, :L], ref) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert p[:, L:].abs().max().item() == 0
    pf = p.float().cpu()[:, :L]
    ds = ops.softmax_bwd_rows(p, dp, L, scale=0.125)
    refds = pf * (dpf[:, :L] - (pf * dpf[:, :L]).sum(-1, keepdim=True)) * 0.125
    assert rel(ds[:, :L], refds) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert ds[:, L:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,L,ld,masked", [(1452, 484, 488, True), (700, 1936, 1936, False), (300, 484, 512, False), (64, 130, 136, True)])
def test_softmax_rows_wide_rows(dtype, M, L, ld, masked):
    """the 484-key window and 1936-key matching softmaxes of the GMFlow training path (transformer.py:33-70: scores / 128**0.5,
    -100 on the pairs a shifted window separates): the bf16 form reads 16 bytes per lane"""
    from emip_amd import ops
    x, xf = prep(rnd(M, ld, seed=1) * 3, dtype)
    dp, dpf = prep(rnd(M, ld, seed=2), dtype)
    gq = gk = None
    sc = xf[:, :L] * 0.25
    if masked:                       # 3 windows of `period` rows each, groups of 2 ids
        period, nwin = M // 3 if M % 3 == 0 else M, 3 if M % 3 == 0 else 1
        gq = torch.randint(0, 2, (nwin, period), dtype=torch.int32)
        gk = torch.randint(0, 2, (nwin, L), dtype=torch.int32)
        win = (torch.arange(M) // period) % nwin
        sc = sc + torch.where(gq[win, torch.arange(M) % period][:, None] != gk[win], -100.0, 0.0)
        p = ops.softmax_rows(x, L, scale=0.25, gid_q=gq.cuda(), gid_k=gk.cuda(), period=period, nwin=nwin)
    else:
        p = ops.softmax_rows(x, L, scale=0.25)
    ref = sc.softmax(-1)
    assert rel(p[:, :L], ref) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert L == ld or p[:, L:].abs().max().item() == 0
    pf = p.float().cpu()[:, :L]
    ds = ops.softmax_bwd_rows(p, dp, L, scale=0.25)
    refds = pf * (dpf[:, :L] - (pf * dpf[:, :L]).sum(-1, keepdim=True)) * 0.25
    assert rel(ds[:, :L], refds) < (1e-5 if dtype == torch.float32 else 2e-2)
    assert L == ld or ds[:, L:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_gelu_dwconv_wgrad(dtype):
    from emip_amd import ops
    k, kf = prep(rnd(6, 121, 64, seed=1), dtype)
    kt = ops.transpose_pad(k, 128)
    assert torch.equal(kt[:, :, :121].float().cpu(), kf.transpose(1, 2)) and kt[:, :, 121:].abs().max().item() == 0
    z, zf = prep(rnd(500, 256, seed=2) * 2, dtype)
    dy, dyf = prep(rnd(500, 256, seed=3), dtype)
    zr = zf.clone().requires_grad_(True)
    F.gelu(zr).backward(dyf)
    assert rel(ops.gelu_bwd(z, dy), zr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    B, H, W, C = 2, 22, 22, 320
    x, xf = prep(rnd(B, H, W, C, seed=4), dtype)
    g, gf = prep(rnd(B, H, W, C, seed=5), dtype)
    w = rnd(C, 1, 3, 3, seed=6).requires_grad_(True)
    b = rnd(C, seed=7).requires_grad_(True)
    F.conv2d(xf.permute(0, 3, 1, 2), w, b, padding=1, groups=C).backward(gf.permute(0, 3, 1, 2))
    dw = torch.zeros(9, C, device="cuda:0")
    db = torch.zeros(C, device="cuda:0")
    ops.dwconv3x3_wgrad(x, g, dw, db)
    assert rel(dw, w.grad.view(C, 9).t()) < (1e-4 if dtype == torch.float32 else 5e-3)
    assert rel(db, b.grad) < (1e-4 if dtype == torch.float32 else 5e-3)
    # dgrad of the depthwise conv = the forward depthwise kernel with the taps reversed
    wt = w.detach().view(C, 9).flip(1).t().contiguous().cuda()
    xr = xf.permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.conv2d(xr, w.detach(), None, padding=1, groups=C).backward(gf.permute(0, 3, 1, 2))
    dx = ops.dwconv3x3(g, wt)
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_train_backward(dtype):
    from emip_amd import ops
    B, H, W, C = 3, 20, 22, 96
    x, xf = prep(rnd(B, H, W, C, seed=1) * 2 + 0.5, dtype)
    dy, dyf = prep(rnd(B, H, W, C, seed=2), dtype)
    g = (1 + 0.1 * rnd(C, seed=3))
    bb = 0.1 * rnd(C, seed=4)
    xr = xf.permute(0, 3, 1, 2).clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), bb.clone().requires_grad_(True)
    out_ref = F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.0, 1e-5))
    out_ref.backward(dyf.permute(0, 3, 1, 2))
    sums = ops.chan_stats(x, 1)
    out = ops.chan_norm_apply(x, sums, 1, 1e-5, relu_inner=True, gamma=g.cuda(), beta=bb.cuda())
    dg = torch.zeros(C, device="cuda:0")
    db = torch.zeros(C, device="cuda:0")
    dx = ops.bn_train_bwd(x, dy, out, sums, g.cuda(), dg, db, 1e-5)
    tol = 2e-4 if dtype == torch.float32 else 3e-2
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < tol and rel(dg, gr.grad) < tol and rel(db, br.grad) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_bilinear_backward(dtype):
    from emip_amd import ops
    x = rnd(2, 32, 11, 11, seed=1).requires_grad_(True)
    dy, dyf = prep(rnd(2, 22, 22, 32, seed=2), dtype)
    F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True).backward(dyf.permute(0, 3, 1, 2))
    dx = ops.bilinear_bwd(dy, 11, 11, True)
    assert rel(dx.permute(0, 3, 1, 2), x.grad) < 1e-5
    pc = rnd(2, 1, 44, 44, seed=3).requires_grad_(True)
    g = rnd(2, 1, 352, 352, seed=4)
    F.interpolate(pc, scale_factor=8, mode="bilinear").backward(g)
    dpc = ops.bilinear_planar_bwd(g.cuda(), 44, 44, False)
    assert rel(dpc.permute(0, 3, 1, 2), pc.grad) < 1e-5


def test_deferred_weight_gradients_grouped_launch():
    """ops.WgradQueue: inside the gradient arena gemm_tn defers; one grouped launch (emip_gemm_tn8_group) then produces every
    weight gradient and bias gradient -- shapes with tails in all three dimensions, a strided dY, problems of very different
    length in one group -- exactly what separate launches give"""
    from emip_amd import ops
    shapes = [(15488, 320, 1280, True), (7744, 640, 320, False), (5003, 328, 200, True), (123904, 128, 128, True),
              (2304, 72, 1936, False), (30976, 1280, 320, True)]
    ops_ = []
    for i, (M, N, K, bias) in enumerate(shapes):
        dy, dyf = prep(rnd(M, N + (8 if i == 2 else 0), seed=10 + i), torch.bfloat16)
        x, xf = prep(rnd(M, K, seed=20 + i), torch.bfloat16)
        ops_.append((dy[:, :N], dyf[:, :N], x, xf, bias))
    dev = ops_[0][0].device
    for rep in range(2):                  # the first pass sizes the arena (everything falls back to torch.zeros)
        ops.ARENA.begin(dev)
        outs = []
        try:
            for dy, _, x, _, bias in ops_:
                outs.append(ops.gemm_tn(dy, x, with_colsum=True, defer=True) if bias else (ops.gemm_tn(dy, x, defer=True), None))
            assert len(ops.WGRADS.items) == len(shapes)            # nothing has been launched yet
            ops.flush_wgrads()
        finally:
            ops.ARENA.end()
        for (dy, dyf, x, xf, bias), (c, db) in zip(ops_, outs):
            assert rel(c, dyf.t() @ xf) < 2e-3
            if bias:
                assert rel(db, dyf.sum(0)) < 2e-3
