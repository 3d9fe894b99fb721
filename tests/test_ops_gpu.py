"""Per-kernel parity: each C-ABI entry point against a plain PyTorch fp32 reference of the same op
(computed on the host), in f32 (parity mode) and bf16 (performance mode; inputs are rounded to
bf16 first so that the reference sees the same operands)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def prep(t, dtype):
    """round to the storage dtype; returns (device tensor in dtype, fp32 host copy of the rounded values)"""
    q = t.to(dtype)
    return q.to(dev()), q.float()


def tol(dtype, f32=1e-4, bf16=2e-2):
    return f32 if dtype == torch.float32 else bf16


def check(out, ref, dtype, f32=1e-4, bf16=2e-2, name=""):
    out = out.float().cpu()
    scale = ref.abs().max().item() + 1e-6
    err = (out - ref).abs().max().item() / scale
    assert err <= tol(dtype, f32, bf16), f"{name} rel-to-max err {err:.3e} (scale {scale:.3g})"


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_identity_asymmetric(dtype):
    """A = I with an asymmetric integer W: catches any row/column or k-order mix-up exactly."""
    from emip_amd import ops
    n = 128
    a = torch.eye(n)
    w = (torch.arange(n).view(n, 1) * 2 + torch.arange(n).view(1, n) % 7).float() % 64
    ad, _ = prep(a, dtype)
    wd, _ = prep(w, dtype)
    out = ops.gemm(ad, wd)
    assert torch.equal(out.float().cpu(), w.t().contiguous())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(7744, 64, 64), (1936, 256, 128), (300, 320, 1280), (121, 1024, 512), (5, 1, 96),
                                   (130, 576, 256), (2000, 128, 344)])
def test_gemm_epilogues(dtype, M, N, K):
    from emip_amd import ops
    a, af = prep(rnd(M, K, seed=1), dtype)
    w, wf = prep(rnd(N, K, seed=2, scale=1 / math.sqrt(K)), dtype)
    r, rf = prep(rnd(M, N, seed=3), dtype)
    b = rnd(N, seed=4).to(dev())
    check(ops.gemm(a, w), af @ wf.t(), dtype, name="plain")
    check(ops.gemm(a, w, bias=b, act=ops.ACT_GELU), F.gelu(af @ wf.t() + b.cpu()), dtype, name="bias+gelu")
    check(ops.gemm(a, w, bias=b, res=r), af @ wf.t() + b.cpu() + rf, dtype, name="bias+res")
    check(ops.gemm(a, w, act=ops.ACT_RELU), F.relu(af @ wf.t()), dtype, name="relu")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_concat_k_and_slices(dtype):
    from emip_amd import ops
    M = 1000
    a1, a1f = prep(rnd(M, 128, seed=1), dtype)
    a2, a2f = prep(rnd(M, 128, seed=2), dtype)
    w, wf = prep(rnd(1024, 256, seed=3, scale=1 / 16), dtype)
    check(ops.gemm(a1, w, a2=a2, act=ops.ACT_GELU), F.gelu(torch.cat([a1f, a2f], 1) @ wf.t()), dtype, name="catK")
    # read a channel slice, write into a channel slice of a wider buffer
    wide, widef = prep(rnd(M, 256, seed=5), dtype)
    w2, w2f = prep(rnd(64, 128, seed=6, scale=1 / 11), dtype)
    dst = torch.zeros(M, 96, dtype=dtype, device=dev())
    ops.gemm(wide[:, 128:], w2, out=dst[:, 32:])
    check(dst[:, 32:], widef[:, 128:] @ w2f.t(), dtype, name="slices")
    assert dst[:, :32].abs().max().item() == 0


def _pack_conv_w(w):  # [Cout,Cin,kh,kw] -> [Cout, kh*kw*Cin]
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [
    (2, 44, 44, 64, 96, 3, 1, 1), (1, 88, 88, 8, 64, 7, 4, 3), (2, 30, 26, 96, 128, 3, 2, 1),
    (1, 88, 88, 64, 64, 8, 8, 0), (2, 22, 22, 128, 40, 1, 1, 0), (1, 44, 44, 1936, 72, 3, 1, 1),
    (3, 11, 11, 32, 32, 3, 1, 1), (1, 64, 48, 8, 64, 7, 2, 3), (2, 20, 20, 64, 32, 1, 2, 0)])
def test_conv2d(dtype, B, H, W, Cin, Cout, k, s, p):
    from emip_amd import ops
    x, xf = prep(rnd(B, H, W, Cin, seed=1), dtype)
    w4 = rnd(Cout, Cin, k, k, seed=2, scale=1 / math.sqrt(Cin * k * k))
    w, wf = prep(_pack_conv_w(w4), dtype)
    w4f = wf.view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    b = rnd(Cout, seed=3).to(dev())
    ref = F.conv2d(xf.permute(0, 3, 1, 2), w4f, b.cpu(), stride=s, padding=p)
    out = ops.conv2d(x, w, k, k, s, p, bias=b, act=ops.ACT_RELU)
    check(out.permute(0, 3, 1, 2), F.relu(ref), dtype, name="conv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C,eps", [(7744, 64, 1e-6), (1936, 128, 1e-5), (485, 320, 1e-6), (121, 512, 1e-5),
                                     (33, 1024, 1e-5)])
def test_layernorm(dtype, M, C, eps):
    from emip_amd import ops
    x, xf = prep(rnd(M, C, seed=1) * 3 + 0.5, dtype)
    g = (1 + 0.1 * rnd(C, seed=2)).to(dev())
    b = (0.1 * rnd(C, seed=3)).to(dev())
    check(ops.layernorm(x, g, b, eps), F.layer_norm(xf, (C,), g.cpu(), b.cpu(), eps), dtype, name="ln")


def _attn_ref(q, k, v, scale, mask=None):
    s = q @ k.transpose(-1, -2) * scale
    if mask is not None:
        s = s + mask
    return s.softmax(-1) @ v, q @ k.transpose(-1, -2) * scale


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,heads", [(7744, 1), (1936, 2), (484, 5), (121, 8)])
def test_attention_sra(dtype, N, heads):
    """PVT SRA geometry: q [B,N,heads*64], kv [B,121,2*heads*64]"""
    from emip_amd import ops
    B, C, Lk = 2, heads * 64, 121
    q, qf = prep(rnd(B, N, C, seed=1), dtype)
    kv, kvf = prep(rnd(B, Lk, 2 * C, seed=2), dtype)
    out = torch.empty(B, N, C, dtype=dtype, device=dev())
    ops.attention(q, kv, kv[:, :, C:], out, batch=B, heads=heads, nwin=1, Lq=N, Lk=Lk, D=64, DV=64, q_bs=N * C,
                  k_bs=Lk * 2 * C, v_bs=Lk * 2 * C, o_bs=N * C, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64, k_hs=64,
                  v_hs=64, o_hs=64, scale=64 ** -0.5)
    qh = qf.view(B, N, heads, 64).permute(0, 2, 1, 3)
    kh = kvf[:, :, :C].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    vh = kvf[:, :, C:].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    ref, _ = _attn_ref(qh, kh, vh, 64 ** -0.5)
    check(out, ref.permute(0, 2, 1, 3).reshape(B, N, C), dtype, f32=2e-4, bf16=3e-2, name="sra")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shift", [False, True])
def test_attention_swin_window(dtype, shift):
    """GMFlow split-window attention with the roll/split/merge folded into index tables."""
    from emip_amd import ops
    from emip_amd.model.EMIP_short.motion.gmflow.tables import window_tables
    from oracle import emip_oracle as O
    B, H, W, C = 2, 44, 44, 128
    q, qf = prep(rnd(B, H * W, C, seed=1), dtype)
    k, kf = prep(rnd(B, H * W, C, seed=2), dtype)
    v, vf = prep(rnd(B, H * W, C, seed=3), dtype)
    rows_t, gid_t = window_tables(H, W, 2, shift, dev())
    out = torch.empty(B, H * W, C, dtype=dtype, device=dev())
    L = (H // 2) * (W // 2)
    ops.attention(q, k, v, out, batch=B, heads=1, nwin=4, Lq=L, Lk=L, D=128, DV=128, q_bs=H * W * C, k_bs=H * W * C,
                  v_bs=H * W * C, o_bs=H * W * C, ldq=C, ldk=C, ldv=C, ldo=C, q_rows=rows_t, k_rows=rows_t,
                  q_gid=gid_t if shift else None, k_gid=gid_t if shift else None, scale=C ** -0.5)
    ref = O.gm_window_attention(qf, kf, vf, H, W, shift, O.gm_shift_mask(H, W))
    check(out, ref, dtype, f32=2e-4, bf16=3e-2, name="swin")


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_scores_and_dv32(dtype):
    """global matching form: raw scores written out, V = 32-wide (2 used), f32 output."""
    from emip_amd import ops
    B, L, C = 2, 1936, 128
    q, qf = prep(rnd(B, L, C, seed=1), dtype)
    k, kf = prep(rnd(B, L, C, seed=2), dtype)
    vv = torch.zeros(L, 32)
    vv[:, 0] = torch.arange(L) % 44
    vv[:, 1] = torch.arange(L) // 44
    v, vf = prep(vv, dtype)
    out = torch.empty(B, L, 32, dtype=torch.float32, device=dev())
    scores = torch.empty(B, L, L, dtype=dtype, device=dev())
    ops.attention(q, k, v, out, batch=B, heads=1, nwin=1, Lq=L, Lk=L, D=128, DV=32, q_bs=L * C, k_bs=L * C, v_bs=0,
                  o_bs=L * 32, ldq=C, ldk=C, ldv=32, ldo=32, scale=C ** -0.5, scores=scores, s_bs=L * L, lds=L)
    ref, sref = _attn_ref(qf, kf, vf.unsqueeze(0), C ** -0.5)
    check(scores, sref, dtype, f32=1e-4, bf16=1e-2, name="scores")
    check(out[:, :, :2], ref[:, :, :2], dtype, f32=2e-4, bf16=3e-2, name="expectation")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Lq,Lk,heads,ksplit", [(300, 2500, 1, 4), (484, 9680, 2, 7), (130, 1100, 1, 5), (1936, 1936, 1, 3)])
def test_attention_kv_split(dtype, Lq, Lk, heads, ksplit):
    """emip_attention_splitkv == the one-pass kernel == softmax(QK^T)V: ragged key counts, splits that own no key tile
    ((130, 1100, ., 5) in bf16: 18 tiles -> 4 per split, the fifth split is empty), two heads, scattered output rows."""
    from emip_amd import ops
    B, D = 2, 128
    C = heads * D
    q, qf = prep(rnd(B, Lq, C, seed=1), dtype)
    k, kf = prep(rnd(B, Lk, C, seed=2) * 1.5, dtype)
    v, vf = prep(rnd(B, Lk, C, seed=3), dtype)
    perm = torch.randperm(Lq, generator=torch.Generator().manual_seed(4)).to(torch.int32).to(dev())
    outs = []
    for ks in (1, ksplit):
        out = torch.zeros(B, Lq, C, dtype=dtype, device=dev())
        ops.attention(q, k, v, out, batch=B, heads=heads, nwin=1, Lq=Lq, Lk=Lk, D=D, DV=D, q_bs=Lq * C, k_bs=Lk * C,
                      v_bs=Lk * C, o_bs=Lq * C, ldq=C, ldk=C, ldv=C, ldo=C, q_hs=D, k_hs=D, v_hs=D, o_hs=D,
                      q_rows=perm, scale=D ** -0.5, ksplit=ks)
        outs.append(out)
    qh = qf[:, perm.cpu().long()].view(B, Lq, heads, D).permute(0, 2, 1, 3)
    kh = kf.view(B, Lk, heads, D).permute(0, 2, 1, 3)
    vh = vf.view(B, Lk, heads, D).permute(0, 2, 1, 3)
    ref, _ = _attn_ref(qh, kh, vh, D ** -0.5)
    ref_rows = torch.zeros(B, Lq, C)
    ref_rows[:, perm.cpu().long()] = ref.permute(0, 2, 1, 3).reshape(B, Lq, C)
    check(outs[1], ref_rows, dtype, f32=2e-4, bf16=3e-2, name="kv-split")
    check(outs[1], outs[0].float().cpu(), dtype, f32=2e-4, bf16=2e-2, name="split-vs-one-pass")


@pytest.mark.parametrize("dtype", DTYPES)
def test_mdta_attn(dtype):
    from emip_amd import ops
    B, P, heads = 2, 1936, 2
    q, qf = prep(rnd(B, P, 128, seed=1), dtype)
    kv, kvf = prep(rnd(B, P, 256, seed=2), dtype)
    temp = torch.tensor([1.3, 0.8]).to(dev())
    attn = ops.mdta_attn(q, kv[:, :, :128], temp, B, heads, P)
    qh = F.normalize(qf.permute(0, 2, 1).reshape(B, heads, 64, P), dim=-1)
    kh = F.normalize(kvf[:, :, :128].permute(0, 2, 1).reshape(B, heads, 64, P), dim=-1)
    ref = ((qh @ kh.transpose(-1, -2)) * temp.cpu().view(1, heads, 1, 1)).softmax(-1)
    check(attn, ref, dtype, f32=1e-4, bf16=1e-2, name="mdta")


@pytest.mark.parametrize("dtype", DTYPES)
def test_dwconv(dtype):
    from emip_amd import ops
    B, H, W, C = 2, 22, 30, 256
    x, xf = prep(rnd(B, H, W, C, seed=1), dtype)
    w = rnd(C, 1, 3, 3, seed=2, scale=0.3)
    b = rnd(C, seed=3, scale=0.1)
    wt = w.view(C, 9).t().contiguous().to(dev())
    out = ops.dwconv3x3(x, wt, b.to(dev()), act=ops.ACT_GELU)
    ref = F.gelu(F.conv2d(xf.permute(0, 3, 1, 2), w, b, padding=1, groups=C))
    check(out.permute(0, 3, 1, 2), ref, dtype, name="dwconv+gelu")
    # gated (GDFN): 680 channels -> 340 (+4 zero pad)
    x2, x2f = prep(rnd(B, H, W, 680, seed=4), dtype)
    w2 = rnd(680, 1, 3, 3, seed=5, scale=0.3)
    out2 = ops.dwconv3x3_gated(x2, w2.view(680, 9).t().contiguous().to(dev()), 344)
    d = F.conv2d(x2f.permute(0, 3, 1, 2), w2, None, padding=1, groups=680)
    ref2 = F.gelu(d[:, :340]) * d[:, 340:]
    check(out2[..., :340].permute(0, 3, 1, 2), ref2, dtype, name="gated")
    assert out2[..., 340:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [64, 96, 128])
def test_instance_norm(dtype, C):
    from emip_amd import ops
    B, H, W = 3, 40, 44
    x, xf = prep(rnd(B, H, W, C, seed=1) * 2 + 1, dtype)
    r, rf = prep(rnd(B, H, W, C, seed=2), dtype)
    sums = ops.chan_stats(x, B)
    out = ops.chan_norm_apply(x, sums, B, 1e-5, relu_inner=True, relu_outer=True, res=r)
    ref = F.relu(rf.permute(0, 3, 1, 2) + F.relu(F.instance_norm(xf.permute(0, 3, 1, 2), eps=1e-5)))
    check(out.permute(0, 3, 1, 2), ref, dtype, name="instnorm")
    # batch-norm statistics form (one group) with affine
    g = (1 + 0.1 * rnd(C, seed=3)).to(dev())
    bb = (0.1 * rnd(C, seed=4)).to(dev())
    sums1 = ops.chan_stats(x, 1)
    out1 = ops.chan_norm_apply(x, sums1, 1, 1e-5, relu_inner=True, gamma=g, beta=bb)
    ref1 = F.relu(F.batch_norm(xf.permute(0, 3, 1, 2), None, None, g.cpu(), bb.cpu(), True, 0.0, 1e-5))
    check(out1.permute(0, 3, 1, 2), ref1, dtype, name="batchnorm-train")


@pytest.mark.parametrize("dtype", DTYPES)
def test_bilinear(dtype):
    from emip_amd import ops
    x, xf = prep(rnd(2, 11, 11, 32, seed=1), dtype)
    out = ops.bilinear(x, 22, 22, True)
    ref = F.interpolate(xf.permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    check(out.permute(0, 3, 1, 2), ref, dtype, f32=1e-5, name="up2 align")
    p, pf = prep(rnd(2, 44, 44, 4, seed=2), dtype)
    out8 = ops.bilinear_planar(p, 1, 1, 352, 352, False)
    ref8 = F.interpolate(pf[..., 1:2].permute(0, 3, 1, 2), scale_factor=8, mode="bilinear")
    check(out8, ref8, dtype, f32=1e-5, name="up8")
    fl = rnd(2, 44, 44, 2, seed=3).to(dev())
    outf = ops.bilinear_planar(fl, 0, 2, 352, 352, True, mul=8.0)
    reff = F.interpolate(fl.cpu().permute(0, 3, 1, 2), scale_factor=8, mode="bilinear", align_corners=True) * 8
    check(outf, reff, torch.float32, f32=1e-5, name="flow up8")


@pytest.mark.parametrize("dtype", DTYPES)
def test_eltwise_layout_copy(dtype):
    from emip_amd import ops
    a, af = prep(rnd(500, 32, seed=1), dtype)
    b, bf = prep(rnd(500, 32, seed=2), dtype)
    c, cf = prep(rnd(500, 32, seed=3), dtype)
    check(ops.eltwise(a, b, 0), af * bf, dtype, name="mul")
    check(ops.eltwise(a, b, 1, c3=c), af * bf * cf, dtype, name="mul3")
    check(ops.eltwise(a, b, 2), af + bf, dtype, name="add")
    pos, posf = prep(rnd(100, 32, seed=4), dtype)
    check(ops.eltwise(a, pos, 3, period=100), af + posf.repeat(5, 1), dtype, name="bcast")
    img = rnd(2, 3, 40, 36, seed=5).to(dev())
    cl = ops.planar_to_cl(img, dtype, 8)
    assert cl.shape == (2, 40, 36, 8)
    check(cl[..., :3].permute(0, 3, 1, 2), img.cpu().to(dtype).float(), dtype, f32=0, bf16=1e-9, name="to_cl")
    assert cl[..., 3:].abs().max().item() == 0
    back = ops.cl_to_planar(cl, 0, 3)
    check(back, img.cpu().to(dtype).float(), dtype, f32=0, bf16=1e-9, name="to_planar")
    dst = torch.ones(500, 136, dtype=dtype, device=dev())
    ops.copy_cols(a, 4, 6, dst, 128, 8)   # 6 columns + 2 columns of zero padding
    ops.copy_cols(b, 0, 32, dst, 0)
    assert torch.equal(dst[:, 128:134].float().cpu(), af[:, 4:10])
    assert dst[:, 134:136].abs().max().item() == 0 and torch.all(dst[:, 32:128] == 1)
    check(dst[:, :32], bf, dtype, f32=0, bf16=1e-9, name="copy")


@pytest.mark.parametrize("dtype", DTYPES)
def test_convex_upsample_and_flow(dtype):
    from emip_amd import ops
    N, H, W = 2, 44, 44
    lg, lgf = prep(rnd(N, H, W, 576, seed=1), dtype)
    flow = rnd(N, H, W, 2, seed=2, scale=3).to(dev())
    out = ops.convex_upsample(lg, flow)
    m = lgf.permute(0, 3, 1, 2).reshape(N, 1, 9, 8, 8, H, W).softmax(2)
    up = F.unfold(8 * flow.cpu().permute(0, 3, 1, 2), [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    ref = (m * up).sum(2).permute(0, 1, 4, 2, 5, 3).reshape(N, 2, 8 * H, 8 * W)
    check(out, ref, torch.float32, f32=1e-5, name="convex")
    o = rnd(N, H * W, 32, seed=3).to(dev())
    fl = ops.corresp_to_flow(o, N, H, W, True)
    from oracle import emip_oracle as O
    g = O.coords_grid(H, W).view(2, -1).t()
    check(fl.view(N, H * W, 2), o.cpu()[:, :, :2] - g, torch.float32, f32=1e-6, name="corresp")


def test_flow_warp_and_occlusion(golden):
    from emip_amd import ops
    g = golden("loss_micro.npz")
    x, flow = torch.from_numpy(g["x"]).to(dev()), torch.from_numpy(g["flow"]).to(dev())
    check(ops.flow_warp(x, flow), torch.from_numpy(g["warped"]), torch.float32, f32=1e-5, name="warp")
    occ = ops.occ_mask_backward(flow)
    assert (occ.cpu().numpy() != g["occ"]).mean() < 1e-3


def test_warp_indices_bit_exact(golden):
    """int64 corner indices == the tensor the REFERENCE handed to scatter_add_ (tests/golden/warp_indices_captured.npz,
    captured by oracle/make_golden_warp_capture.py inside loss/warp_utils.py:62-76), corner weights to 1 ulp-level, masks"""
    from emip_amd import ops
    g = golden("warp_indices_captured.npz")
    flows = {"a": torch.from_numpy(np.random.RandomState(11).normal(0, 6.0, (1, 2, 352, 352)).astype(np.float32)),
             "b": torch.from_numpy(np.random.RandomState(12).normal(0, 40.0, (2, 2, 352, 352)).astype(np.float32))}
    for name, fl in flows.items():
        idx, val = ops.occ_corners(fl.to(dev()))
        assert idx.dtype == torch.int64
        keep = slice(None) if name == "a" else slice(1, 2)
        assert np.array_equal(idx.cpu().numpy().astype(np.int32)[keep], g[name + "_indices"])
        assert np.abs(val.cpu().numpy()[:, ::61] - g[name + "_weights_sample"]).max() < 1e-6
        occ = ops.occ_mask_backward(fl.to(dev()))
        assert (occ.cpu().numpy().astype(np.uint8) != g[name + "_occ"]).mean() < 1e-4


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("M,N,K", [(3872, 320, 320), (968, 640, 320), (500, 64, 64), (7744, 512, 128)])
def test_gemm_with_folded_layernorm_and_row_statistics(dt, tol, M, N, K):
    """emip_gemm_ln: LN folded into the consumer (loader normalisation + gamma/beta in the weights) == LN then Linear;
    producer row statistics == sums of the stored rows; LayerNorm's own out_stats likewise"""
    from emip_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.7).to(dt)
    gamma, beta = 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = 0.1 * torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g).to(dt)
    ref = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-6) @ w.t() + b + res.float()
    dev = "cuda:0"
    xs = x.to(dev)
    stats = torch.stack([xs.float().sum(1), (xs.float() ** 2).sum(1)], 1).contiguous()
    wf = (w * gamma).to(dt).to(dev).contiguous()                 # gamma folded into the weights
    bf = (b + w @ beta).to(dev)                                  # beta folded into the bias
    out_stats = torch.zeros(M, 2, device=dev)
    y = ops.gemm(xs, wf, bias=bf, res=res.to(dev), ln_stats=stats, ln_eps=1e-6, out_stats=out_stats)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err
    want = torch.stack([y.float().sum(1), (y.float() ** 2).sum(1)], 1)
    assert ((out_stats - want).abs() / (want.abs() + 1.0)).max().item() < 1e-3
    # the output-side form of the same LayerNorm (emip_gemm_lne): rstd (x W^T) - rstd mean colsum(W)
    out_stats2 = torch.zeros(M, 2, device=dev)
    y2 = ops.gemm(xs, wf, bias=bf, res=res.to(dev), ln_stats=stats, ln_eps=1e-6, out_stats=out_stats2,
                  colsum=wf.float().sum(1).contiguous())
    err2 = (y2.float().cpu() - ref).abs().max().item()
    assert err2 < tol * max(1.0, ref.abs().max().item()), err2
    want2 = torch.stack([y2.float().sum(1), (y2.float() ** 2).sum(1)], 1)
    assert ((out_stats2 - want2).abs() / (want2.abs() + 1.0)).max().item() < 1e-3
    # a row mean far from zero (|mean| = 40 sigma): the cancellation in the output-side form stays inside the tolerance
    xo = (x.float() * 0.25 + 10.0).to(dt).to(dev)
    so = torch.stack([xo.float().sum(1), (xo.float() ** 2).sum(1)], 1).contiguous()
    refo = torch.nn.functional.layer_norm(xo.float().cpu(), (K,), gamma, beta, 1e-6) @ w.t() + b
    yo = ops.gemm(xo, wf, bias=bf, ln_stats=so, ln_eps=1e-6, colsum=wf.float().sum(1).contiguous())
    yl = ops.gemm(xo, wf, bias=bf, ln_stats=so, ln_eps=1e-6)
    lim = (4 if dt == torch.bfloat16 else 40) * tol * max(1.0, refo.abs().max().item())
    assert (yo.float().cpu() - refo).abs().max().item() < lim, ((yo.float().cpu() - refo).abs().max().item(), (yl.float().cpu() - refo).abs().max().item())
    # the scratch-clearing hook: a following launch zeroes a buffer while doing its own work
    scratch = torch.ones(37, device=dev)
    ops.gemm(xs, wf, bias=bf, zero=scratch)
    assert scratch.abs().max().item() == 0
    ls = torch.empty(M, 2, device=dev)
    z = ops.layernorm(xs, gamma.to(dev), beta.to(dev), 1e-6, out_stats=ls)
    wantl = torch.stack([z.float().sum(1), (z.float() ** 2).sum(1)], 1)
    assert ((ls - wantl).abs() / (wantl.abs() + 1.0)).max().item() < 1e-3


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,H,C,k", [(2, 88, 64, 8), (3, 44, 128, 4), (2, 22, 320, 2)])
def test_sr_conv_with_folded_layernorm(dt, tol, B, H, C, k):
    """emip_conv2d_ln on the spatial-reduction conv (kernel = stride): per-INPUT-pixel normalisation in the im2col loader"""
    from emip_amd import ops
    g = torch.Generator().manual_seed(B + H)
    x = (torch.randn(B, H, H, C, generator=g) + 0.3).to(dt)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w = torch.randn(C, C, k, k, generator=g) / (C * k * k) ** 0.5
    b = 0.1 * torch.randn(C, generator=g)
    xn = torch.nn.functional.layer_norm(x.float(), (C,), gamma, beta, 1e-6)
    ref = torch.nn.functional.conv2d(xn.permute(0, 3, 1, 2), w, b, stride=k).permute(0, 2, 3, 1)
    dev = "cuda:0"
    xs = x.to(dev)
    stats = torch.stack([xs.float().sum(-1), (xs.float() ** 2).sum(-1)], -1).view(-1, 2).contiguous()
    wf = (w * gamma.view(1, C, 1, 1)).permute(0, 2, 3, 1).reshape(C, -1).to(dt).to(dev).contiguous()
    bf = (b + (w * beta.view(1, C, 1, 1)).sum((1, 2, 3))).to(dev)
    out_stats = torch.zeros(B * (H // k) ** 2, 2, device=dev)
    y = ops.conv2d(xs, wf, k, k, k, 0, bias=bf, ln_stats=stats, ln_eps=1e-6, out_stats=out_stats)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err
    want = torch.stack([y.float().sum(-1), (y.float() ** 2).sum(-1)], -1).view(-1, 2)
    assert ((out_stats - want).abs() / (want.abs() + 1.0)).max().item() < 1e-3


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,H,C,k,ks", [(2, 88, 64, 8, 16), (3, 44, 128, 4, 8), (2, 22, 320, 2, 4), (1, 22, 320, 2, 7)])
def test_sr_conv_splitk_and_finalize(dt, tol, B, H, C, k, ks):
    """split-K spatial-reduction conv (f32 atomics, LayerNorm folded into the loader) + emip_rows_finalize == LN -> conv"""
    from emip_amd import ops
    g = torch.Generator().manual_seed(B * 7 + H)
    x = (torch.randn(B, H, H, C, generator=g) + 0.3).to(dt)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w = torch.randn(C, C, k, k, generator=g) / (C * k * k) ** 0.5
    b = 0.1 * torch.randn(C, generator=g)
    xn = torch.nn.functional.layer_norm(x.float(), (C,), gamma, beta, 1e-6)
    ref = torch.nn.functional.conv2d(xn.permute(0, 3, 1, 2), w, b, stride=k).permute(0, 2, 3, 1)
    dev = "cuda:0"
    xs = x.to(dev)
    stats = torch.stack([xs.float().sum(-1), (xs.float() ** 2).sum(-1)], -1).view(-1, 2).contiguous()
    wf = (w * gamma.view(1, C, 1, 1)).permute(0, 2, 3, 1).reshape(C, -1).to(dt).to(dev).contiguous()
    bf = (b + (w * beta.view(1, C, 1, 1)).sum((1, 2, 3))).to(dev)
    Ms = B * (H // k) ** 2
    acc = torch.zeros(Ms, C, device=dev)
    ops.conv2d_splitk(xs, wf, k, k, k, 0, bf, acc, ks, ln_stats=stats, ln_eps=1e-6)
    st = torch.empty(Ms, 2, device=dev)
    y = ops.rows_finalize(acc, dt, (B, H // k, H // k, C), out_stats=st)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < tol * max(1.0, ref.abs().max().item()), err
    want = torch.stack([y.float().sum(-1), (y.float() ** 2).sum(-1)], -1).view(-1, 2)
    assert ((st - want).abs() / (want.abs() + 1.0)).max().item() < 1e-3


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,H,C,k,ksplit", [(2, 88, 64, 8, 1), (2, 88, 64, 8, 16), (3, 44, 128, 4, 8), (2, 22, 320, 2, 4),
                                            (1, 22, 320, 2, 5)])
def test_conv_pair_with_fused_split_k(dt, tol, B, H, C, k, ksplit):
    """emip_conv2d_pair (q projection as a 1x1 conv + the spatial-reduction conv, both behind the folded LayerNorm) against
    the two separate launches; with ksplit > 1 the sr conv's K walk is split inside the launch and the last split to arrive
    runs the epilogue.  Launched three times on the same accumulator / tickets: every launch must leave them zero."""
    from emip_amd import ops
    g = torch.Generator().manual_seed(B * 100 + C + ksplit)
    dev = "cuda:0"
    x = (torch.randn(B, H, H, C, generator=g) * 1.3 + 0.4).to(dt).to(dev)
    wq = (torch.randn(C, C, generator=g) / C ** 0.5).to(dt).to(dev)
    wsr = (torch.randn(C, k * k * C, generator=g) / (k * k * C) ** 0.5).to(dt).to(dev)
    bq, bsr = (0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    xf = x.float().view(-1, C)
    stats = torch.stack([xf.sum(1), (xf ** 2).sum(1)], 1).contiguous()
    Ho = H // k
    Ms = B * Ho * Ho
    q_ref = ops.conv2d(x, wq, 1, 1, 1, 0, bias=bq, ln_stats=stats, ln_eps=1e-6)
    st_ref = torch.zeros(Ms, 2, device=dev)
    s_ref = ops.conv2d(x, wsr, k, k, k, 0, bias=bsr, ln_stats=stats, ln_eps=1e-6, out_stats=st_ref)
    tiles = ((Ms + 63) // 64) * ((C + 63) // 64)
    acc = torch.zeros(Ms * C, device=dev)
    ticket = torch.zeros(tiles, dtype=torch.int32, device=dev)
    for _ in range(3):
        q = torch.empty_like(q_ref)
        s_out = torch.empty_like(s_ref)
        st = torch.zeros(Ms, 2, device=dev)
        dense_q = ksplit in (1, 8)         # the q projection as a dense GEMM with the output-side LayerNorm
        ops.conv2d_pair(ops.conv_desc(x, wq, 1, 1, 0, bq, q, stats, 1e-6,
                                      colsum=wq.float().sum(1).contiguous() if dense_q else None),
                        ops.conv_desc(x, wsr, k, k, 0, bsr, s_out, stats, 1e-6, out_stats=st, acc=acc, ticket=ticket,
                                      ksplit=ksplit), dt)
        if dense_q:
            assert (q.float() - q_ref.float()).abs().max().item() < tol * max(1.0, q_ref.float().abs().max().item())
        else:
            assert torch.equal(q, q_ref)
        scale = max(1.0, s_ref.float().abs().max().item())
        assert (s_out.float() - s_ref.float()).abs().max().item() < tol * scale
        assert ((st - st_ref).abs() / (st_ref.abs() + 1.0)).max().item() < (2e-2 if dt == torch.bfloat16 else 1e-3)
        assert acc.abs().max().item() == 0 and ticket.abs().max().item() == 0


@pytest.mark.parametrize("B,H,W,K,N", [(8, 22, 22, 320, 1280), (3, 22, 22, 320, 1280), (5, 20, 24, 64, 128), (2, 17, 23, 96, 192),
                                       (4, 44, 44, 128, 512), (3, 88, 88, 64, 256), (2, 30, 37, 64, 128), (8, 24, 24, 64, 64)])
def test_mlp_fc1dw_fused_head(B, H, W, K, N):
    """emip_mlp_fc1dw (lib/pvt_v2.py:45-54 with norm2 folded, 165-169): fc1 + depthwise 3x3 + GELU in one launch == the two
    launches it replaces (bit for bit where both run the same K order, else to one bf16 step) == torch on the rounded operands.
    B = 8: the XCD-aware id decode; odd widths and a non-square image: the window logic at the borders."""
    from emip_amd import ops
    # maps of more than 512 tokens: row bands with a recomputed halo row on each side (44 x 44: 9-row bands, 88 x 88: 3-row bands,
    # 30 x 37: 11-row bands with a ragged last one, 24 x 24: 19 + 5 rows)
    assert ops.mlp_fc1dw_eligible(B, H, W, K, N) or ops.mlp_fc1dw_band_rows(B, H, W, K, N) > 0
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = (torch.randn(B, H, W, K, generator=g) * 1.3 + 0.2).to(torch.bfloat16).cuda()
    w1 = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b1 = (torch.randn(N, generator=g) * 0.1).cuda()
    wd = (torch.randn(9, N, generator=g) * 0.3).cuda()
    bd = (torch.randn(N, generator=g) * 0.1).cuda()
    xf = x.float().view(-1, K)
    stats = torch.stack((xf.sum(1), (xf * xf).sum(1)), 1).contiguous()
    cs = w1.float().sum(1).contiguous()
    fused = ops.mlp_fc1dw(x, w1, b1, cs, stats, 1e-6, wd, bd)
    two = ops.dwconv3x3(ops.gemm(x, w1, bias=b1, ln_stats=stats, ln_eps=1e-6, colsum=cs), wd, bd, act=ops.ACT_GELU)
    if B * H * W >= 2048:      # both paths then accumulate over K in the same MFMA order (the 8-wave LDS-DMA loops): no rounding
        assert torch.equal(fused, two)        # differs.  Fewer rows: the two-launch fc1 runs on the 4-wave 32 x 32 body.
    assert (fused.float() - two.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, two.float().abs().max().item())
    mu = xf.mean(1, keepdim=True)
    xn = (xf - mu) * torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-6)
    h = (xn @ w1.float().t() + b1).to(torch.bfloat16).float().view(B, H, W, N).permute(0, 3, 1, 2)
    ref = F.gelu(F.conv2d(h, wd.t().reshape(N, 1, 3, 3), bd, padding=1, groups=N)).permute(0, 2, 3, 1)
    err = (fused.float() - ref).abs().max().item()
    assert err < 0.03 * max(1.0, ref.abs().max().item()), err
    assert not ops.mlp_fc1dw_eligible(B, 11, 11, K, N)         # 121 tokens would pad to 512 MFMA rows: two launches stay


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,H,Cin,Cout,k,s,p,ks,ln", [(16, 11, 512, 32, 3, 1, 1, 9, False), (4, 22, 320, 32, 3, 1, 1, 3, False),
                                                      (4, 88, 64, 64, 8, 8, 0, 5, True), (5, 44, 128, 128, 4, 4, 0, 3, True),
                                                      (2, 22, 320, 320, 2, 2, 0, 4, True)])
def test_conv_split_k_reduced_inside_the_launch(dt, tol, B, H, Cin, Cout, k, s, p, ks, ln):
    """emip_conv2d_ksplit against emip_conv2d / emip_conv2d_ln on the same operands: the decoder-side reductions (3 x 3, pad 1,
    ReLU) and the spatial-reduction convs behind the folded LayerNorm with their row statistics.  Three launches: each gets a
    fresh accumulator, and (second check) one explicit accumulator is left zero by every launch."""
    from emip_amd import _lib, ops
    g = torch.Generator().manual_seed(B * 100 + Cin + ks)
    x = (torch.randn(B, H, H, Cin, generator=g) * 1.2 + 0.3).to(dt).cuda()
    w = (torch.randn(Cout, k * k * Cin, generator=g) / (k * k * Cin) ** 0.5).to(dt).cuda()
    b = (0.1 * torch.randn(Cout, generator=g)).cuda()
    xf = x.float().view(-1, Cin)
    stats = torch.stack([xf.sum(1), (xf ** 2).sum(1)], 1).contiguous() if ln else None
    act = ops.ACT_NONE if ln else ops.ACT_RELU
    Ho = (H + 2 * p - k) // s + 1
    M = B * Ho * Ho
    st_ref = torch.zeros(M, 2, device="cuda") if ln else None
    ref = ops.conv2d(x, w, k, k, s, p, bias=b, act=act, ln_stats=stats, ln_eps=1e-6, out_stats=st_ref)
    scale = max(1.0, ref.float().abs().max().item())
    for _ in range(3):
        st = torch.zeros(M, 2, device="cuda") if ln else None
        y = ops.conv2d_ksplit(x, w, k, k, s, p, ks, bias=b, act=act, ln_stats=stats, ln_eps=1e-6, out_stats=st)
        assert (y.float() - ref.float()).abs().max().item() < tol * scale
        if ln:
            assert ((st - st_ref).abs() / (st_ref.abs() + 1.0)).max().item() < (2e-2 if dt == torch.bfloat16 else 1e-3)
    ntick = ((M + 63) // 64) * ((Cout + 63) // 64)
    scratch = torch.zeros(M * Cout + ntick, device="cuda")
    y = torch.empty_like(ref)
    for _ in range(2):
        _lib.call("emip_conv2d_ksplit", x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), B, H, H, Cin, Cin, Cout, k, k, s, p,
                  Cout, act, stats.data_ptr() if ln else None, 1e-6, None, scratch.data_ptr(), scratch.data_ptr() + 4 * M * Cout,
                  ks, ops.dt_code(dt), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert (y.float() - ref.float()).abs().max().item() < tol * scale and scratch.abs().max().item() == 0
