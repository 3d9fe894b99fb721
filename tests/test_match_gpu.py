"""emip_match (GMFlow global matching in both directions + raw correlation volume, and flow propagation) against a plain
PyTorch f32 evaluation of matching.py:8-41 / transformer.py:503-533 on the same bf16-rounded features, and against the
generic attention kernel it replaces on the inference path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _feat(Z, n, seed, gain):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(Z, n, 128, generator=g) * gain).cuda().to(torch.bfloat16)


def _ref(q, k, v, scale, rot, W, sub):
    Z, n, _ = q.shape
    kk = torch.roll(k, -rot, 0) if rot else k            # batch z reads keys of batch (z + rot) mod Z
    s = torch.einsum("zqc,zkc->zqk", q.float(), kk.float()) * scale
    p = torch.softmax(s, -1)
    if v is None:
        idx = torch.arange(n, device=q.device)
        grid = torch.stack((idx % W, idx // W), -1).float()
        out = p @ grid
        if sub:
            out = out - grid
    else:
        vv = torch.roll(v, -rot, 0) if rot else v
        out = torch.einsum("zqk,zkc->zqc", p, vv.to(torch.bfloat16).float())
    return s, out


@pytest.mark.parametrize("Z,n,W,rot,gain", [(4, 1936, 44, 2, 1.0), (16, 1936, 44, 8, 0.6), (3, 136, 34, 1, 1.5), (2, 2048, 64, 0, 1.0)])
def test_matching_both_directions_with_scores(Z, n, W, rot, gain):
    from emip_amd import ops
    q = _feat(Z, n, 1, gain)
    scale = 128 ** -0.5
    Zs = max(1, Z // 2)
    scores = torch.full((Zs, n, n), 7.0, dtype=torch.bfloat16, device="cuda")
    out = ops.match(q, q, W, scale, scores=scores, kv_rot=rot)
    s_ref, o_ref = _ref(q, q, None, scale, rot, W, True)
    torch.cuda.synchronize()
    # the raw correlation: every element written once, bf16 rounding of the f32 value
    err_s = (scores.float() - s_ref[:Zs]).abs().max().item()
    assert err_s <= 4e-3 * s_ref.abs().max().item() + 1e-3, err_s
    # the flow: P is rounded to bf16 in front of the value MFMA (8 bits on weights that sum to one, coordinates < W)
    err = (out - o_ref).abs().max().item()
    print(f"  Z={Z} n={n}: max |d score| {err_s:.4f} (scores up to {s_ref.abs().max().item():.1f}), max |d flow| {err:.4f} px")
    assert err < 0.25, err
    assert torch.isfinite(out).all()


def test_matching_without_scores_and_peaked_rows():
    """identical frames, strongly peaked softmax: the flow is zero to a fraction of a pixel; no score buffer"""
    from emip_amd import ops
    q = _feat(2, 1936, 3, 4.0)
    out = ops.match(q, q, 44, 128 ** -0.5)
    _, o_ref = _ref(q, q, None, 128 ** -0.5, 0, 44, True)
    assert (out - o_ref).abs().max().item() < 0.25 and out.abs().max().item() < 0.5


@pytest.mark.parametrize("Z,n", [(16, 1936), (5, 1936)])
def test_flow_propagation_values(Z, n):
    from emip_amd import ops
    q, k = _feat(Z, n, 5, 1.0), _feat(Z, n, 6, 1.0)
    g = torch.Generator().manual_seed(9)
    flow = (torch.randn(Z, n, 2, generator=g) * 30).cuda()
    out = ops.match(q, k, 44, 128 ** -0.5, v=flow, sub_grid=False)
    _, o_ref = _ref(q, k, flow, 128 ** -0.5, 0, 44, False)
    err = (out - o_ref).abs().max().item()
    print(f"  propagation Z={Z}: max |d flow| {err:.4f} px on flows of +-{flow.abs().max().item():.0f} px")
    assert err < 0.02 * flow.abs().max().item()


def test_against_the_generic_attention_path():
    """the launch pair emip_match replaces in gmflow.py: same scores (bitwise: the same MFMA chain and rounding), flows equal
    to the rounding of P"""
    from emip_amd import ops
    from emip_amd.model.EMIP_short.motion.gmflow.gmflow import grid_values
    B, h, w, C = 3, 44, 44, 128
    n = h * w
    c0 = _feat(2 * B, n, 11, 1.0)
    corr_a = torch.empty((B, n, n), dtype=torch.bfloat16, device="cuda")
    flow_a = ops.match(c0, c0, w, C ** -0.5, scores=corr_a, kv_rot=B).view(2 * B, h, w, 2)
    grid = grid_values(h, w, torch.bfloat16, c0.device)
    corr_b = torch.empty_like(corr_a)
    o = torch.empty((2 * B, n, 32), dtype=torch.float32, device="cuda")
    common = dict(batch=B, heads=1, nwin=1, Lq=n, Lk=n, D=C, DV=32, q_bs=n * C, k_bs=n * C, v_bs=0, o_bs=n * 32, ldq=C, ldk=C,
                  ldv=32, ldo=32, scale=C ** -0.5)
    ops.attention(c0[:B], c0[B:], grid, o[:B], scores=corr_b, s_bs=n * n, lds=n, **common)
    ops.attention(c0[B:], c0[:B], grid, o[B:], **common)
    flow_b = ops.corresp_to_flow(o, 2 * B, h, w, True)
    assert torch.equal(corr_a, corr_b)
    assert (flow_a - flow_b).abs().max().item() < 0.2


@pytest.mark.parametrize("Z,n,W,rot,same,use_v,up", [(4, 1936, 44, 2, True, False, True), (2, 136, 34, 1, True, False, True),
                                                     (4, 1936, 44, 0, False, True, False), (3, 256, 16, 1, False, False, False),
                                                     (2, 2048, 64, 1, True, False, True)])
def test_matching_backward(Z, n, W, rot, same, use_v, up):
    """emip_match_bwd (statistics, dQ with the queries stationary, dK with the keys stationary; P from the forward's log-sum-exp;
    the upstream gradient of the returned score volume added to d score) against torch autograd through the f32 evaluation"""
    from emip_amd import ops
    scale = 128 ** -0.5
    q = _feat(Z, n, 11, 0.9)
    k = q if same else _feat(Z, n, 12, 0.9)
    g = torch.Generator().manual_seed(5)
    v = (torch.randn(Z, n, 2, generator=g) * 10).cuda() if use_v else None
    do = torch.randn(Z, n, 2, generator=g).cuda()
    Zs = max(1, Z // 2) if up else 0
    ds = (torch.randn(Zs, n, n, generator=g) * 0.05).cuda().to(torch.bfloat16) if up else None
    lse = torch.empty((Z, n), dtype=torch.float32, device="cuda")
    scores = torch.empty((Zs, n, n), dtype=torch.bfloat16, device="cuda") if up else None
    out = ops.match(q, k, W, scale, v=v, scores=scores, kv_rot=rot, sub_grid=not use_v, lse=lse)
    qf = q.float().clone().requires_grad_(True)
    kf = qf if same else k.float().clone().requires_grad_(True)
    kk = torch.roll(kf, -rot, 0) if rot else kf
    s = torch.einsum("zqc,zkc->zqk", qf, kk) * scale
    l2 = torch.logsumexp(s, -1) * 1.4426950408889634
    assert (lse - l2).abs().max().item() < 2e-2
    p = torch.softmax(s, -1)
    if use_v:
        vv = torch.roll(v, -rot, 0) if rot else v
        o = torch.einsum("zqk,zkc->zqc", p, vv)
    else:
        idx = torch.arange(n, device=q.device)
        o = p @ torch.stack((idx % W, idx // W), -1).float()
    loss = (o * do).sum()
    if up:
        loss = loss + (s[:Zs] * ds.float()).sum()
    loss.backward()
    dq, dk = ops.match_bwd(q, k, W, scale, out, do, lse, v=v, dscores=ds, kv_rot=rot, sub_grid=not use_v, accum=same)
    torch.cuda.synchronize()
    if same:
        e = (dq.float() - qf.grad).abs().max().item() / qf.grad.abs().max().item()
        print(f"  Z={Z} n={n} rot={rot} token gradient: rel {e:.4f}")
        assert e < 2.5e-2, e
    else:
        eq = (dq.float() - qf.grad).abs().max().item() / qf.grad.abs().max().item()
        ek = (dk.float() - kf.grad).abs().max().item() / kf.grad.abs().max().item()
        print(f"  Z={Z} n={n} rot={rot}: dq rel {eq:.4f}, dk rel {ek:.4f}")
        assert eq < 2.5e-2 and ek < 2.5e-2, (eq, ek)
