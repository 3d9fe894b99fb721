"""emip_window_attention (GMFlow split-window attention, transformer.py:46-105) against a plain PyTorch f32 evaluation on
the same bf16-rounded operands and against the generic attention kernel it replaces on the inference path: plain and shifted
windows (additive -100 mask), self and cross attention (keys / values of the other frame), strided views of a fused
projection buffer."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(q, k, v, rows, gid, rot, scale):
    B = q.shape[0]
    out = torch.zeros(q.shape[0], q.shape[1], 128, device=q.device)
    kk = torch.roll(k, -rot, 0) if rot else k
    vv = torch.roll(v, -rot, 0) if rot else v
    for wi in range(rows.shape[0]):
        r = rows[wi].long()
        s = torch.einsum("bqc,bkc->bqk", q[:, r].float(), kk[:, r].float()) * scale
        if gid is not None:
            g = gid[wi]
            s = s + (g.view(-1, 1) != g.view(1, -1)).float() * -100.0
        out[:, r] = torch.softmax(s, -1) @ vv[:, r].float()
    return out


@pytest.mark.parametrize("B2,shift,rot", [(4, False, 0), (4, True, 0), (6, True, 3), (32, False, 16)])
def test_window_attention(B2, shift, rot):
    from emip_amd import ops
    from emip_amd.model.EMIP_short.motion.gmflow.tables import window_tables
    h = w = 44
    n, C = h * w, 128
    g = torch.Generator().manual_seed(B2 + rot)
    big = (torch.randn(B2, n, 5 * C, generator=g) * 1.5).cuda().to(torch.bfloat16)
    q, k, v = big[..., :C], big[..., 3 * C:4 * C], big[..., 4 * C:]
    rows, gid = window_tables(h, w, 2, shift, big.device)
    out = torch.full((B2, n, C), 3.0, dtype=torch.bfloat16, device="cuda")
    ops.window_attention(q, k, v, out, rows, gid if shift else None, n, C ** -0.5, rot)
    ref = _reference(q, k, v, rows, gid if shift else None, rot, C ** -0.5)
    err = (out.float() - ref).abs().max().item()
    old = torch.empty_like(out)
    L = rows.shape[1]
    ops.attention(q, k, v, old, batch=B2, heads=1, nwin=4, Lq=L, Lk=L, D=C, DV=C, q_bs=n * 5 * C, k_bs=n * 5 * C, v_bs=n * 5 * C,
                  o_bs=n * C, ldq=5 * C, ldk=5 * C, ldv=5 * C, ldo=C, q_rows=rows, k_rows=rows, q_gid=gid if shift else None,
                  k_gid=gid if shift else None, scale=C ** -0.5, kv_rot=rot)
    d_old = (out.float() - old.float()).abs().max().item()
    print(f"  B2={B2} shift={shift} rot={rot}: max |d| vs PyTorch {err:.4f}, vs the generic kernel {d_old:.4f} (values up to {ref.abs().max().item():.2f})")
    assert err < 2e-2 * max(1.0, ref.abs().max().item()) and d_old < 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B2,shift,rot,hw", [(4, False, 0, 44), (4, True, 0, 44), (6, True, 3, 44), (2, True, 1, 24)])
def test_window_attention_backward(B2, shift, rot, hw):
    """emip_window_attention_bwd (three launches: delta, dQ with the queries stationary, dK / dV with the keys stationary; P from
    the forward's log-sum-exp) against torch autograd through the f32 evaluation on the same bf16-rounded operands"""
    from emip_amd import ops
    from emip_amd.model.EMIP_short.motion.gmflow.tables import window_tables
    h = w = hw
    n, C = h * w, 128
    g = torch.Generator().manual_seed(7 + B2 + rot)
    big = (torch.randn(B2, n, 3 * C, generator=g) * 1.2).cuda().to(torch.bfloat16)
    q, k, v = big[..., :C], big[..., C:2 * C], big[..., 2 * C:]
    do = torch.randn(B2, n, C, generator=g).cuda().to(torch.bfloat16)
    rows, gid = window_tables(h, w, 2, shift, big.device)
    gm = gid if shift else None
    out = torch.empty((B2, n, C), dtype=torch.bfloat16, device="cuda")
    lse = torch.empty((B2, n), dtype=torch.float32, device="cuda")
    ops.window_attention(q, k, v, out, rows, gm, n, C ** -0.5, rot, lse=lse)
    # log-sum-exp in log2 units against the f32 evaluation
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref = _reference(qf, kf, vf, rows, gm, rot, C ** -0.5)
    kk = torch.roll(k, -rot, 0) if rot else k
    for wi in range(rows.shape[0]):
        r = rows[wi].long()
        s = torch.einsum("bqc,bkc->bqk", q[:, r].float(), kk[:, r].float()) * C ** -0.5
        if gm is not None:
            s = s + (gm[wi].view(-1, 1) != gm[wi].view(1, -1)).float() * -100.0
        l2 = torch.logsumexp(s, -1) * 1.4426950408889634
        assert (lse[:, r] - l2).abs().max().item() < 2e-2
    ref.backward(do.float())
    dq, dk, dv = ops.window_attention_bwd(q, k, v, out, do, lse, rows, gm, n, C ** -0.5, rot)
    torch.cuda.synchronize()
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        e = (got.float() - want).abs().max().item() / (want.abs().max().item() + 1e-9)
        print(f"  B2={B2} shift={shift} rot={rot} hw={hw} {name}: rel {e:.4f}")
        assert e < 2.5e-2, (name, e)


@pytest.mark.parametrize("B2,shift,rot,residual", [(4, False, 0, True), (4, True, 2, False), (32, True, 0, True)])
def test_window_attention_with_merge_norm_residual(B2, shift, rot, residual):
    """emip_window_attention_merge: attention + merge Linear + norm1 (+ residual) in one launch (transformer.py:330-338) against
    the two launches it replaces and the f32 evaluation"""
    import torch.nn.functional as F
    from emip_amd import ops
    from emip_amd.model.EMIP_short.motion.gmflow.tables import window_tables
    h = w = 44
    n, C = h * w, 128
    g = torch.Generator().manual_seed(100 + B2 + rot)
    big = (torch.randn(B2, n, 5 * C, generator=g) * 1.5).cuda().to(torch.bfloat16)
    q, k, v = big[..., :C], big[..., 3 * C:4 * C], big[..., 4 * C:]
    c0 = (torch.randn(B2, n, C, generator=g)).cuda().to(torch.bfloat16)
    wm = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    rows, gid = window_tables(h, w, 2, shift, big.device)
    gm = gid if shift else None
    att = torch.empty((B2, n, C), dtype=torch.bfloat16, device="cuda")
    ops.window_attention(q, k, v, att, rows, gm, n, C ** -0.5, rot)
    two = ops.gemm_ln_out(att, wm.to(torch.bfloat16).contiguous(), gamma, beta, 1e-5, res=c0 if residual else None)
    ref = F.layer_norm(_reference(q, k, v, rows, gm, rot, C ** -0.5).to(torch.bfloat16).float() @ wm.to(torch.bfloat16).float().t(),
                       (C,), gamma, beta, 1e-5)
    if residual:
        ref = ref + c0.float()
    out = c0.clone() if residual else torch.empty_like(c0)
    ops.window_attention_merge(q, k, v, out, rows, gm, n, C ** -0.5, ops.wattn_merge_pack(wm), gamma, beta, 1e-5,
                               res=out if residual else None, kv_rot=rot)
    torch.cuda.synchronize()
    scale = max(1.0, ref.abs().max().item())
    e_ref, e_two = (out.float() - ref).abs().max().item(), (out.float() - two.float()).abs().max().item()
    print(f"  B2={B2} shift={shift} rot={rot}: vs PyTorch {e_ref:.4f}, vs the two launches {e_two:.4f} (values up to {scale:.1f})")
    assert e_ref < 4e-2 * scale and e_two < 4e-2 * scale
    # the q projection in the launch too: q = tokens Wq^T computed in the prologue from the (residual) token rows
    wq = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    toks = c0.clone()
    qq = ops.gemm(toks, wq.to(torch.bfloat16).contiguous())
    want = torch.empty_like(c0)
    ops.window_attention_merge(qq, k, v, want, rows, gm, n, C ** -0.5, ops.wattn_merge_pack(wm), gamma, beta, 1e-5,
                               res=toks if residual else None, kv_rot=rot)
    got = toks.clone()
    ops.window_attention_merge(got, k, v, got, rows, gm, n, C ** -0.5, ops.wattn_merge_pack(wm), gamma, beta, 1e-5,
                               res=got if residual else None, kv_rot=rot, wq_pack=ops.wattn_q_pack(wq))
    torch.cuda.synchronize()
    e_q = (got.float() - want.float()).abs().max().item()
    print(f"     with the q projection in the prologue: vs the separate q GEMM {e_q:.4f}")
    assert e_q < 4e-2 * max(1.0, want.float().abs().max().item())
