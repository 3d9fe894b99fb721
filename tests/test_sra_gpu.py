"""emip_sra_attention (keys in registers, queries streamed) against a plain PyTorch f32 reference of
/root/reference/lib/pvt_v2.py:113-125 -- attn = softmax((q @ k^T) * scale), x = attn @ v -- on the four stage shapes of
pvt_v2_b5 and on ragged ones (query counts that are not multiples of 32, fewer than 121 keys)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, kv, heads):
    B, N, C = q.shape
    Lk = kv.shape[1]
    qh = q.float().view(B, N, heads, 64).permute(0, 2, 1, 3)
    k = kv.float()[..., :C].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    v = kv.float()[..., C:].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    a = torch.softmax(qh @ k.transpose(-1, -2) * 0.125, -1)
    return (a @ v).permute(0, 2, 1, 3).reshape(B, N, C)


@pytest.mark.parametrize("B,heads,N,Lk", [(2, 1, 7744, 121), (3, 2, 1936, 121), (4, 5, 484, 121), (2, 8, 121, 121),
                                           (1, 5, 100, 121), (2, 2, 33, 5), (1, 1, 1, 128), (32, 5, 484, 121)])
def test_sra_matches_reference(B, heads, N, Lk):
    from emip_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cuda").manual_seed(N + Lk)
    C = heads * 64
    q = (torch.randn(B, N, C, device=dev, generator=g) * 1.5).to(torch.bfloat16)
    kv = (torch.randn(B, Lk, 2 * C, device=dev, generator=g) * 1.5).to(torch.bfloat16)
    kv[0, 0, :64] *= 6.0                               # one dominant key: a peaked row next to flat ones
    out = torch.empty(B, N, C, device=dev, dtype=torch.bfloat16)
    ops.sra_attention(q, kv, out, B, heads, N, Lk, 0.125)
    got = out.float()
    ref = _ref(q, kv, heads)
    assert (got - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())
    # the generic flash kernel computes the same op: same inputs, same rounding class
    old = torch.empty(B, N, C, device=dev, dtype=torch.bfloat16)
    ops.attention(q, kv, kv[..., C:], old, batch=B, heads=heads, nwin=1, Lq=N, Lk=Lk, D=64, DV=64, q_bs=N * C, k_bs=Lk * 2 * C,
                  v_bs=Lk * 2 * C, o_bs=N * C, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, q_hs=64, k_hs=64, v_hs=64, o_hs=64, scale=0.125)
    assert (got - old.float()).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())


def test_sra_writes_nothing_beyond_its_rows():
    from emip_amd import ops
    dev = torch.device("cuda:0")
    B, heads, N, Lk = 2, 2, 45, 121
    C = heads * 64
    q = torch.randn(B, N, C, device=dev).to(torch.bfloat16)
    kv = torch.randn(B, Lk, 2 * C, device=dev).to(torch.bfloat16)
    buf = torch.full((B * N * C + 4096,), 9.0, device=dev, dtype=torch.bfloat16)
    ops.sra_attention(q, kv, buf[:B * N * C].view(B, N, C), B, heads, N, Lk, 0.125)
    assert (buf[B * N * C:] == 9.0).all()


@pytest.mark.parametrize("B,heads,Lq,Lk", [(3, 5, 484, 121), (2, 1, 7744, 121), (4, 2, 1936, 121), (2, 8, 121, 121), (2, 2, 300, 77),
                                           (52, 5, 484, 121), (32, 8, 121, 121)])
def test_sra_attention_backward_fused(B, heads, Lq, Lk):
    """emip_sra_attention_lse + emip_sra_attention_bwd (one launch each) against torch autograd on the rounded operands:
    the forward's output and log-sum-exp, dQ, dK, dV.  Shapes: the four PVT stages (stage 1 splits its 242 query blocks over
    workgroups: dK / dV meet by atomics), and one with a ragged query count and fewer keys."""
    from emip_amd import ops
    C = heads * 64
    g = torch.Generator().manual_seed(B * 100 + heads)
    q = torch.randn(B, Lq, C, generator=g).to(torch.bfloat16).cuda()
    kv = torch.randn(B, Lk, 2 * C, generator=g).to(torch.bfloat16).cuda()
    do = torch.randn(B, Lq, C, generator=g).to(torch.bfloat16).cuda()
    scale = 0.125
    out = torch.empty_like(q)
    L = ops.sra_attention_lse(q, kv, out, B, heads, Lq, Lk, scale)
    dq, dkv = ops.sra_attention_bwd(q, kv, out, do, L, B, heads, Lq, Lk, scale)
    qf = q.float().view(B, Lq, heads, 64).permute(0, 2, 1, 3).requires_grad_(True)
    kf = kv.float()[..., :C].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3).requires_grad_(True)
    vf = kv.float()[..., C:].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3).requires_grad_(True)
    s = (qf @ kf.transpose(-1, -2)) * scale
    ref = torch.softmax(s, -1) @ vf
    ref.backward(do.float().view(B, Lq, heads, 64).permute(0, 2, 1, 3))
    rel = lambda a, b: ((a.float() - b).abs().max() / (b.abs().max() + 1e-9)).item()
    assert rel(out.view(B, Lq, heads, 64).permute(0, 2, 1, 3), ref.detach()) < 1e-2
    lref = torch.logsumexp(s.detach(), -1) * 1.4426950408889634
    assert (L - lref).abs().max().item() < 2e-2
    assert rel(dq.view(B, Lq, heads, 64).permute(0, 2, 1, 3), qf.grad) < 2e-2
    dk = dkv[:, :Lk, :C].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    dv = dkv[:, :Lk, C:].reshape(B, Lk, heads, 64).permute(0, 2, 1, 3)
    assert rel(dk, kf.grad) < 2e-2 and rel(dv, vf.grad) < 2e-2
    assert dkv.shape == (B, Lk, 2 * C)
    # batch * heads >= 256: one workgroup per (image, head) stores final bf16 values (emip_sra_attention_bwd_bf16)
    assert dkv.dtype == (torch.bfloat16 if B * heads >= 256 else torch.float32)
