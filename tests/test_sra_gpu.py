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
