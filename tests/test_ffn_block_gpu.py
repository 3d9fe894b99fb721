"""emip_ffn_block (GMFlow transformer FFN in one launch: mlp[0] + GELU + mlp[2] + norm2 + residual, transformer.py:316-345)
against the two launches it replaces and a plain PyTorch f32 evaluation on the same bf16-rounded operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,strided", [(61952, False), (1000, False), (256, False), (4096, True), (77, False)])
def test_ffn_block(M, strided):
    from emip_amd import ops
    g = torch.Generator().manual_seed(M)
    bf = torch.bfloat16
    if strided:         # x1 / x2 as column slices of wider buffers, the residual = x1 written in place
        big = (torch.randn(M, 384, generator=g) * 1.2).to(bf).cuda()
        x1, x2 = big[:, :128], big[:, 256:]
    else:
        x1 = (torch.randn(M, 128, generator=g) * 1.2).to(bf).cuda()
        x2 = (torch.randn(M, 128, generator=g) * 1.2).to(bf).cuda()
    w0 = (torch.randn(1024, 256, generator=g) / 16).cuda()
    w2 = (torch.randn(128, 1024, generator=g) / 32).cuda()
    gamma = (1 + 0.1 * torch.randn(128, generator=g)).cuda()
    beta = (0.1 * torch.randn(128, generator=g)).cuda()
    eps = 1e-5
    w0b, w2b = w0.to(bf), w2.to(bf)
    # reference: f32 on the rounded operands, the hidden tensor rounded to bf16 like both device paths do
    xcat = torch.cat([x1.float(), x2.float()], 1)
    hid = F.gelu(xcat @ w0b.float().t()).to(bf).float()
    ref = x1.float() + F.layer_norm(hid @ w2b.float().t(), (128,), gamma, beta, eps)
    # the two launches
    h2 = ops.gemm(x1, w0b.contiguous(), a2=x2, act=ops.ACT_GELU)
    two = ops.gemm_ln_out(h2, w2b.contiguous(), gamma, beta, eps, res=x1)
    p0, p2 = ops.ffn_block_packs(w0, w2)
    out = ops.ffn_block(x1, x2, p0, p2, gamma, beta, eps, res=x1)
    torch.cuda.synchronize()
    scale = max(1.0, ref.abs().max().item())
    e_ref, e_two = (out.float() - ref).abs().max().item(), (out.float() - two.float()).abs().max().item()
    print(f"  M={M}: vs PyTorch {e_ref:.4f}, vs the two launches {e_two:.4f} (values up to {scale:.1f})")
    assert e_ref < 3e-2 * scale and e_two < 3e-2 * scale
    assert (two.float() - ref).abs().max().item() < 3e-2 * scale
    # in place (residual and first source = the output rows), as the transformer calls it
    buf = x1.clone() if not strided else None
    if buf is not None:
        ops.ffn_block(buf, x2, p0, p2, gamma, beta, eps, res=buf, out=buf)
        assert torch.equal(buf, out)
