"""Fused clamp+AdamW (one launch) against clip_gradient + torch.optim.AdamW, the reference's step
(/root/reference/train.py:61-62,380; utils/utils.py:1-11)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_clamp_adamw_matches_torch():
    from emip_amd.optim import FusedClampAdamW
    torch.manual_seed(0)
    shapes = [(968, 1936, 3, 3), (320,), (1280, 320), (7,), (2049,), (64, 3, 7, 7), (1, 1)]
    a = [torch.nn.Parameter(torch.randn(s, device="cuda:0")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    opt_a = FusedClampAdamW(a, lr=1e-3, weight_decay=1e-2, clip=0.5)
    opt_b = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-2)
    for it in range(4):
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa) * (2.0 if it % 2 else 0.3)     # some elements beyond the +-0.5 clamp
            pa.grad = g.clone()
            pb.grad = g.clone().clamp_(-0.5, 0.5)
        opt_a.step()
        opt_b.step()
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6), (pa - pb).abs().max().item()
    sa, sb = opt_a.state[a[0]], opt_b.state[b[0]]
    assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-7)
    assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-4, atol=1e-9)
    assert sa["step"] == 4
