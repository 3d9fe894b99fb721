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
@pytest.mark.parametrize("M,C", [(7744, 64), (1936, 128), (485, 320), (121, 512)])
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
