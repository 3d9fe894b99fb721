"""Row statistics combined INSIDE a GEMM launch (emip_gemm_ln_ws: the last column tile of a row tile to finish adds the
tiles' partials in column order) against the statistics pass it replaces: same values to f32 rounding, the same bits on every
run, tickets left at zero, and ignored where a launch does not need it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(7744, 320, 1280), (7744, 320, 320), (15488, 320, 1280), (1000 + 936, 384, 128), (7744, 128, 320)])
def test_in_launch_statistics(M, N, K):
    from emip_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    out0 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st0 = torch.zeros(M, 2, device="cuda")
    ops.gemm(a, w, bias=bias, res=res, out=out0, out_stats=st0.view(-1))                    # statistics pass behind the launch
    nb = ops.gemm_stats_ws_bytes(M, N)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    runs = []
    for _ in range(3):
        out = torch.empty_like(out0)
        st = torch.full((M, 2), -7.0, device="cuda")
        ops.gemm(a, w, bias=bias, res=res, out=out, out_stats=st.view(-1), stats_ws=ws)
        runs.append((out, st))
    torch.cuda.synchronize()
    tickets = ws[:4 * ((M + 63) // 64)]
    assert int(tickets.max()) == 0                                                            # every launch leaves them at zero
    for out, st in runs:
        assert torch.equal(out, out0)
        assert torch.equal(st, runs[0][1])                                                    # fixed order: the same bits
        of = out.float()
        assert torch.allclose(st[:, 0], of.sum(1), rtol=1e-4, atol=2e-2) and torch.allclose(st[:, 1], (of * of).sum(1), rtol=1e-4, atol=2e-2)
        assert torch.allclose(st, st0, rtol=1e-5, atol=1e-3)


def test_launches_with_different_tiles_share_a_workspace():
    """proj (K = 320) and fc2 (K = 1280) of a block share the stage's workspace and may run on different tiles"""
    from emip_amd import ops
    M, N = 15488, 320
    ws = torch.zeros(ops.gemm_stats_ws_bytes(M, N), dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    for rep in range(2):
        for K in (320, 1280, 320):
            a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            st = torch.full((M, 2), -7.0, device="cuda")
            ops.gemm(a, w, out=out, out_stats=st.view(-1), stats_ws=ws)
            of = out.float()
            assert torch.allclose(st[:, 0], of.sum(1), rtol=1e-4, atol=2e-2), (rep, K)
            assert torch.allclose(st[:, 1], (of * of).sum(1), rtol=1e-4, atol=2e-2), (rep, K)
