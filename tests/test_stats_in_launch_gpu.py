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
    tickets = ws[:16384]
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
    N = 320
    ws = torch.zeros(ops.gemm_stats_ws_bytes(15488, N), dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    for rep in range(2):
        for M, K in ((15488, 320), (15488, 1280), (1936, 1280), (7744, 320)):       # other row counts too: one ticket block size
            a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            st = torch.full((M, 2), -7.0, device="cuda")
            ops.gemm(a, w, out=out, out_stats=st.view(-1), stats_ws=ws)
            of = out.float()
            assert torch.allclose(st[:, 0], of.sum(1), rtol=1e-4, atol=2e-2), (rep, K)
            assert torch.allclose(st[:, 1], (of * of).sum(1), rtol=1e-4, atol=2e-2), (rep, K)


def test_spatial_reduction_conv_with_in_launch_statistics():
    """emip_conv8_ws on the 22 x 22 stage's sr conv (per-tap LayerNorm, 64 x 128 tiles: three column tiles per row) against the
    statistics pass, sharing its workspace with GEMMs of other row counts"""
    from emip_amd import ops
    B = 16
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, 22, 22, 320, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(320, 1280, device="cuda", generator=g) / 36).to(torch.bfloat16)
    bias = torch.randn(320, device="cuda", generator=g)
    tsum = w.float().view(320, 4, 320).sum(2).t().contiguous()
    xf = x.float().view(-1, 320)
    stats = torch.stack((xf.sum(1), (xf * xf).sum(1)), 1).contiguous().view(-1)
    ws = torch.zeros(ops.gemm_stats_ws_bytes(B * 484, 320), dtype=torch.uint8, device="cuda")
    st0 = torch.zeros(B * 121, 2, device="cuda")
    y0 = ops.conv8(x, w, 2, 2, 2, 0, bias=bias, ln_stats=stats, tapsum=tsum, ln_eps=1e-6, out_stats=st0.view(-1), cfg=9)
    for _ in range(2):
        a = torch.randn(B * 484, 320, device="cuda", generator=g).to(torch.bfloat16)       # a token GEMM in between, same workspace
        ops.gemm(a, w[:, :320].contiguous(), out_stats=torch.empty(B * 484 * 2, device="cuda"), stats_ws=ws)
        st = torch.full((B * 121, 2), -3.0, device="cuda")
        y = ops.conv8(x, w, 2, 2, 2, 0, bias=bias, ln_stats=stats, tapsum=tsum, ln_eps=1e-6, out_stats=st.view(-1), cfg=9, stats_ws=ws)
        torch.cuda.synchronize()
        assert torch.equal(y, y0) and torch.isfinite(st).all()
        yf = y.float().view(-1, 320)
        assert torch.allclose(st[:, 0], yf.sum(1), rtol=1e-4, atol=2e-2) and torch.allclose(st[:, 1], (yf * yf).sum(1), rtol=1e-4, atol=2e-2)
        assert torch.allclose(st, st0, rtol=1e-5, atol=1e-3)
    assert int(ws[:16384].max()) == 0
