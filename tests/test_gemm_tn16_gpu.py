"""emip_gemm_tn16_group (256 x 320 / 320 x 256 weight-gradient tiles, Linear and convolution problems in one persistent launch)
against f32 references, through the training step's deferral queue (emip_amd.ops.WgradQueue) -- the reference's
loss.backward() weight gradients (train.py:52-58 through lib/pvt_v2.py:45-54,101-129 and model/EMIP_short/model.py conv_corr)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lin_case(M, N, K, lda, ldb, colsum, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = (torch.randn(M, lda, device="cuda", generator=g)).to(torch.bfloat16)[:, :N]
    b = (torch.randn(M, ldb, device="cuda", generator=g)).to(torch.bfloat16)[:, :K]
    return a, b, colsum


LIN = [(30976, 1280, 320, 1280, 320, True), (30976, 320, 1280, 320, 1280, True), (30976, 320, 320, 320, 320, False),
       (7744, 640, 320, 640, 320, True), (4100, 968, 600, 968, 600, True), (2049, 320, 328, 336, 344, True),
       (5000, 512, 512, 512, 1024, False), (2500, 2048, 512, 2048, 512, True)]


def test_wide_tiles_take_the_320_shapes():
    from emip_amd import _lib
    lib = _lib.load()
    for N, K in ((320, 320), (1280, 320), (640, 320), (512, 512)):
        assert lib.emip_gemm_tn16_eligible(30976, N, K, N, K) == 1
    for N, K in ((320, 1280), (2048, 512)):
        assert lib.emip_gemm_tn16_eligible(30976, N, K, N, K) == 2
    for N, K in ((128, 128), (64, 256), (256, 64), (128, 512)):
        assert lib.emip_gemm_tn16_eligible(30976, N, K, N, K) == 0          # mostly padding: the 128 x 128 tiles keep them
    assert lib.emip_gemm_tn16_eligible(1024, 320, 320, 320, 320) == 0        # too short to split
    assert lib.emip_conv_wgrad16_eligible(64, 22, 22, 320, 320, 320, 320, 2, 2, 2, 0) == 1
    assert lib.emip_conv_wgrad16_eligible(64, 88, 88, 64, 64, 64, 64, 8, 8, 8, 0) == 0


def test_grouped_linear_problems_match_f32():
    from emip_amd import ops
    cases = [_lin_case(*c, seed=i) for i, c in enumerate(LIN)]
    ops.ARENA.begin("cuda:0")
    try:
        outs = []
        for a, b, cs in cases:
            assert ops.WGRADS.enabled
            outs.append(ops.gemm_tn(a, b, with_colsum=cs, defer=True))
        kinds = [it[9] for it in ops.WGRADS.items]
        assert kinds == [16] * len(cases), kinds
        ops.flush_wgrads()
    finally:
        ops.ARENA.end()
    for (a, b, cs), o in zip(cases, outs):
        c, db = o if cs else (o, None)
        ref = a.float().t() @ b.float()
        assert (c - ref).abs().max() <= 2e-5 * ref.abs().max(), (a.shape, b.shape)
        if cs:
            rdb = a.float().sum(0)
            assert (db - rdb).abs().max() <= 2e-5 * rdb.abs().max() + 1e-3


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,s,p", [(64, 22, 22, 320, 320, 2, 2, 0), (2, 44, 44, 1936, 968, 3, 1, 1),
                                                   (8, 32, 32, 64, 320, 3, 2, 1), (3, 30, 35, 320, 256, 3, 1, 1)])
def test_grouped_convolution_problem_matches_f32(B, H, W, Cin, Cout, k, s, p):
    from emip_amd import ops
    from emip_amd.autograd import conv_weight_grad
    torch.manual_seed(B + Cin)
    x = torch.randn(B, H, W, Cin, device="cuda").to(torch.bfloat16)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B, Ho, Wo, Cout, device="cuda").to(torch.bfloat16)
    w = torch.nn.Parameter(torch.zeros(Cout, Cin, k, k, device="cuda"))
    bias = torch.nn.Parameter(torch.zeros(Cout, device="cuda"))
    ops.ARENA.begin("cuda:0")
    try:
        g, db = conv_weight_grad(dy, x, w, k, s, p, bias=bias)
        assert len(ops.WGRADS.items) == 1 and ops.WGRADS.items[0][10] is not None      # it waits for the grouped launch
        assert not g.any() and not db.any()
        w.grad, bias.grad = g, db                    # what AccumulateGrad does with the first gradient of a parameter
    finally:
        ops.ARENA.end()                              # flush + fixup: the unpacked result is added to .grad
    assert not ops.WGRADS.post and not ops.WGRADS.items
    wf = torch.zeros(Cout, Cin, k, k, device="cuda", requires_grad=True)
    y = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wf, stride=s, padding=p)
    y.backward(dy.float().permute(0, 3, 1, 2))
    assert (w.grad - wf.grad).abs().max() <= 3e-5 * wf.grad.abs().max()
    rdb = dy.float().sum((0, 1, 2))
    assert (bias.grad - rdb).abs().max() <= 3e-5 * rdb.abs().max() + 1e-3           # the bias gradient of the same launch


def test_wide_and_narrow_problems_share_a_flush():
    """a flush with both kinds launches both groups; the 128 x 128 path keeps the shapes the wide tiles would mostly pad"""
    from emip_amd import ops
    cases = [_lin_case(30976, 320, 320, 320, 320, True, 1), _lin_case(30976, 128, 128, 128, 128, True, 2),
             _lin_case(7744, 640, 320, 640, 320, False, 3), _lin_case(123904, 64, 256, 64, 256, True, 4)]
    ops.ARENA.begin("cuda:0")
    try:
        outs = [ops.gemm_tn(a, b, with_colsum=cs, defer=True) for a, b, cs in cases]
        assert [it[9] for it in ops.WGRADS.items] == [16, 8, 16, 8]
    finally:
        ops.ARENA.end()
    for (a, b, cs), o in zip(cases, outs):
        c = o[0] if cs else o
        ref = a.float().t() @ b.float()
        assert (c - ref).abs().max() <= 2e-5 * ref.abs().max()
