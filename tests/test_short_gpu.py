"""End-to-end parity of the HIP EMIP-short forward: against the CPU oracle on the same seeded inputs, and
against the fixtures the reference itself produced (tests/golden).  f32 mode carries the 1e-3 mask bound of
BASELINE.json; bf16 (performance mode) carries per-stage relative bounds."""
import numpy as np
import pytest
import torch

from emip_amd.filler import synthetic_pair

pytestmark = pytest.mark.gpu


def _net(model_args, sd):
    from emip_amd.model.EMIP_short.model import CoUpdater
    net = CoUpdater(model_args)
    net.load_state_dict(sd)
    return net.to("cuda:0").eval()


def _rel(a, b):
    a, b = a.float().cpu(), torch.as_tensor(b).float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-6)).item()


def _planar(t):
    from emip_amd import ops
    return ops.cl_to_planar(t).cpu()


@pytest.fixture(scope="module")
def short_f32(model_args, short_sd):
    from emip_amd import nn_base
    nn_base.set_default_dtype(torch.float32)
    return _net(model_args, short_sd)


def test_f32_stages_and_mask_vs_reference_golden(short_f32, golden):
    from emip_amd import nn_base
    nn_base.set_default_dtype(torch.float32)
    g = golden("short_eval_b1.npz")
    im1, im2 = synthetic_pair(1, seed=1234)
    with torch.no_grad():
        mask, fw, bw = short_f32(im1.cuda(), im2.cuda())
    L = short_f32.last
    assert _rel(_planar(L["fea"][0][:1])[:, :, ::2, ::2], g["pvt1_s2"]) < 1e-3
    assert _rel(_planar(L["fea"][1][:1]), g["pvt1_s3"]) < 1e-3
    assert _rel(_planar(L["fea"][2][:1]), g["pvt1_s4"]) < 1e-3
    assert _rel(_planar(L["gm"][:1])[:, :, ::2, ::2], g["gm1"]) < 1e-3
    assert _rel(_planar(L["ab"][:1])[:, :, ::2, ::2], g["inj_a"]) < 1e-3
    corr = short_f32.last_corr().float().cpu()          # [B, src, tgt]; golden block is [tgt<64, src<64]
    assert _rel(corr[:, :64, :64].transpose(1, 2), g["corr_block"]) < 1e-3
    assert _rel(_planar(L["conv_corr"])[:, :, ::2, ::2], g["conv_corr"]) < 1e-3
    assert _rel(_planar(L["inj1"])[:, :, ::2, ::2], g["inj1"]) < 1e-3
    assert _rel(_planar(L["dr"][1]), g["dr2"]) < 1e-3 and _rel(_planar(L["dr"][2]), g["dr3"]) < 1e-3
    err = (mask.cpu() - torch.from_numpy(g["mask"])).abs().max().item()
    assert mask.shape == (1, 1, 352, 352) and mask.dtype == torch.float32
    assert err < 1e-3, f"mask logits max abs err vs reference {err}"      # BASELINE.json north_star bound
    assert len(fw) == 1 and fw[0].shape == (1, 2, 352, 352)
    # flow is ill-conditioned under random weights: softmax over 1936 near-ties
    assert (fw[0].cpu()[:, :, ::4, ::4] - torch.from_numpy(g["flow_fw"])).abs().max().item() < 0.5


def test_f32_batch2_vs_oracle_and_iou(short_f32, short_sd, golden):
    from emip_amd import nn_base
    from oracle import emip_oracle as O
    nn_base.set_default_dtype(torch.float32)
    im1, im2 = synthetic_pair(2, seed=1234)
    with torch.no_grad():
        mask, fw, bw = short_f32(im1.cuda(), im2.cuda())
        ref, rfw, rbw = O.short_forward(im1, im2, short_sd)
    m = mask.cpu()
    assert (m - ref).abs().max().item() < 1e-3
    assert (m - torch.from_numpy(golden("short_eval_b2.npz")["mask"])).abs().max().item() < 1e-3
    a, b = m >= 0, ref >= 0                                   # sigmoid(logit) >= 0.5, eval/metrics.py:488-492
    iou = (a & b).sum().item() / max((a | b).sum().item(), 1)
    assert iou > 0.999
    assert (bw[0].cpu() - rbw[0]).abs().max().item() < 0.5


def test_bf16_mode_stage_bounds(model_args, short_sd, golden):
    from emip_amd import nn_base
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        net = _net(model_args, short_sd)
        g = golden("short_eval_b1.npz")
        im1, im2 = synthetic_pair(1, seed=1234)
        with torch.no_grad():
            mask, fw, bw = net(im1.cuda(), im2.cuda())
        L = net.last
        # round 4: the bf16 forward is reproducible bit for bit (tests/test_determinism_gpu.py), so these are bounds on the bf16
        # error itself against the f32 reference -- 1.5 x the measured values (profiles/r04_bf16_stage_errors.json: 0.035, 0.017,
        # 0.016, 0.030), no run-to-run band on top
        e = (_rel(_planar(L["fea"][1][:1]), g["pvt1_s3"]), _rel(_planar(L["gm"][:1])[:, :, ::2, ::2], g["gm1"]),
             _rel(_planar(L["inj1"])[:, :, ::2, ::2], g["inj1"]), _rel(mask, torch.from_numpy(g["mask"])))
        print("  bf16 vs the f32 reference: pvt stage 3 %.4f, gmflow cnn %.4f, injector1 %.4f, mask %.4f" % e)
        assert e[0] < 0.052 and e[1] < 0.026 and e[2] < 0.024 and e[3] < 0.045, e
        ref = torch.from_numpy(g["mask"])
        a, b = mask.cpu() >= 0, ref >= 0
        assert (a & b).sum().item() / max((a | b).sum().item(), 1) > 0.97
    finally:
        nn_base.set_default_dtype(torch.float32)


def test_train_mode_forward_shapes(model_args, short_sd, golden):
    """train-mode semantics that do not need backward: BatchNorm batch statistics, two flow predictions."""
    from emip_amd import nn_base
    nn_base.set_default_dtype(torch.float32)
    net = _net(model_args, short_sd)
    net.train()
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    g = golden("short_train_b2.npz")
    im1, im2 = synthetic_pair(2, seed=77)
    with torch.no_grad():
        mask, fw, bw = net(im1.cuda(), im2.cuda())
    assert len(fw) == 2 and len(bw) == 2
    assert (mask.cpu() - torch.from_numpy(g["mask"])).abs().max().item() < 5e-3
    assert (fw[0].cpu()[:, :, ::4, ::4] - torch.from_numpy(g["flow0_fw"])).abs().max().item() < 0.5


def test_graph_replay_with_stream_splits_matches_eager(model_args, short_sd):
    """hipGraph replay (batch split over 2 concurrent streams) == eager forward"""
    from emip_amd import nn_base
    from emip_amd.graph import GraphedShort
    nn_base.set_default_dtype(torch.float32)
    net = _net(model_args, short_sd)
    im1, im2 = synthetic_pair(2, seed=4321)
    im1, im2 = im1.cuda(), im2.cuda()
    with torch.no_grad():
        ref_mask, ref_fw, ref_bw = net(im1, im2)
    runner = GraphedShort(net, 2, splits=2)
    for _ in range(2):                                   # replay twice: static buffers are reused
        mask, fw, bw = runner(im1, im2)
        torch.cuda.synchronize()
        assert (mask - ref_mask).abs().max().item() < 1e-3
        assert (fw[0] - ref_fw[0]).abs().max().item() < 0.5 and (bw[0] - ref_bw[0]).abs().max().item() < 0.5


def test_full_bench_batch_is_sample_independent_and_frame_symmetric(short_f32):
    """Size-independent properties at BASELINE.json's full batch (16 pairs, f32 mode): eval-mode outputs of a sample do
    not depend on its batch mates (16-pair launch == the same pairs run 3 / 1 at a time: odd and unit batches take
    other tile shapes), and swapping the two frames swaps the two flow directions."""
    from emip_amd import nn_base
    nn_base.set_default_dtype(torch.float32)
    im1, im2 = synthetic_pair(16, seed=2024)
    im1, im2 = im1.cuda(), im2.cuda()
    with torch.no_grad():
        mask, fw, bw = short_f32(im1, im2)
        m3, fw3, bw3 = short_f32(im1[5:8], im2[5:8])
        m1, fw1, bw1 = short_f32(im1[15:16], im2[15:16])
        ms, fws, bws = short_f32(im2[5:8], im1[5:8])
    assert (mask[5:8] - m3).abs().max().item() < 1e-3 and (mask[15:16] - m1).abs().max().item() < 1e-3
    assert (fw[0][5:8] - fw3[0]).abs().max().item() < 0.5 and (bw[0][15:16] - bw1[0]).abs().max().item() < 0.5
    # frame swap: the GMFlow stream is symmetric in its two inputs, so forward and backward flows trade places
    assert (fws[0] - bw3[0]).abs().max().item() < 0.5 and (bws[0] - fw3[0]).abs().max().item() < 0.5
    assert torch.isfinite(mask).all() and torch.isfinite(fw[0]).all()


def test_bf16_bench_batch_matches_f32(model_args, short_sd, short_f32):
    """the benchmarked configuration (16 pairs, bf16) against the f32 parity path on the same inputs: mask IoU"""
    from emip_amd import nn_base
    im1, im2 = synthetic_pair(16, seed=77)
    im1, im2 = im1.cuda(), im2.cuda()
    nn_base.set_default_dtype(torch.float32)
    with torch.no_grad():
        ref, _, _ = short_f32(im1, im2)
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        net = _net(model_args, short_sd)
        with torch.no_grad():
            out, _, _ = net(im1, im2)
    finally:
        nn_base.set_default_dtype(torch.float32)
    a, b = out > 0, ref > 0
    iou = ((a & b).sum().item() + 1e-9) / ((a | b).sum().item() + 1e-9)
    err = (out - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)
    print(f"bf16 vs f32 at 16 pairs: IoU {iou:.4f}, relative logit error {err:.3f}")
    assert iou > 0.97 and err < 0.08


def test_flow_outputs_under_the_well_conditioned_filler(model_args, short_sd, golden):
    """flow_fw / flow_bw (gmflow/gmflow.py:130-155) against the REFERENCE's values on a problem where the reference repeats
    itself to 1e-6 px (tests/golden/short_eval_flow.npz, oracle/make_golden_flow.py): f32 mode to 0.05 px on flows of up to
    300 px; bf16 mode: median error below 2 px (see the comment at the assertion); the mask of the same run to 1e-3 (f32)."""
    from emip_amd import nn_base
    from emip_amd.filler import flow_conditioned, textured_pair
    from emip_amd.model.EMIP_short.model import CoUpdater
    g = golden("short_eval_flow.npz")
    im1, im2 = textured_pair()
    sd = flow_conditioned(short_sd)
    res = {}
    try:
        for dt in (torch.float32, torch.bfloat16):
            nn_base.set_default_dtype(dt)
            net = CoUpdater(model_args)
            net.load_state_dict(sd)
            net = net.to("cuda:0").eval()
            with torch.no_grad():
                m, fw, bw = net(im1.cuda(), im2.cuda())
            dfw = (fw[0].float().cpu()[:, :, ::4, ::4] - torch.from_numpy(g["fw"])).norm(dim=1).flatten()
            dbw = (bw[0].float().cpu()[:, :, ::4, ::4] - torch.from_numpy(g["bw"])).norm(dim=1).flatten()
            efw, ebw = dfw.max().item(), dbw.max().item()
            em = (m.float().cpu()[:, :, ::4, ::4] - torch.from_numpy(g["mask"])).abs().max().item()
            res[dt] = (efw, ebw, em, dfw.median().item(), dbw.median().item())
            print(f"  {dt}: max |d flow_fw| {efw:.4f} px, |d flow_bw| {ebw:.4f} px (flows up to {float(g['fw_stats'][2]):.0f} px), "
                  f"median {res[dt][3]:.3f} / {res[dt][4]:.3f} px, mask {em:.2e}")
            del net
    finally:
        nn_base.set_default_dtype(torch.float32)
    assert res[torch.float32][0] < 0.05 and res[torch.float32][1] < 0.05 and res[torch.float32][2] < 1e-3
    # bf16: the fixture scales the matching features by 3, i.e. the correlation logits by 9: 8-bit mantissas on 128-channel
    # features then move logits of O(100) by O(1), and wherever the softmax over 1936 candidates has a runner-up within that
    # margin (white-noise frames under random CNN weights: ~40 % of the pixels) the expectation jumps by tens of pixels.
    # Measured on MI355X: median 0.75 / 0.19 px, 90th percentile 14 / 7 px, max 64 / 99 px.  The bf16 bound is therefore
    # a bound on the MEDIAN displacement error; the f32 mode above is the one that pins the flow arithmetic.
    assert res[torch.bfloat16][3] < 2.0 and res[torch.bfloat16][4] < 2.0


def test_bf16_inference_beyond_the_single_launch_batch(model_args, short_sd):
    """24 pairs = 48 images: the spatial-reduction conv of the 22 x 22 stage would need 273 tiles, more than the one-tile-per-
    workgroup launch form takes -- the batch goes in image chunks.  Same masks as two 12-pair forwards (bf16 repeatability band)."""
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.model import CoUpdater
    try:
        nn_base.set_default_dtype(torch.bfloat16)
        net = CoUpdater(model_args)
        net.load_state_dict(short_sd)
        net = net.to("cuda:0").eval()
        im1, im2 = synthetic_pair(24, seed=77)
        im1, im2 = im1.cuda(), im2.cuda()
        with torch.no_grad():
            whole = net(im1, im2)[0].float()
            halves = torch.cat([net(im1[:12], im2[:12])[0], net(im1[12:], im2[12:])[0]], 0).float()
            again = torch.cat([net(im1[:12], im2[:12])[0], net(im1[12:], im2[12:])[0]], 0).float()
        d, jit = (whole - halves).abs().max().item(), (again - halves).abs().max().item()
        print(f"  24 pairs at once vs 2 x 12: max |dlogit| {d:.4f} (two runs of the halves: {jit:.4f})")
        # round 4: no run-to-run band any more (fixed-order reductions).  The batch size still selects tiles and launch forms, hence
        # summation orders and bf16 rounding points: two bf16 evaluations may differ by about twice the bf16-vs-f32 error of the
        # mask logits (0.30-0.36 on a range of 7.9: bench.py parity.timed_outputs; measured here 0.04-0.39)
        assert torch.isfinite(whole).all() and jit == 0.0 and d <= 0.7
    finally:
        nn_base.set_default_dtype(torch.float32)
