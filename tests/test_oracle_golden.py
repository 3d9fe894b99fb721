"""The CPU oracle against fixtures produced by the reference itself
(oracle/make_golden.py).  This is the pin that lets the oracle stand in for the
reference on the GPU box."""
import numpy as np
import torch

from emip_amd.filler import synthetic_gt, synthetic_pair
from oracle import emip_oracle as O


def _stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.pow(2).sum().sqrt().item(), t.abs().max().item()])


def _close(a, b, atol, rtol=1e-4, name=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max()
    assert np.allclose(a, b, atol=atol, rtol=rtol), f"{name}: max abs err {err}"


def test_short_eval_b1(golden, short_sd):
    g = golden("short_eval_b1.npz")
    im1, im2 = synthetic_pair(1, seed=1234)
    cap = {}
    with torch.no_grad():
        mask, fw, bw = O.short_forward(im1, im2, short_sd, capture=cap)
    _close(cap["pvt1"][0][:, :, ::2, ::2], g["pvt1_s2"], 2e-4, name="pvt s2")
    _close(cap["pvt1"][1], g["pvt1_s3"], 2e-4, name="pvt s3")
    _close(cap["pvt1"][2], g["pvt1_s4"], 2e-4, name="pvt s4")
    _close(cap["gm1"][:, :, ::2, ::2], g["gm1"], 2e-4, name="gm cnn")
    _close(cap["inj_a"][:, :, ::2, ::2], g["inj_a"], 2e-4, name="injector a")
    _close(cap["f0"][:, :, ::2, ::2], g["f0"], 1e-3, name="gm transformer")
    corr = cap["corr"]
    _close(corr[:, :64].reshape(1, 64, -1)[:, :, :64], g["corr_block"], 2e-3, name="corr")
    _close(cap["flow_lr"], g["flow_lr"], 5e-2, name="flow_lr")  # ill-conditioned: softmax over 1936
    _close(cap["flow_prop"], g["flow_prop"], 5e-2, name="flow_prop")
    _close(cap["conv_corr"][:, :, ::2, ::2], g["conv_corr"], 5e-3, name="conv_corr")
    _close(cap["inj1"][:, :, ::2, ::2], g["inj1"], 5e-4, name="injector1")
    _close(cap["dr2"], g["dr2"], 2e-4, name="dr2")
    _close(cap["dr3"], g["dr3"], 2e-4, name="dr3")
    _close(cap["pc"], g["pc"], 1e-3, name="pc")
    _close(mask, g["mask"], 1e-3, name="mask")
    _close(fw[0][:, :, ::4, ::4], g["flow_fw"], 0.5, name="flow_fw")


def test_short_eval_b2_mask(golden, short_sd):
    g = golden("short_eval_b2.npz")
    im1, im2 = synthetic_pair(2, seed=1234)
    with torch.no_grad():
        mask, fw, bw = O.short_forward(im1, im2, short_sd)
    _close(mask, g["mask"], 1e-3, name="mask b2")
    assert mask.shape == (2, 1, 352, 352) and fw[0].shape == (2, 2, 352, 352) and len(fw) == 1


def test_short_train_losses(golden, short_sd):
    g = golden("short_train_b2.npz")
    im1, im2 = synthetic_pair(2, seed=77)
    gt = synthetic_gt(2, seed=99)
    with torch.no_grad():
        mask, fw, bw = O.short_forward(im1, im2, short_sd, training=True)
        lp = O.hybrid_e_loss(mask, gt)
        lf = O.unflow_loss([torch.cat([fw[i], bw[i]], 1) for i in range(len(fw))], torch.cat((im1, im2), 1))
    assert len(fw) == int(g["n_preds"]) == 2
    _close(mask, g["mask"], 2e-3, name="train mask")
    assert abs(lp.item() - float(g["loss_pred"])) < 1e-4
    assert abs(lf.item() - float(g["loss_flow"])) < 2e-3


def test_loss_micro(golden):
    g = golden("loss_micro.npz")
    x, y, flow = (torch.from_numpy(g[k]) for k in ("x", "y", "flow"))
    _close(O.flow_warp(x, flow), g["warped"], 1e-6, name="warp")
    _close(O.occu_mask_backward(flow), g["occ"], 0, name="occ")
    B, _, H, W = flow.shape
    idx, vals = O.corresponding_indices(O.mesh_grid(B, H, W).type_as(flow) + flow)
    cmap = torch.zeros(B, H * W).scatter_add_(1, idx, vals).view(B, 1, H, W)
    _close(cmap, g["cmap"], 1e-6, name="cmap")
    _close(O.ssim_dist(x, y), g["ssim"], 1e-6, name="ssim")
    assert abs(O.hybrid_e_loss(torch.from_numpy(g["pred"]), torch.from_numpy(g["gt"])).item() - float(g["hybrid"])) < 1e-6
    flows4 = [torch.cat([flow, -flow * 0.5], 1), torch.cat([flow * 0.9, -flow * 0.4], 1)]
    assert abs(O.unflow_loss(flows4, torch.cat((x, y), 1)).item() - float(g["unflow"])) < 1e-5


def _captured_flows():
    a = torch.from_numpy(np.random.RandomState(11).normal(0, 6.0, (1, 2, 352, 352)).astype(np.float32))
    b = torch.from_numpy(np.random.RandomState(12).normal(0, 40.0, (2, 2, 352, 352)).astype(np.float32))
    return a, b


def test_warp_indices_bit_exact(golden):
    """against the tensors the reference itself passed to scatter_add_ (oracle/make_golden_warp_capture.py)"""
    g = golden("warp_indices_captured.npz")
    for name, fl in zip("ab", _captured_flows()):
        idx, val = O.corresponding_indices(O.mesh_grid(fl.shape[0], 352, 352).type_as(fl) + fl)
        keep = slice(None) if name == "a" else slice(1, 2)
        assert np.array_equal(idx.numpy().astype(np.int32)[keep], g[name + "_indices"])
        assert np.array_equal(val.numpy()[:, ::61], g[name + "_weights_sample"])
        assert np.allclose(val.double().sum(1).numpy(), g[name + "_weights_sum"], rtol=1e-12)
        assert np.array_equal(O.occu_mask_backward(fl).numpy().astype(np.uint8), g[name + "_occ"])


def test_long_stream(golden, long_sd):
    g = golden("long_eval.npz")
    seq = [synthetic_pair(1, seed=500, shift=(t - 4, 4 - t))[1][0] for t in range(8)]
    mk = mv = None
    with torch.no_grad():
        for i in range(4):  # index 0 (short only), 1 (first memory), 2, 3
            if i == 0:
                m, _, _ = O.long_forward(seq[0], seq[1], 0, None, None, long_sd)
            else:
                m, mk, mv = O.long_forward(seq[i - 1], seq[i], i, mk, mv, long_sd)
                assert mk.shape[3] == int(g[f"T_{i}"])
                assert np.allclose(_stats(mk), g[f"k_{i}_stats"], rtol=1e-4, atol=1e-4)
            _close(m[:, :, ::2, ::2], g[f"mask_{i}"], 2e-3, name=f"long mask {i}")


def test_postprocess_restatement_known_answers():
    """oracle.postprocess_mask (test.py:28-31): identity resize of a logit ramp -> min-max normalised sigmoid, x255,
    truncated like PIL's F -> L conversion"""
    import numpy as np
    import torch
    from oracle import emip_oracle as O
    x = torch.linspace(-6, 6, 16 * 16).view(1, 1, 16, 16)
    out = O.postprocess_mask(x, (16, 16))
    p = torch.sigmoid(x).numpy().squeeze()
    want = np.floor((p - p.min()) / (p.max() - p.min() + 1e-8) * 255).astype(np.uint8)
    assert out.dtype == np.uint8 and out.shape == (16, 16)
    assert np.abs(out.astype(int) - want.astype(int)).max() <= 1 and out.min() == 0 and out.max() >= 254
    up = O.postprocess_mask(x, (40, 24))
    assert up.shape == (40, 24) and up[0, 0] == 0 and up[-1, -1] >= 254


def test_long_training_step_vs_reference(golden, long_sd):
    """oracle long_forward(training=True) + hybrid_e_loss + autograd against the reference's own train-mode step
    (oracle/make_golden_long_train.py): mask, loss, memory statistics and long-branch gradients"""
    import numpy as np
    import torch
    from emip_amd.filler import synthetic_gt, synthetic_pair
    from oracle import emip_oracle as O
    g = golden("long_train.npz")
    names = [str(n) for n in g["names"]]
    sd = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not k.startswith("short_term.")) else v)
          for k, v in long_sd.items()}
    seq = [synthetic_pair(1, seed=900, shift=(t - 2, 2 - t))[1][0] for t in range(3)]
    gt = synthetic_gt(1, seed=901)
    _, mk, mv = O.long_forward(seq[0], seq[1], 1, None, None, sd, training=True)
    mask, k2, v2 = O.long_forward(seq[1], seq[2], 2, mk.detach(), mv.detach(), sd, training=True)
    loss = O.hybrid_e_loss(mask, gt)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert (mask.detach()[:, :, ::2, ::2] - torch.from_numpy(g["mask"])).abs().max().item() < 1e-4
    assert int(g["T"]) == k2.shape[3] == 2

    def st(t):
        t = t.detach().double()
        return np.array([t.mean().item(), t.pow(2).sum().sqrt().item(), t.abs().max().item()])
    assert np.allclose(st(k2), g["k_stats"], rtol=1e-4, atol=1e-6)
    for i, n in enumerate(names):
        gr = sd[n].grad
        assert gr is not None, n
        assert np.allclose(st(gr), g["g%d_stats" % i], rtol=2e-3, atol=1e-7), n
        assert np.allclose(gr.reshape(-1)[:64].numpy(), g["g%d_head" % i], rtol=2e-3, atol=1e-6 * abs(g["g%d_stats" % i][2]) + 1e-9), n


def test_short_training_step_gradients_vs_reference(golden, short_sd):
    """oracle train-mode forward + both losses + torch autograd against the gradients the reference itself produced
    (oracle/make_golden_short_train.py, DropPath off)"""
    import numpy as np
    import torch
    from emip_amd.filler import synthetic_gt, synthetic_pair
    from oracle import emip_oracle as O
    g = golden("short_train_grads.npz")
    names = [str(n) for n in g["names"]]
    frozen = lambda k: "GMFlow" in k and 'dwconv' not in k and 'adaptor' not in k
    sd = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not frozen(k)) else v)
          for k, v in short_sd.items()}
    im1, im2 = synthetic_pair(1, seed=99)
    gt = synthetic_gt(1, seed=99)
    mask, fw, bw = O.short_forward(im1, im2, sd, training=True)
    lp = O.hybrid_e_loss(mask, gt)
    lf = O.unflow_loss([torch.cat((fw[i], bw[i]), 1) for i in range(len(fw))], torch.cat((im1, im2), 1))
    (lp + lf).backward()
    # flows are O(100 px) under the random filler: the batched restatement and the reference differ in the last f32 bits
    assert abs(lp.item() - float(g["loss_pred"])) < 1e-5 and abs(lf.item() - float(g["loss_flow"])) < 1e-4
    worst = 0.0
    for i, n in enumerate(names):
        gr = sd[n].grad
        ref_stats, ref_head = g["g%d_stats" % i], g["g%d_head" % i]
        err = np.abs(gr.reshape(-1)[:64].numpy() - ref_head).max() / max(ref_stats[2], 1e-12)
        l2 = abs(gr.double().pow(2).sum().sqrt().item() - ref_stats[1]) / max(ref_stats[1], 1e-12)
        worst = max(worst, err, l2)
        # the photometric loss is piecewise in the flow: the few parameters that see it only through the flows inherit
        # that sensitivity (same bound as the GPU test), everything else is tight
        tol = 0.3 if n.startswith("injector.") else (0.06 if ("conv_corr.0" in n or "block1." in n or "block2." in n
                                                              or "patch_embed1" in n) else 1e-2)
        assert err < tol and l2 < tol, (n, err, l2)
    print("oracle vs reference gradients, worst relative deviation", worst)


def test_short_eval_flow_well_conditioned(golden, short_sd):
    """flow outputs under the well-conditioned filler (oracle/make_golden_flow.py): the oracle vs the reference's values"""
    from emip_amd.filler import flow_conditioned, textured_pair
    g = golden("short_eval_flow.npz")
    assert float(g["thread_sensitivity_px"]) < 1e-4          # the reference repeats itself: the problem is well-conditioned
    im1, im2 = textured_pair()
    with torch.no_grad():
        m, fw, bw = O.short_forward(im1, im2, flow_conditioned(short_sd))
    assert (fw[0][:, :, ::4, ::4] - torch.from_numpy(g["fw"])).abs().max().item() < 1e-3
    assert (bw[0][:, :, ::4, ::4] - torch.from_numpy(g["bw"])).abs().max().item() < 1e-3
    assert (m[:, :, ::4, ::4] - torch.from_numpy(g["mask"])).abs().max().item() < 1e-3


def test_the_forward_reads_only_stage_two_of_the_second_frame(short_sd, monkeypatch):
    """The audit behind CoUpdater's PVT_DEEP_ONE_FRAME, on the oracle (the restatement of model/EMIP_short/model.py:86-102 the
    fixtures above pin): stages 3 and 4 of the SECOND frame are not connected to any output.  The backbone's outputs are
    replaced by autograd leaves; after a backward from mask + both flows the leaves of fea_2[1], fea_2[2] carry NO gradient
    (not a small one: none, they are outside the graph), fea_2[0], fea_1[0..2] do."""
    im1, im2 = synthetic_pair(1, seed=5)
    real = O.pvt_forward
    leaves = []

    def tapped(img, sd, p, drop=None):
        with torch.no_grad():
            outs = real(img, sd, p, drop)
        outs = [o.clone().requires_grad_(True) for o in outs]
        leaves.append(outs)
        return outs
    monkeypatch.setattr(O, "pvt_forward", tapped)
    mask, fw, bw = O.short_forward(im1, im2, short_sd)
    (mask.sum() + fw[-1].sum() + bw[-1].sum()).backward()
    f1, f2 = leaves                                  # 4 stage outputs each; the forward takes [1:]
    assert all(t.grad is not None and t.grad.abs().max() > 0 for t in f1[1:]), "frame 1: stages 2, 3, 4 are read"
    assert f2[1].grad is not None and f2[1].grad.abs().max() > 0, "frame 2: stage 2 is read (the camouflage feeder's prompt)"
    assert f2[2].grad is None and f2[3].grad is None, "frame 2: stages 3 and 4 feed nothing"
    assert f1[0].grad is None and f2[0].grad is None            # stage 1's output is only the next stage's input


def test_the_long_step_reads_only_stage_two_of_the_first_frame(long_sd, monkeypatch):
    """... and the mirror image for Model_long from frame 1 on (model_long.py:89-90,113-116): the deep stages of the FIRST frame
    feed only the short-term mask, which the step discards.  The short-term part runs under no_grad there, so the audit
    perturbs instead: with stages 3 and 4 of frame 0 replaced by noise the step's outputs (mask, keys, values) are bit-identical."""
    f0, f1 = synthetic_pair(1, seed=9)
    real = O.pvt_forward
    with torch.no_grad():
        ref = O.long_forward(f0[0], f1[0], 1, None, None, long_sd)
    calls = []

    def spoiled(img, sd, p, drop=None):
        outs = list(real(img, sd, p, drop))
        calls.append(len(calls))
        if len(calls) == 1:                       # the first call is frame 0 (short_forward: fea_1, then fea_2)
            outs[2] = torch.randn_like(outs[2]) * 7.0
            outs[3] = torch.randn_like(outs[3]) * 7.0
        return outs
    monkeypatch.setattr(O, "pvt_forward", spoiled)
    with torch.no_grad():
        got = O.long_forward(f0[0], f1[0], 1, None, None, long_sd)
    assert len(calls) == 2
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_postprocess_oracle_equals_the_bytes_the_reference_statements_wrote():
    """tests/golden/postprocess.npz = the PNG bytes test.py:29-31,35-36 wrote for six predictions (oracle/make_golden_postprocess.py
    executes those lines); the oracle's restatement, on one thread like the generator, must produce the same bytes"""
    import os
    import numpy as np
    from oracle.make_golden_postprocess import mask_logits
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "postprocess.npz"))
    kinds = ["field", "constant", "lowcontrast"]
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for i, (seed, h, w, kind) in enumerate(g["cases"].tolist()):
            out = O.postprocess_mask(torch.from_numpy(mask_logits(seed, kinds[kind])), (h, w))
            assert out.dtype == np.uint8 and np.array_equal(out, g["u8_%d" % i]), i
    finally:
        torch.set_num_threads(nt)
