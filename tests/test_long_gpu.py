"""EMIP-long on the GPU against the fixture produced by the reference (a 7-step stream, memory window
saturating at T=5), plus batched streams == independent streams."""
import numpy as np
import pytest
import torch

from emip_amd.filler import synthetic_pair

pytestmark = pytest.mark.gpu


def _stats(t):
    t = t.detach().double()
    return np.array([t.mean().item(), t.pow(2).sum().sqrt().item(), t.abs().max().item()])


@pytest.fixture(scope="module")
def long_net(model_args, long_sd):
    from emip_amd import nn_base
    from emip_amd.model.EMIP_long.model_long import Model_long
    nn_base.set_default_dtype(torch.float32)
    net = Model_long(model_args)
    net.load_state_dict(long_sd)
    return net.to("cuda:0").eval()


def _seq(seed):
    return [synthetic_pair(1, seed=seed, shift=(t - 4, 4 - t))[1][0].cuda() for t in range(8)]


def test_long_stream_vs_reference_golden(long_net, golden):
    g = golden("long_eval.npz")
    seq = _seq(500)
    mk = mv = None
    with torch.no_grad():
        for i in range(7):
            if i == 0:
                m, mk, mv = long_net(seq[0], seq[1], 0, None, None)
                assert mk is None and mv is None
            else:
                m, mk, mv = long_net(seq[i - 1], seq[i], i, mk, mv)
                mk, mv = mk.detach(), mv.detach()
                assert mk.shape == (1, 1, 128, int(g[f"T_{i}"]), 44, 44)
                assert np.allclose(_stats(mk.cpu()), g[f"k_{i}_stats"], rtol=2e-3, atol=1e-3)
                assert np.allclose(_stats(mv.cpu()), g[f"v_{i}_stats"], rtol=2e-3, atol=1e-3)
            assert m.shape == (1, 1, 352, 352)
            err = (m.cpu()[:, :, ::2, ::2] - torch.from_numpy(g[f"mask_{i}"])).abs().max().item()
            assert err < 2e-3, f"frame {i}: mask max abs err {err}"


def test_batched_streams_equal_single_streams(long_net):
    a, b = _seq(500), _seq(77)
    with torch.no_grad():
        singles = []
        for seq in (a, b):
            mk = mv = None
            outs = []
            for i in range(1, 4):
                m, mk, mv = long_net(seq[i - 1], seq[i], i, mk, mv)
                outs.append(m)
            singles.append(outs)
        mk = mv = None
        for i in range(1, 4):
            f0 = torch.stack([a[i - 1], b[i - 1]])
            f1 = torch.stack([a[i], b[i]])
            m, mk, mv = long_net.forward_streams(f0, f1, i, mk, mv)
            assert m.shape == (2, 1, 352, 352) and mk.shape[0] == 2
            for s in range(2):
                assert (m[s] - singles[s][i - 1][0]).abs().max().item() < 2e-4


def test_memory_fed_back_as_fresh_tensor(long_net):
    """the caller may hand back a copy (not the cached object): the layout conversion path must agree"""
    seq = _seq(500)
    with torch.no_grad():
        _, mk, mv = long_net(seq[0], seq[1], 1, None, None)
        m1, _, _ = long_net(seq[1], seq[2], 2, mk, mv)
        _, mk2, mv2 = long_net(seq[0], seq[1], 1, None, None)
        m2, _, _ = long_net(seq[1], seq[2], 2, mk2.clone(), mv2.clone())
    assert (m1 - m2).abs().max().item() < 1e-3   # run-to-run jitter of the f32-atomic Gram accumulation


def test_graphed_steady_state_matches_eager(long_net):
    """hipGraph replay of the steady-state step (window full, memory sliding inside the graph) == eager forward_streams"""
    from emip_amd.graph import GraphedLong
    a, b = _seq(500), _seq(77)
    f = lambda i: (torch.stack([a[i - 1], b[i - 1]]), torch.stack([a[i], b[i]]))
    with torch.no_grad():
        mk = mv = None
        for i in range(1, 6):                                   # fill the 5-frame window eagerly
            m, mk, mv = long_net.forward_streams(*f(i), i, mk, mv)
        assert mk.shape[3] == 5
        runner = GraphedLong(long_net, 2, splits=2)
        runner.seed_memory(mk, mv)
        for i in (6, 7):
            ref, mk, mv = long_net.forward_streams(*f(i), i, mk, mv)
            out = runner(*f(i))
            torch.cuda.synchronize()
            assert (out - ref).abs().max().item() < 1e-3
            gk, gv = runner.memory()
            assert (gk - mk).abs().max().item() < 1e-3 and (gv - mv).abs().max().item() < 1e-3


def test_long_training_step_vs_reference_golden(model_args, long_sd, golden):
    """one EMIP-long training step (train_long.py:37-58: train mode, short-term part frozen and under no_grad,
    hybrid_e_loss, backward) against the fixture the reference itself produced: mask, loss, memory and the gradients of
    the long-branch parameters"""
    from emip_amd import nn_base
    from emip_amd.filler import synthetic_gt
    from emip_amd.loss.loss_pred import hybrid_e_loss
    from emip_amd.model.EMIP_long.model_long import Model_long
    nn_base.set_default_dtype(torch.float32)
    g = golden("long_train.npz")
    net = Model_long(model_args)
    net.load_state_dict(long_sd)
    net = net.to("cuda:0")
    for name, para in net.named_parameters():              # train_long.py:404-406
        if "short_term" in name:
            para.requires_grad_(False)
    net.train()
    for m in net.modules():                                # the fixture was produced with DropPath off
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    seq = [synthetic_pair(1, seed=900, shift=(t - 2, 2 - t))[1][0].cuda() for t in range(3)]
    gt = synthetic_gt(1, seed=901).cuda()
    with torch.enable_grad():
        _, mk, mv = net(seq[0], seq[1], 1, None, None)
        mk, mv = mk.detach(), mv.detach()
        mask, k2, v2 = net(seq[1], seq[2], 2, mk, mv)
        loss = hybrid_e_loss(mask, gt)
        loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    assert (mask.detach().cpu()[:, :, ::2, ::2] - torch.from_numpy(g["mask"])).abs().max().item() < 2e-3
    assert k2.shape == (1, 1, 128, 2, 44, 44)
    assert np.allclose(_stats(k2.cpu()), g["k_stats"], rtol=2e-3, atol=1e-3)
    p = dict(net.named_parameters())
    worst = 0.0
    for i, n in enumerate(str(x) for x in g["names"]):
        gr = p[n].grad
        assert gr is not None, n
        ref_head, ref_stats = g["g%d_head" % i], g["g%d_stats" % i]
        err = np.abs(gr.detach().reshape(-1)[:64].cpu().numpy() - ref_head).max() / max(ref_stats[2], 1e-12)
        l2 = abs(_stats(gr.cpu())[1] - ref_stats[1]) / max(ref_stats[1], 1e-12)
        worst = max(worst, err, l2)
        assert err < 2e-2 and l2 < 2e-2, (n, err, l2)
    print("EMIP-long training step: worst relative gradient deviation vs the reference", worst)
    assert not [n for n, q in p.items() if "short_term" in n and q.grad is not None]
    have = {n for n, q in p.items() if q.requires_grad and q.grad is None}
    assert have == {str(x) for x in g["no_grad"]}, sorted(have ^ {str(x) for x in g["no_grad"]})[:6]


def test_train_long_video_runs_with_full_window(model_args, long_sd):
    """emip_amd.train.train_long_video over a 7-frame clip in bf16: the window saturates at 5 frames (9680-key memory
    read and its backward), long-branch parameters move, short-term ones do not"""
    from emip_amd import nn_base
    from emip_amd.filler import synthetic_gt
    from emip_amd.model.EMIP_long.model_long import Model_long
    from emip_amd.train import build_optimizer, freeze_short_term, train_long_video
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        net = Model_long(model_args)
        net.load_state_dict(long_sd)
        net = freeze_short_term(net.to("cuda:0")).train()
        opt = build_optimizer(net, lr=1e-5, weight_decay=1e-7, clip=0.5)
        frames = torch.stack(_seq(321)[:7])
        masks = torch.cat([synthetic_gt(1, seed=40 + t) for t in range(7)]).cuda()
        w_long = net.LTM.KV_Q_r4.Key.weight.detach().clone()
        w_short = net.short_term.decoder.conv5.weight.detach().clone()
        loss = train_long_video(net, opt, None, frames, masks)
        assert torch.isfinite(loss).item() and 0 < loss.item() < 100
        assert (net.LTM.KV_Q_r4.Key.weight.detach() - w_long).abs().max().item() > 0
        assert torch.equal(net.short_term.decoder.conv5.weight.detach(), w_short)
    finally:
        nn_base.set_default_dtype(torch.float32)
