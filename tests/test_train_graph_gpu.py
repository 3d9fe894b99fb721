"""The training step replayed as one hipGraph (emip_amd.train.GraphedTrainStep) against the eager step it captures
(emip_amd.train.forward_backward / train_step; reference: train.py:43-62), and the forked branch of the training step
(model.FORK_DEEP_TRAIN: PVT stages 3-4 forward and backward on a second stream) against the single-stream order."""
import copy

import pytest
import torch

from emip_amd.filler import synthetic_pair

pytestmark = pytest.mark.gpu


def _net(model_args, short_sd, drop):
    from emip_amd.model.EMIP_short.model import CoUpdater
    from emip_amd.train import freeze_like_reference
    net = CoUpdater(model_args)
    net.load_state_dict(short_sd)
    net = freeze_like_reference(net.to("cuda:0").train())
    if not drop:
        for m in net.modules():
            if hasattr(m, "drop_path_rate"):
                m.drop_path_rate = 0.0
    return net


def _grads(net):
    return {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None}


def _rel(a, b):
    """per gradient: largest |a - b| relative to the tensor's largest |b|"""
    assert a.keys() == b.keys(), sorted(a.keys() ^ b.keys())[:6]
    out = {}
    for n in a:
        d = (a[n] - b[n]).abs().max().item() / (b[n].abs().max().item() + 1e-20)
        assert d == d, n
        out[n] = d
    return out


def _check(got, ref, again):
    """`got` against `ref`, every gradient within 8 x what two eager runs of the same pass differ by (`again` vs `ref`: the
    weight- and bias-gradient reductions add in f32 with atomics; gradients that are zero in exact arithmetic -- a bias in
    front of a BatchNorm -- are ALL rounding noise, hence the per-tensor yardstick) or 2e-3 of the tensor's largest entry (one
    pair of runs underestimates the spread of a 60 000-term f32 sum whose order changes with what runs beside it: the decoder's
    gradients repeat to 3e-6 back to back and move by 3e-4 when another stream shares the chip; a missing ordering between
    streams shows as O(1) differences, not as this)"""
    noise, d = _rel(again, ref), _rel(got, ref)
    bad = {n: (d[n], noise[n]) for n in d if d[n] > max(8 * noise[n], 2e-3)}
    med = lambda v: sorted(v)[len(v) // 2]
    return (med(list(d.values())), med(list(noise.values()))), bad, max((d[n] / max(noise[n], 1e-7), n) for n in d)


@pytest.fixture()
def bf16():
    from emip_amd import nn_base
    nn_base.set_default_dtype(torch.bfloat16)
    yield
    nn_base.set_default_dtype(torch.float32)


def test_forked_backward_gives_the_single_stream_gradients(model_args, short_sd, bf16):
    """FORK_DEEP_TRAIN on / off: same kernels, same operands, other streams -- losses identical, gradients equal up to the
    order of the f32 atomic adds in the weight-gradient reductions"""
    import emip_amd.model.EMIP_short.model as M
    from emip_amd.filler import synthetic_gt
    from emip_amd.train import build_optimizer, forward_backward
    net = _net(model_args, short_sd, drop=False)
    opt = build_optimizer(net)
    im1, im2 = (t.cuda() for t in synthetic_pair(2, seed=7))
    gt = synthetic_gt(2, seed=7).cuda()
    out = {}
    prev = M.FORK_DEEP_TRAIN
    try:
        for flag in (False, True, False):
            M.FORK_DEEP_TRAIN = flag
            opt.zero_grad(set_to_none=True)
            losses = [float(x) for x in forward_backward(net, im1, im2, gt)]
            torch.cuda.synchronize()
            out.setdefault(flag, []).append((losses, _grads(net)))
    finally:
        M.FORK_DEEP_TRAIN = prev
    (l0, g0), (l2, g2) = out[False]
    (l1, g1), = out[True]
    assert l0 == l1 == l2, (l0, l1, l2)
    (md, mn), bad, worst = _check(g1, g0, g2)
    print("forked vs single-stream gradients: median difference %.2e, median of two single-stream runs %.2e; largest ratio "
          "%.1f (%s)" % ((md, mn) + worst))
    assert not bad, sorted(bad.items())[:6]
    assert md <= 4 * mn + 1e-6


def test_graphed_step_reproduces_the_eager_gradients(model_args, short_sd, bf16):
    """one replay of the captured forward + backward against the eager pass on the same weights and batch (DropPath off: the
    graph draws its own random numbers)"""
    from emip_amd.filler import synthetic_gt
    from emip_amd.train import GraphedTrainStep, build_optimizer, forward_backward
    net = _net(model_args, short_sd, drop=False)
    opt = build_optimizer(net)
    im1, im2 = (t.cuda() for t in synthetic_pair(2, seed=11))
    gt = synthetic_gt(2, seed=11).cuda()
    opt.zero_grad(set_to_none=True)
    le = [float(x) for x in forward_backward(net, im1, im2, gt)]
    ge = _grads(net)
    opt.zero_grad(set_to_none=True)
    le2 = [float(x) for x in forward_backward(net, im1, im2, gt)]
    ge2 = _grads(net)
    gs = GraphedTrainStep(net, opt, im1, im2, gt)
    for rep in range(2):                        # the second replay starts from whatever the first left in the pool
        lg = [float(x) for x in gs.replay()]
        torch.cuda.synchronize()
        (md, mn), bad, worst = _check(_grads(net), ge, ge2)
        print("replay %d: losses %s (eager %s); gradients: median difference %.2e, median of two eager runs %.2e, largest "
              "ratio %.1f (%s)" % ((rep, lg, le, md, mn) + worst))
        assert lg == le == le2
        assert not bad, sorted(bad.items())[:6]
        assert md <= 4 * mn + 1e-6


def test_graphed_training_follows_the_eager_trajectory(model_args, short_sd, bf16):
    """four optimizer steps: graph replays + the eager optimizer against eager train_step on copies of the model -- the weight
    packs the captured kernels read are the ones the refresh launch rewrites (a pack rebuilt lazily by the eager forward would
    go stale under replay and show up here).  Yardstick: a SECOND eager run (the backward's f32 atomics make two eager
    trajectories drift apart too, and AdamW's normalised update amplifies that)"""
    from emip_amd.filler import synthetic_gt
    from emip_amd.train import GraphedTrainStep, build_optimizer, train_step
    a, b, c = (_net(model_args, short_sd, drop=False) for _ in range(3))
    oa, ob, oc = (build_optimizer(m, lr=1e-4) for m in (a, b, c))
    im1, im2 = (t.cuda() for t in synthetic_pair(2, seed=3))
    gt = synthetic_gt(2, seed=3).cuda()
    gs = GraphedTrainStep(b, ob, im1, im2, gt)
    la, lb, lc = [], [], []
    for _ in range(4):
        la.append([float(x) for x in train_step(a, oa, None, im1, im2, gt)])
        lb.append([float(x) for x in gs.step(im1, im2, gt)])
        lc.append([float(x) for x in train_step(c, oc, None, im1, im2, gt)])
    print("eager  ", la)
    print("graphed", lb)
    print("eager 2", lc)
    assert la[0] == lb[0] == lc[0]
    # the second step sees one update made from gradients that agree to rounding; later steps drift (lr 1e-4 = 10 x the
    # reference's, AdamW normalises rounding-level gradient differences into full-size updates): bounded loosely
    for i, (x, y, z) in enumerate(zip(la, lb, lc)):
        tol = 5e-3 if i <= 1 else 5e-2
        assert all(abs(u - v) <= 4 * abs(u - w) + tol * max(1.0, abs(u)) for u, v, w in zip(x, y, z)), (i, x, y, z)
    assert la[-1][0] != la[0][0]                # the parameters moved
    pa, pb, pc = (dict(m.named_parameters()) for m in (a, b, c))
    moved = dg = de = 0.0
    for n, p in pa.items():
        if p.requires_grad:
            moved = max(moved, (p.detach() - short_sd[n].to(p.device)).abs().max().item())
            dg = max(dg, (p.detach() - pb[n].detach()).abs().max().item())
            de = max(de, (p.detach() - pc[n].detach()).abs().max().item())
    print("largest parameter change %.2e; eager vs graphed %.2e, eager vs eager %.2e" % (moved, dg, de))
    assert moved > 0 and dg <= 2 * de + 1e-7
    # the weight packs the captured kernels read ARE the current weights: the replayed forward on the trained model gives the
    # loss the eager forward gives, bit for bit (DropPath off, and the bf16 forward is reproducible); a pack that only the
    # eager path rebuilds would be stale inside the graph and show here
    assert lb[1] != lb[0]
    from emip_amd import nn_base
    assert nn_base.packs_not_kept_current(b) == []
    lg = [float(x) for x in gs.replay()]
    ob.zero_grad(set_to_none=True)
    from emip_amd.train import forward_backward
    le = [float(x) for x in forward_backward(b, im1, im2, gt)]
    print("after four steps: replayed forward", lg, "eager forward", le)
    assert lg == le


def test_graphed_step_with_stochastic_depth_draws_new_factors(model_args, short_sd, bf16):
    """DropPath on: every replay advances the generator (losses of two replays on the same weights differ), values finite"""
    from emip_amd.filler import synthetic_gt
    from emip_amd.train import GraphedTrainStep, build_optimizer
    net = _net(model_args, short_sd, drop=True)
    opt = build_optimizer(net)
    im1, im2 = (t.cuda() for t in synthetic_pair(2, seed=5))
    gt = synthetic_gt(2, seed=5).cuda()
    gs = GraphedTrainStep(net, opt, im1, im2, gt)
    ls = []
    for _ in range(4):
        ls.append(float(gs.replay()[0]))
    torch.cuda.synchronize()
    print("losses of four replays on the same weights:", ls)
    assert all(l == l and abs(l) < 1e4 for l in ls)
    assert len(set(ls)) > 1
    l = [float(x) for x in gs.step()]
    assert all(v == v for v in l)
