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
