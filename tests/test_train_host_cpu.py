"""Host-side bookkeeping of the training step that needs no device: the kept intermediates are detached (a kept grad_fn pins the
step's autograd graph and the streams of its AccumulateGrad nodes), the registry of weight packs a captured step depends on,
and the per-stream weight-gradient queues."""
import torch
import torch.nn as nn


def test_detach_tree_keeps_structure_and_drops_graphs():
    from emip_amd.model.EMIP_short.model import _detach_tree
    w = torch.ones(3, requires_grad=True)
    a, b = w * 2, (w * 3).sum()
    tree = {"x": a, "pair": (b, [a, 5, None]), "s": "name"}
    out = _detach_tree(tree)
    assert out["s"] == "name" and out["pair"][1][1] == 5 and out["pair"][1][2] is None
    assert isinstance(out["pair"], tuple) and isinstance(out["pair"][1], list)
    for t in (out["x"], out["pair"][0], out["pair"][1][0]):
        assert t.grad_fn is None and not t.requires_grad
    assert torch.equal(out["x"], a.detach()) and a.grad_fn is not None


def test_non_refreshable_trainable_packs_are_registered_for_in_place_rebuild():
    from emip_amd import nn_base

    class Head(nn_base.EmipModule):
        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 1)
            self.frozen = nn.Linear(4, 1)
            self.frozen.weight.requires_grad_(False)

        def packs(self):
            pad = lambda a: torch.cat([a.detach(), a.new_zeros(7, a.shape[1])], 0)      # not a recorded permutation
            return (self.packed("pad", (self.lin.weight,), pad), self.packed("padf", (self.frozen.weight,), pad),
                    self.packed("plain", (self.lin.weight,), lambda a: nn_base.pack_linear(a, torch.float32)))

    m = Head().train()
    p1, p2, p3 = m.packs()
    assert p1.shape == (8, 4) and p3.data_ptr() == m.lin.weight.data_ptr()          # f32 mode: the plain "pack" is the parameter
    assert (id(m), "pad") in nn_base._REBUILD and (id(m), "padf") not in nn_base._REBUILD
    assert m._pack_cache["pad"].builder is not None and m._pack_cache["pad"].recs is None
    assert nn_base.packs_not_kept_current(m) == []
    del nn_base._REBUILD[(id(m), "pad")]
    assert nn_base.packs_not_kept_current(m) == [("", "pad")]
    # on the host nothing is rewritten in place (no device, no captured graph): the entry is rebuilt on its next use
    with torch.no_grad():
        m.lin.weight.add_(1.0)
    nn_base.refresh_packs()
    q1 = m.packs()[0]
    assert torch.equal(q1[:1], m.lin.weight.detach()) and q1.data_ptr() != p1.data_ptr()


def test_weight_gradient_queues_are_per_stream_and_reset_drops_everything():
    from emip_amd import ops
    q = ops.WgradQueue()
    a, b = torch.zeros(64, 16), torch.zeros(64, 8)
    c = torch.zeros(16, 8)
    w = nn.Parameter(torch.zeros(16, 8))
    q.MAX = 10 ** 6                                   # nothing may launch here
    q.add(a, b, c, None, 64, 16, 8, 16, 8, owners=((c, w), (None, None)))
    q.add(a, b, c, None, 64, 16, 8, 16, 8)
    assert list(q.queues.keys()) == [0] and q.queues[0][0] is None and len(q.items) == 2      # host tensors: one queue, no stream
    assert q.owners == [(c, w)]
    q.reset()
    assert q.queues == {} and q.items == [] and q.owners == [] and q.post == []
