"""Round 4: the bf16 inference forward is reproducible BIT FOR BIT.  Rounds 1-3 built LayerNorm / InstanceNorm statistics and the
MDTA Gram sums with f32 / f64 atomics: the sums differed in the last bit from run to run, a bf16 rounding of a normalised value
flipped now and then, and two forwards of the same batch ended 0.3 apart on logits spanning 8 (mask IoU 0.985).  Now every
reduction has a fixed order (gemm8.hip: the waves of a tile row meet through LDS; mlp_band.hip / sra_block.hip: a row is owned by
one wave; mdta.hip: slot partials added in slot order; pointwise.hip: block partials rounded to a grid on which f64 addition is
exact), and two runs give the same bits: eager against eager, graph replay against eager, steps in flight against eager."""
import pytest
import torch

from emip_amd.filler import synthetic_pair

pytestmark = pytest.mark.gpu


def _net(model_args, sd):
    from emip_amd import nn_base
    from emip_amd.model.EMIP_short.model import CoUpdater
    nn_base.set_default_dtype(torch.bfloat16)
    net = CoUpdater(model_args)
    net.load_state_dict(sd)
    return net.to("cuda:0").eval()


def _flat(d, prefix=""):
    """every tensor of the forward's intermediates (stage outputs, GMFlow features, prompts, correlation, decoder inputs)"""
    out = {}
    items = d.items() if isinstance(d, dict) else enumerate(d)
    for k, v in items:
        name = f"{prefix}{k}"
        if torch.is_tensor(v):
            out[name] = v.clone()
        elif isinstance(v, (dict, list, tuple)):
            out.update(_flat(v, name + "."))
    return out


@pytest.mark.parametrize("pairs", [16, 2])
def test_two_eager_bf16_forwards_are_bit_identical(model_args, short_sd, pairs):
    from emip_amd import nn_base
    try:
        net = _net(model_args, short_sd)
        im1, im2 = synthetic_pair(pairs, seed=1234)
        im1, im2 = im1.cuda(), im2.cuda()
        outs = []
        with torch.no_grad():
            for k in range(3):
                mask, fw, bw = net(im1, im2)
                outs.append((mask.clone(), fw[0].clone(), bw[0].clone(), _flat(net.last)))
                if k == 0:       # another stream / another allocation pattern between the runs
                    torch.cuda.empty_cache()
                    _ = torch.empty(123457, device="cuda")
        torch.cuda.synchronize()
        for k in (1, 2):
            for name in outs[0][3]:
                if name in outs[k][3] and outs[0][3][name].shape == outs[k][3][name].shape:
                    assert torch.equal(outs[0][3][name], outs[k][3][name]), (pairs, k, name)
            for a, b, what in zip(outs[0][:3], outs[k][:3], ("mask", "flow_fw", "flow_bw")):
                assert torch.equal(a, b), (pairs, k, what, (a.float() - b.float()).abs().max().item())
    finally:
        nn_base.set_default_dtype(torch.float32)


def test_steps_in_flight_reproduce_the_eager_bits(model_args, short_sd):
    """the benchmark's arrangement (whole-batch graphs, four steps in flight): every slot's last replay = the eager forward"""
    from emip_amd import nn_base
    from emip_amd.graph import PipelinedShort
    try:
        net = _net(model_args, short_sd)
        im1, im2 = synthetic_pair(16, seed=77)
        im1, im2 = im1.cuda(), im2.cuda()
        with torch.no_grad():
            o = net(im1, im2)
            ref, ref_fw, ref_bw = o[0].clone(), o[1][0].clone(), o[2][0].clone()
        runner = PipelinedShort(net, 16, inflight=4)
        runner.load(im1, im2)
        torch.cuda.synchronize()
        # 30 rounds x 4 slots: round 3's build failed ~4 % of such samples (one wrong channel for 16 tokens out of emip_ffn_block's
        # packed-f32 LayerNorm epilogue, only with other steps' kernels on the chip: emip_amd/csrc/Makefile) -- inside the
        # run-to-run band its tests allowed; a bit-exact comparison sees it
        for rnd in range(30):
            for _ in range(11):
                runner.replay_free()
            torch.cuda.synchronize()
            for slot in range(4):
                m, fw, bw = runner.outputs(slot)
                assert torch.equal(m, ref), (rnd, slot, (m.float() - ref.float()).abs().max().item())
                assert torch.equal(fw[0], ref_fw) and torch.equal(bw[0], ref_bw), (rnd, slot)
        # the runner's latency graph (PVT stages 3-4 on a forked branch of the graph): the same bits, alone and after in-flight work
        for _ in range(3):
            m, fw, bw = runner.replay_alone()
            assert torch.equal(m, ref) and torch.equal(fw[0], ref_fw) and torch.equal(bw[0], ref_bw)
            runner.replay_free()
        torch.cuda.synchronize()
        from emip_amd.model.EMIP_short import model as M
        prev, M.FORK_DEEP = M.FORK_DEEP, True                     # the eager forked forward (real streams, no graph)
        try:
            with torch.no_grad():
                o = net(im1, im2)
            torch.cuda.synchronize()
            assert torch.equal(o[0], ref) and torch.equal(o[1][0], ref_fw)
        finally:
            M.FORK_DEEP = prev
    finally:
        nn_base.set_default_dtype(torch.float32)
