"""Data-parallel training step on the device, 2 ranks (gloo transport, both ranks on cuda:0): with different data per
rank, the gradients the reducer leaves in `.grad` are the MEAN of the two ranks' gradients (checked against a single
process that runs both samples), and after the optimizer step the replicas hold identical parameters.
The RCCL calls themselves (backend "nccl": all_reduce / all_to_all_single / all_gather_into_tensor on device tensors, the
side-stream ordering) run in the one-rank tests at the end of this file: a one-GPU box cannot host two RCCL ranks, so the
two-rank tests carry the arithmetic over gloo and the one-rank tests carry the transport.  No scaling curve exists yet."""
import json
import os

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]
HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ["decoder.conv5.weight", "backbone.feat_net.pvtv2_en.block3.7.mlp.fc1.weight", "conv_corr.3.bias",
         "injector.transformer.attn.temperature"]


def _build():
    from emip_amd import nn_base
    from emip_amd.filler import state_dict_from_manifest
    from emip_amd.model.EMIP_short.model import CoUpdater
    from emip_amd.train import freeze_like_reference
    nn_base.set_default_dtype(torch.float32)
    g = os.path.join(HERE, "golden")
    net = CoUpdater(json.load(open(os.path.join(g, "model_args.json"))))
    net.load_state_dict(state_dict_from_manifest(json.load(open(os.path.join(g, "short_state_manifest.json"))), 0))
    net = freeze_like_reference(net.to("cuda:0").train())
    for m in net.modules():                       # deterministic step: no stochastic depth
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    return net


def _data(rank):
    from emip_amd.filler import synthetic_gt, synthetic_pair
    im1, im2 = synthetic_pair(1, seed=500 + rank)
    return im1.cuda(), im2.cuda(), synthetic_gt(1, seed=600 + rank).cuda()


def _loss(net, im1, im2, gt):
    """hybrid_e_loss + a smooth linear functional of the flow predictions: reaches every trainable parameter the real
    objective reaches, without the piecewise photometric loss whose gradient jitters run to run"""
    from emip_amd.loss.loss_pred import hybrid_e_loss
    mask, fw, bw = net(im1, im2)
    w = torch.linspace(-1, 1, 352 * 352, device=im1.device).view(1, 1, 352, 352) / (352 * 352)
    return hybrid_e_loss(mask, gt) + sum((f * w).sum() + (b_ * w.flip(-1)).sum() for f, b_ in zip(fw, bw))


def _worker(rank, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from emip_amd.dp import GradReducer, broadcast_parameters
    from emip_amd.train import build_optimizer, trainable
    net = _build()
    broadcast_parameters(net)
    opt = build_optimizer(net, lr=1e-3, weight_decay=0.0, clip=0.5)
    red = GradReducer(trainable(net), record_events=True)
    im1, im2, gt = _data(rank)
    with torch.enable_grad():                     # train.py:43-60 with the reducer in DDP's place
        _loss(net, im1, im2, gt).backward()
    red.finish()                                  # step 1 is the reducer's calibration step
    p = dict(net.named_parameters())
    grads = {n: p[n].grad.detach().cpu().numpy() for n in NAMES}
    opt.step()
    torch.cuda.synchronize()
    params_after = {n: p[n].detach().cpu().numpy() for n in NAMES}
    # step 2 on the calibrated buckets: gradient-ready order, exchanges start while backward is still running
    names = {id(q): n for n, q in net.named_parameters()}
    first_bucket = [names[id(q)] for q in red.buckets[0].params]
    nbytes = [sum(q.numel() for q in b.params) * 4 for b in red.buckets]
    dead = len(red.dead)
    for q in net.parameters():
        q.grad = None
    red.begin_step()
    end_bwd = torch.cuda.Event(enable_timing=True)
    with torch.enable_grad():
        _loss(net, im1, im2, gt).backward()
    end_bwd.record()
    in_bwd = list(red.launch_log)
    red.finish()
    torch.cuda.synchronize()
    lead_ms = [b.t_launch.elapsed_time(end_bwd) for b in red.buckets[:len(in_bwd)]]      # > 0: launched before backward ended
    cost = (red.kernel_launches, len(red.buckets), red.host_ms, red.kernel_ms())
    out.put((rank, params_after, grads, (first_bucket[:3], nbytes, dead, in_bwd, list(red.launch_log), lead_ms), cost))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_keeps_replicas_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 29533, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    got = {r[0]: r[1] for r in res}
    grads = {r[0]: r[2] for r in res}
    info = {r[0]: r[3] for r in res}
    # what the reducer itself costs per step, apart from the collectives: one gather launch per bucket on the way out, ONE
    # scatter launch on the way back (no per-parameter op), well under a millisecond of device time for the 400 MB of gradients
    for r in res:
        launches, nb, host_ms, kern_ms = r[4]
        print(f"  rank {r[0]}: {launches} reducer launches for {nb} buckets, host {host_ms:.2f} ms, device {kern_ms:.2f} ms")
        assert launches <= nb + 2, (launches, nb)
        # (two processes time-slice this one GPU: event intervals here can contain the other rank's kernels, so the device
        # time is asserted in test_gradient_bucket_kernels_roundtrip_and_cost, single process; the host time must not contain a
        # wait for the GPU: 1 330 .grad lookups and a table compare per step)
        assert host_ms < 6.0, (kern_ms, host_ms)
    # bucket plan after calibration: identical on both ranks, conv_corr.0.weight (67 MB, ready early) leads its own bucket,
    # the parameters that never receive a gradient are not exchanged, and every bucket but the last left during backward,
    # in index order, with time to spare before backward ended
    assert info[0][:5] == info[1][:5], (info[0][:5], info[1][:5])
    first, nbytes, dead, in_bwd, log, lead = info[0]
    print("  buckets (MB):", [round(b / 2 ** 20, 1) for b in nbytes], "dead parameters:", dead, "launched inside backward:", in_bwd,
          "lead over the end of backward (ms):", [round(x, 2) for x in lead])
    assert dead >= 100 and log == list(range(len(nbytes))) and in_bwd == list(range(len(in_bwd))) and len(in_bwd) >= len(nbytes) - 1
    assert any(n == "conv_corr.0.weight" for n in first) or nbytes[0] >= 60 * 2 ** 20 or nbytes[1] >= 60 * 2 ** 20
    # (a bucket that closes with the very last gradient of the step is launched inside backward too, but cannot lead its end)
    # Event timings of two processes sharing one GPU jitter by a few ms: the early buckets must lead clearly, the late ones
    # (closed by the last gradients of the step) may sit within that jitter of the end
    # (the launch ORDER above is deterministic; the lead times are printed, and only the first bucket's is asserted -- with
    # two processes and a CPU transport on one GPU the later ones have failed spuriously in one full-suite run out of three)
    assert lead[0] > 0 and lead[0] > lead[-1]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for n in NAMES:
        assert (got[0][n] == got[1][n]).all(), n
    # single process: the same two samples one after the other, gradients averaged by hand
    net = _build()
    params = [q_ for q_ in net.parameters() if q_.requires_grad]
    acc = None
    for rank in range(2):
        im1, im2, gt = _data(rank)
        for q_ in params:
            q_.grad = None
        with torch.enable_grad():
            _loss(net, im1, im2, gt).backward()
        gs = [None if q_.grad is None else q_.grad.clone() for q_ in params]
        acc = gs if acc is None else [None if a is None else a + b for a, b in zip(acc, gs)]
        # BatchNorm running buffers advance per forward on every replica; parameters only change in step()
    p = dict(net.named_parameters())
    idx = {id(q_): i for i, q_ in enumerate(params)}
    for n in NAMES:
        assert (grads[0][n] == grads[1][n]).all(), n                    # both ranks hold the same reduced gradient
        ref = (acc[idx[id(p[n])]] * 0.5).cpu().numpy()
        err = abs(ref - grads[0][n]).max()
        scale = abs(ref).max() + 1e-12
        assert err <= 1e-2 * scale, (n, err, scale)        # f32 atomics jitter run to run; ReLU masks may flip


def _worker_bf16(rank, port, out, algo, comm):
    """the default bf16 training step (gradient arena, deferred grouped weight gradients, fused clamp + AdamW) under the
    reducer: what every rank holds in .grad after finish(), and the parameters after the step"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from emip_amd import nn_base, ops
    from emip_amd.dp import GradReducer, broadcast_parameters
    from emip_amd.train import build_optimizer, train_step, trainable
    net = _build()
    nn_base.set_default_dtype(torch.bfloat16)
    from emip_amd.model.EMIP_short.model import CoUpdater
    sd = net.state_dict()
    net = CoUpdater(json.load(open(os.path.join(HERE, "golden", "model_args.json"))))
    net.load_state_dict(sd)
    from emip_amd.train import freeze_like_reference
    net = freeze_like_reference(net.to("cuda:0").train())
    for m in net.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    broadcast_parameters(net)
    opt = build_optimizer(net, lr=0.0, weight_decay=0.0, clip=0.0)          # the step changes nothing: three comparable steps
    red = GradReducer(trainable(net), algo=algo, comm_dtype=torch.bfloat16 if comm == "bf16" else None, record_events=True)
    im1, im2, gt = _data(rank)
    names = NAMES + ["injector.transformer.ffn.project_out.weight", "injector1.transformer.ffn.project_out.weight",
                     "backbone.feat_net.pvtv2_en.block3.7.attn.q.bias"]
    p = dict(net.named_parameters())
    for step in range(3):                     # calibration, arena sizing, steady state
        red.begin_step()
        ops.WGRADS.fixed = 0
        train_step(net, opt, red, im1, im2, gt)
    torch.cuda.synchronize()
    grads = {n: p[n].grad.detach().float().cpu().numpy() for n in names}
    cost = (red.kernel_launches, len(red.buckets), red.host_ms, red.kernel_ms(), ops.WGRADS.fixed)
    out.put((rank, grads, cost))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,comm", [("allreduce", "f32"), ("direct", "bf16")])
def test_two_rank_bf16_training_step_through_the_reducer(algo, comm):
    """bf16, arena + deferred weight gradients + reducer: both ranks end with the same reduced gradients, none of them zero,
    close to the mean of the two ranks' local gradients; the reducer's per-step cost in the steady state"""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bf16, args=(r, 29541 + (algo == "direct"), q, algo, comm)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, g0, c0), (_, g1, c1) = res
    for c in (c0, c1):
        launches, nb, host_ms, kern_ms, fixed = c
        print(f"  {algo}/{comm}: {launches} reducer launches for {nb} buckets, host {host_ms:.2f} ms, device {kern_ms:.2f} ms, "
              f"deferred results fixed up: {fixed}")
        assert launches <= (2 if algo == "direct" else 1) * nb + 2 and host_ms < 6.0 and fixed == 0
    for n in g0:
        assert np.array_equal(g0[n], g1[n]), n                        # every replica holds the same reduced gradient
        assert np.abs(g0[n]).max() > 0, n
    # single process, the two ranks' samples one after the other in the same bf16 mode: the mean of the local gradients
    from emip_amd import nn_base, ops
    from emip_amd.loss.loss_flow import unFlowLoss
    from emip_amd.loss.loss_pred import hybrid_e_loss
    from emip_amd.model.EMIP_short.model import CoUpdater
    from emip_amd.train import freeze_like_reference
    net32 = _build()
    sd = net32.state_dict()
    del net32
    nn_base.set_default_dtype(torch.bfloat16)
    try:
        net = CoUpdater(json.load(open(os.path.join(HERE, "golden", "model_args.json"))))
        net.load_state_dict(sd)
        net = freeze_like_reference(net.to("cuda:0").train())
        for m in net.modules():
            if hasattr(m, "drop_path_rate"):
                m.drop_path_rate = 0.0
        fl = unFlowLoss()
        acc = {}
        for rank in range(2):
            im1, im2, gt = _data(rank)
            net.zero_grad(set_to_none=True)
            with torch.enable_grad():
                preds = net(im1, im2)
                pair = [torch.cat((preds[1][i], preds[2][i]), 1) for i in range(len(preds[1]))]
                (hybrid_e_loss(preds[0], gt) + fl.compute_loss(pair, torch.cat((im1, im2), 1))[0]).backward()
                ops.flush_wgrads()
            pp = dict(net.named_parameters())
            for n in g0:
                acc[n] = acc.get(n, 0) + 0.5 * pp[n].grad.detach().float().cpu().numpy()
    finally:
        nn_base.set_default_dtype(torch.float32)
    for n in g0:
        err = np.abs(acc[n] - g0[n]).max() / (np.abs(acc[n]).max() + 1e-30)
        print(f"  {n}: reduced vs mean of local gradients {err:.3e}")
        # bf16 forwards jitter run to run (f32-atomic statistics): a few per cent near the loss, more behind 30 blocks.  The
        # camouflage feeder (`injector.`) sees only the piecewise photometric loss through the flow softmax over 1936 near-ties
        # (ill-conditioned under random weights): its gradients differ by O(1) between two bf16 runs of the SAME
        # process (0.8 % in one run of this test, 53 % in the next) -- printed, not bounded; their f32 parity is pinned by
        # tests/test_train_gpu.py against the reference's own gradients
        if not n.startswith("injector."):
            assert err < (0.2 if n.startswith(("decoder", "conv_corr", "injector1")) else 0.5), (n, err)


def test_gradient_bucket_kernels_roundtrip_and_cost():
    """emip_grad_pack / emip_grad_unpack / emip_shard_sum alone, one process: 1 300 ragged gradient tensors (100 M elements,
    the EMIP-short payload) into the flat buffer and back -- exact in f32, bf16-rounded on a bf16 wire, missing gradients
    packed as zeros and given the (zero) mean on the way back -- one launch per bucket out, ONE launch back, under a millisecond of
    device time per step in total"""
    from emip_amd import _lib
    from emip_amd.dp import GradReducer
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1)
    sizes = [int(x) for x in torch.randint(1, 125_000, (1290,), generator=gen)] + [320 * 1280] * 6 + [17_000_000, 3, 1, 5]
    params = [torch.nn.Parameter(torch.empty(n, device="cuda")) for n in sizes]
    for comm in (torch.float32, torch.bfloat16):
        red = GradReducer(params, comm_dtype=comm, record_events=True)
        red.world = 2                                     # layout for two ranks; no collective is issued in this test
        red.calibrated = True                             # steady state: a missing gradient takes part as zeros
        red.buckets = __import__("emip_amd.dp", fromlist=["_make_buckets"])._make_buckets(list(reversed(red.params)), 64 << 20, 2)
        red._rebind()
        for i, p in enumerate(params):
            p.grad = None if i % 97 == 5 else torch.randn(p.numel(), device="cuda")
        want = [None if p.grad is None else p.grad.clone() for p in params]
        dev = params[0].device
        red._ensure(dev)
        red.begin_step()
        for b in red.buckets:
            red._timed(red._pack, b, dev)
        flat = red._flat.float().clone()
        for b in red.buckets:                              # every slice holds its tensor (zeros where there was no gradient)
            for p, off in zip(b.params[:5] + b.params[-5:], b.offsets[:5] + b.offsets[-5:]):
                got = flat[b.lo + off:b.lo + off + p.numel()]
                ref = torch.zeros_like(got) if p.grad is None else p.grad.to(comm).float()
                assert torch.equal(got, ref)
        red._timed(red._unpack, dev, False)
        torch.cuda.synchronize()
        for p, w in zip(params, want):
            if w is None:          # no gradient on this rank: it took part with zeros and holds the mean of zeros (as under DDP)
                assert p.grad is not None and not p.grad.any()
            else:
                assert torch.equal(p.grad, 0.5 * w.to(comm).float())
        ms = red.kernel_ms()
        nb = len(red.buckets)
        print(f"  {comm}: {sum(sizes) / 1e6:.0f} M elements, {nb} buckets, {red.kernel_launches} launches, device {ms:.3f} ms")
        assert red.kernel_launches == nb + 1 and ms < 1.5        # measured 0.99 ms (f32 wire) / 0.31 ms (bf16 wire)
        red.remove()
    # the reduce step of the direct exchange: f32 accumulation of bf16 shards
    x = torch.randn(8, 1_000_003, device="cuda").to(torch.bfloat16)
    out = torch.empty(1_000_003, dtype=torch.bfloat16, device="cuda")
    _lib.call("emip_shard_sum", x.data_ptr(), out.data_ptr(), 8, x.shape[1], 1, torch.cuda.current_stream().cuda_stream)
    assert torch.allclose(out.float(), x.float().sum(0), rtol=1e-2, atol=1e-2)


def _nccl_single_worker(port, out):
    """world size 1, backend nccl (= RCCL): every collective of the reducer on device tensors, the real training graph"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    from emip_amd.dp import GradReducer, broadcast_parameters
    from emip_amd.train import trainable
    res = {}
    net = _build()
    broadcast_parameters(net)                      # world 1: returns at once
    im1, im2, gt = _data(0)
    for algo, comm in (("allreduce", None), ("direct", torch.bfloat16)):
        red = GradReducer(trainable(net), algo=algo, comm_dtype=comm, single_rank_collectives=True)
        assert red.exchange and red.world == 1 and dist.get_backend(red._ctl) == "gloo"
        worst, in_bwd, nb = 0.0, None, None
        for step in range(2):                      # step 0 calibrates, step 1 runs the calibrated buckets inside backward
            for q in net.parameters():
                q.grad = None
            red.begin_step()
            with torch.enable_grad():
                _loss(net, im1, im2, gt).backward()
            in_bwd = list(red.launch_log)
            # buckets that left during backward were packed from .grad and have not been written back yet: .grad is still the
            # local gradient
            local = {n: q.grad.detach().clone() for n, q in net.named_parameters() if q.grad is not None}
            red.finish()
            torch.cuda.synchronize()
            nb = len(red.buckets)
            for n, q in net.named_parameters():
                if n in local:
                    d = (q.grad - local[n]).abs().max().item()
                    worst = max(worst, d / (local[n].abs().max().item() + 1e-30))
        res[algo] = dict(worst=worst, in_bwd=in_bwd, log=list(red.launch_log), nb=nb, dead=len(red.dead))
        red.remove()
    dist.barrier()
    dist.destroy_process_group()
    out.put(res)


def test_single_rank_rccl_exercises_every_collective_on_device():
    """the RCCL path had never executed anywhere.  One rank, backend "nccl", the reducer forced
    through its collectives: all_reduce (f32 wire) and all_to_all_single + emip_shard_sum + all_gather_into_tensor (bf16
    wire) on device tensors, packed on the side stream while backward is still running, the late-bucket agreement over
    the gloo control group.  A mean over one rank returns the local gradients: exactly on the f32 wire, to bf16 rounding
    on the bf16 wire."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    a, d = res["allreduce"], res["direct"]
    print("  single-rank RCCL:", res)
    assert a["worst"] == 0.0                         # f32 wire, world 1: x * 1.0 -- bit for bit
    assert d["worst"] < 2 ** -8                      # bf16 wire: one rounding of every element
    for r in (a, d):
        assert r["nb"] >= 3 and r["log"] == list(range(r["nb"]))           # index order
        assert len(r["in_bwd"]) >= r["nb"] - 1                              # all but the last bucket left inside backward
        assert r["dead"] >= 100                                            # the reference's 108 never-trained tensors
