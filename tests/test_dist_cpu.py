"""N>1 bookkeeping of bench.py on CPU: two gloo ranks (127.0.0.1 rendezvous)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    from emip_amd import dist as ed
    d = ed.init("gloo")
    w, r, _ = ed.env_world()
    assert (w, r) == (world, rank)
    lo, hi = ed.shard_range(37, world, rank)
    d.barrier()
    tmax = ed.max_over_ranks(0.010 * (rank + 1))          # the slowest rank defines the step time
    total = ed.sum_over_ranks(hi - lo)
    out.put((rank, lo, hi, tmax, total, ed.pair_seed(1234, rank)))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, tot0, s0), (r1, lo1, hi1, t1, tot1, s1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37)        # disjoint, exhaustive, sizes differ by <= 1
    assert abs(t0 - 0.020) < 1e-12 and abs(t1 - 0.020) < 1e-12
    assert tot0 == tot1 == 37
    assert s0 != s1


def test_shard_range_properties():
    from emip_amd.dist import shard_range
    for n in (0, 1, 7, 16, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _dp_worker(rank, world, port, out, algo, comm):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from emip_amd.dp import GradReducer, broadcast_parameters
    dist.init_process_group("gloo")
    torch.manual_seed(rank)                      # different init per rank -> broadcast must equalise
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 2))
    dead = torch.nn.Parameter(torch.ones(3))     # never receives a gradient (like the reference's dead modules)
    broadcast_parameters(net)
    w0 = net[0].weight.detach().clone()
    params = list(net.parameters()) + [dead]
    red = GradReducer(params, bucket_bytes=300, algo=algo,            # tiny buckets: several buckets, one incomplete
                      comm_dtype=torch.bfloat16 if comm == "bf16" else None)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(5, 8, generator=g)
    net(x).pow(2).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    red.finish()                                 # step 1 = calibration: ready order logged, dead parameters found
    first = [p.grad.clone().numpy() for p in net.parameters()]
    assert red.calibrated and [id(p) for p in red.dead] == [id(dead)]
    assert all(id(p) != id(dead) for b in red.buckets for p in b.params)
    # gradient-ready order of this graph: last layer first (bias before weight is autograd's accumulation order per layer)
    order_names = [red.ready_order[i] for i in range(len(red.ready_order))]
    # a second step on the calibrated buckets: buckets complete and launch in index order, identically on every rank
    for p in net.parameters():
        p.grad = None
    red.begin_step()
    net(x * 0.5).pow(2).sum().backward()
    local2 = [p.grad.clone() for p in net.parameters()]
    launched_during_backward = list(red.launch_log)
    red.finish()
    out.put((rank, w0.numpy(), [t.numpy() for t in local], first, dead.grad is None, len(red.buckets), order_names,
             launched_during_backward, list(red.launch_log), [t.numpy() for t in local2],
             [p.grad.clone().numpy() for p in net.parameters()]))   # numpy: tensors in a Queue are shared-memory handles
    # a parameter that was dead in the calibration step gets a gradient later: reduced in a trailing bucket, still the mean
    for p in params:
        p.grad = None
    (net(x).pow(2).sum() + (dead * (rank + 1.0)).sum()).backward()
    red.finish()
    assert torch.allclose(dead.grad, torch.full((3,), 1.5)), dead.grad
    # a rank whose local graph lacks gradients of REGULAR buckets (rank 1 skips the two last layers here, so
    # their buckets -- the first ones in ready order -- never complete in its hooks while later buckets do) must still issue
    # the same collectives in the same order and end with the same mean as rank 0 (zeros for what it did not compute)
    for p in params:
        p.grad = None
    red.begin_step()
    if rank == 0:
        net(x).pow(2).sum().backward()
    else:
        net[1](net[0](x)).pow(2).sum().backward()
    mine = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    in_bwd = list(red.launch_log)
    red.finish()
    out.put((rank, "skip", [None if t is None else t.numpy() for t in mine], in_bwd, list(red.launch_log),
             [p.grad.clone().numpy() for p in net.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,comm", [("allreduce", "f32"), ("direct", "f32"), ("direct", "bf16")])
def test_grad_reducer_two_ranks_mean_and_unused_params(algo, comm):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, algo, comm)) for r in range(2)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(4)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    res = sorted((t for t in got if not isinstance(t[1], str)), key=lambda t: t[0])
    skip = sorted((t for t in got if isinstance(t[1], str)), key=lambda t: t[0])
    (_, w0a, la, ra, da, nb, oa, lda, lfa, l2a, r2a), (_, w0b, lb, rb, db, _, ob, ldb, lfb, l2b, r2b) = res
    import numpy as np
    tol = 1e-6 if comm == "f32" else 2e-2                          # bf16 transport: 8-bit mantissa on the wire, f32 sums
    assert np.array_equal(w0a, w0b)                                # broadcast equalised the replicas
    assert nb >= 3 and da and db                                   # dead parameter: skipped, no gradient invented
    for a, b, x, y in zip(la, lb, ra, rb):
        assert np.allclose(x, (a + b) / 2, atol=tol * max(1.0, np.abs(a).max())) and np.array_equal(x, y)
    for a, b, x, y in zip(l2a, l2b, r2a, r2b):
        assert np.allclose(x, (a + b) / 2, atol=tol * max(1.0, np.abs(a).max())) and np.array_equal(x, y)
    # calibration: both ranks hold the same ready order (the last layer's parameters first) and the same bucket order;
    # after it, every bucket but possibly the last was launched from inside backward, in index order, on both ranks
    assert oa == ob and oa[0] in (4, 5) and oa[-1] in (0, 1)
    assert lda == ldb and lfa == lfb == list(range(nb))
    assert lda == list(range(len(lda))) and len(lda) >= nb - 1
    # the step in which rank 1 skipped two layers: same launch order on both ranks (index order; rank 1 launched nothing
    # out of order from its hooks), same result on both, = rank 0's gradient / 2 where rank 1 had none, the mean elsewhere
    (_, _, m0, b0, f0, r0), (_, _, m1, b1, f1, r1) = skip
    assert f0 == f1 == list(range(nb)) and b1 == list(range(len(b1))) and len(b1) < nb
    assert sum(t is None for t in m1) == 4 and all(t is not None for t in m0)
    for a, b, x, y in zip(m0, m1, r0, r1):
        want = a / 2 if b is None else (a + b) / 2
        assert np.allclose(x, want, atol=tol * max(1.0, np.abs(a).max())) and np.array_equal(x, y)


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    e = dict(os.environ, EMIP_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=e,
                       timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher environment must itself start 2 ranks (as fresh child processes) and
    rank 0 must print ONE JSON line with n_gpus = 2; the slowest rank defines the step time (dry run: no GPU)"""
    r, recs = _bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2 and recs[0]["steps"] == 4
    assert recs[0]["ms_per_step"] >= 3.9            # rank 1 sleeps 4 ms per step, rank 0 only 2
    r1, recs1 = _bench("--steps", "4", "--warmup", "1", "--dry-run")
    assert r1.returncode == 0 and len(recs1) == 1 and recs1[0]["n_gpus"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r, recs = _bench("--gpus", "4", "--dry-run", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not recs
