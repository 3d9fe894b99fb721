"""Multi-GPU plumbing of the EMIP path (one process per GPU, torch.distributed over RCCL/xGMI).

Inference shards by frame pair / video stream: every rank runs an independent replica on its own pairs and
NO data-path collective exists.  What is shared is bookkeeping: which pairs a rank owns,
a barrier around the timed region and the max-over-ranks step time.  These helpers are backend-agnostic so the
N>1 logic is covered by world_size-2 gloo tests on CPU."""
import os

import torch


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return dist


def shard_range(n_items, world, rank):
    """Contiguous shard [lo, hi) of n_items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pair_seed(base_seed, rank):
    """Synthetic-input seed of a rank (weak scaling: every rank gets its own batch of pairs)."""
    return base_seed + rank


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float over the default process group (identity when not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def sum_over_ranks(value, device="cpu"):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.item()
