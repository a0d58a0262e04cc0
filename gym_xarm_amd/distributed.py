"""One process per GPU: environments are independent, so the batch shards by contiguous global
env ranges with NO data-path collective.  The per-env RNG is keyed by the global env
id (xarm_config.env_id_offset), which makes results independent of the world size.  The only
collective offered is the optional observation gather to rank 0 for a single learner
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
    """Contiguous range [lo, hi) of global env ids owned by `rank`; sizes differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, rem = divmod(int(total_envs), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_from_torchrun():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def gather_to_rank0(x, total_envs=None, group=None):
    """Gather per-rank rows [E_r, d] to rank 0 -> [sum E_r, d] (rows ordered by global env id).
    Direct gather (every rank sends once to rank 0) rather than a ring all-gather: on MI355X rank 0
    has 7 inbound xGMI links that work in parallel, a ring would be per-link bound."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return x
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [None] * ws
    dist.all_gather_object(sizes, int(x.shape[0]), group=group)
    if rank == 0:
        bufs = [torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device) for n in sizes]
        bufs[0].copy_(x)
        reqs = [dist.irecv(bufs[r], src=r, group=group) for r in range(1, ws)]
        for r in reqs:
            r.wait()
        out = torch.cat(bufs, dim=0)
        if total_envs is not None:
            assert out.shape[0] == total_envs
        return out
    dist.send(x.contiguous(), dst=0, group=group)
    return None


def _coll_device(device):
    """gloo reduces host tensors, RCCL device tensors"""
    return None if dist.get_backend() == "gloo" else device


def max_over_ranks(value, device=None, group=None):
    """MAX all-reduce of a python float (bench timing)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value, device=None, group=None):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
