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


def reproducible_limits(total_envs, family=None):
    """Kernel-family pins (`reset_coop_limit`, `step_coop_limit` of xarm_config / make()) that make a shard's result
    BITWISE independent of the world size.

    By default a handle picks its kernels from its OWN batch size and from COUNTS (include/xarm_hip.h: batches / reset
    lists of at most 8 192 envs run on the cooperative 16-lanes-per-env kernels, hand-off lists of more than
    XARM_EJECT_COOP_CAP envs on the one-env-per-lane kernel), so the 8 192-env shard of a 65 536-env job on 8 GPUs steps on
    k_step_coop while the same envs on one GPU step on the fast pipeline - the two agree to float32 rounding, and contact
    chaos then separates the trajectories.  With these limits passed to EVERY shard (and to the single-GPU run) which kernel
    handles an env is a function of the env's own state and of the job's config, never of a shard's size or of a count:
      family 'fast' (default when total_envs > 8 192): (total_envs, 1) - the fast pipeline at every shard size (pad-free
          fast step, every hand-off on the cooperative kernel) and the cooperative reset for every reset list.  The speed
          of the default choice at the BASELINE sizes, within a few per cent; a bulk reset() of a large batch is slower
          (cooperative kernel on every env);
      family 'lane': (-1, -1), the one-env-per-lane kernels everywhere; resets then cost six ticks of a lone wavefront
          (~3x slower per step call: the price of that pin);
      family 'coop' (default otherwise): (total_envs, total_envs), the cooperative kernels everywhere.
    Speed, not semantics, depends on the choice (tests/test_gpu_parity.py::test_world_size_invariance_at_the_baseline_split)."""
    from ._native import RESET_COOP_LIMIT_DEFAULT
    if family is None:
        family = "fast" if int(total_envs) > RESET_COOP_LIMIT_DEFAULT else "coop"
    if family == "fast":
        return {"reset_coop_limit": int(total_envs), "step_coop_limit": 1}
    if family == "lane":
        return {"reset_coop_limit": -1, "step_coop_limit": -1}
    if family == "coop":
        return {"reset_coop_limit": int(total_envs), "step_coop_limit": int(total_envs)}
    raise ValueError("family must be 'fast', 'lane' or 'coop'")


def make_shard(env_id, total_envs, rank=None, world_size=None, reproducible=False, **kwargs):
    """This rank's shard of a `total_envs` job: gym_xarm_amd.make with num_envs / env_id_offset from shard_range.
    reproducible=True (or 'fast' / 'lane' / 'coop') adds reproducible_limits(total_envs): bitwise the same per-env results at
    every world size, at the cost of the per-shard kernel choice."""
    import gym_xarm_amd
    r, _, w = env_from_torchrun()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    lo, hi = shard_range(total_envs, rank, world_size)
    if reproducible:
        kwargs.update(reproducible_limits(total_envs, None if reproducible is True else reproducible))
    return gym_xarm_amd.make(env_id, num_envs=hi - lo, env_id_offset=lo, **kwargs)


def env_from_torchrun():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


class Rank0Gather:
    """Observation gather to rank 0 for a single learner: per-rank rows [E_r, d] -> [sum E_r, d] on rank 0, ordered by
    global env id.  The shard sizes are exchanged ONCE, when the object is built (one small all_gather); every call
    after that is one send per rank into receive buffers rank 0 keeps, with no pickling and no host synchronisation
    beyond the transfers themselves.  Direct sends rather than a ring all-gather: on MI355X rank 0 has 7 inbound xGMI
    links that work in parallel, a ring would be per-link bound."""

    def __init__(self, rows, row_shape, dtype, device=None, total_envs=None, group=None):
        self.group, self.row_shape, self.dtype = group, tuple(row_shape), dtype
        self.active = dist.is_initialized() and dist.get_world_size(group) > 1
        self.rows = int(rows)
        if not self.active:
            return
        self.ws, self.rank = dist.get_world_size(group), dist.get_rank(group)
        cdev = _coll_device(device)
        mine = torch.tensor([self.rows], dtype=torch.int64, device=cdev)
        sizes = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(self.ws)]
        dist.all_gather(sizes, mine, group=group)
        self.sizes = [int(t.item()) for t in sizes]
        if total_envs is not None and sum(self.sizes) != int(total_envs):
            raise ValueError("shards hold %d envs, expected %d" % (sum(self.sizes), total_envs))
        self.out = None
        if self.rank == 0:
            self.out = torch.empty((sum(self.sizes),) + self.row_shape, dtype=dtype, device=device)
            offs = [0]
            for n in self.sizes:
                offs.append(offs[-1] + n)
            self.views = [self.out[offs[r]:offs[r + 1]] for r in range(self.ws)]

    def __call__(self, x):
        """x: this rank's rows.  Returns the gathered tensor on rank 0 (a buffer reused by the next call), None elsewhere."""
        if not self.active:
            return x
        if tuple(x.shape) != (self.rows,) + self.row_shape:
            raise ValueError("expected rows of shape %s, got %s" % ((self.rows,) + self.row_shape, tuple(x.shape)))
        if self.rank == 0:
            self.views[0].copy_(x)
            reqs = [dist.irecv(self.views[r], src=r, group=self.group) for r in range(1, self.ws)]
            for r in reqs:
                r.wait()
            return self.out
        dist.send(x.contiguous(), dst=0, group=self.group)
        return None


def gather_to_rank0(x, total_envs=None, group=None):
    """One-off form of Rank0Gather (sizes exchanged in this call); use the class for a gather per step."""
    g = Rank0Gather(x.shape[0], x.shape[1:], x.dtype, device=x.device if x.is_cuda else None, total_envs=total_envs, group=group)
    return g(x)


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
