"""Multi-process path on CPU (gloo, world_size 2): contiguous env sharding, world-size-invariant
RNG keyed by the global env id, observation gather to rank 0, max-over-ranks timing reduction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from gym_xarm_amd.distributed import shard_range
    for total in (1, 7, 64, 65536, 65537):
        for ws in (1, 2, 3, 8):
            r = [shard_range(total, k, ws) for k in range(ws)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(ws - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from gym_xarm_amd.distributed import shard_range, gather_to_rank0, max_over_ranks, sum_over_ranks
    from oracle import oracle as O
    lo, hi = shard_range(total, rank, ws)
    env = O.OraclePnP(hi - lo, seed=13, env_id_offset=lo)   # CPU stand-in for the per-rank handle
    obs, ag, dg = env.reset()
    a = np.random.default_rng(5).uniform(-1, 1, size=(total, 4))[lo:hi]
    obs, ag, dg, rew, done, succ = env.step(a)
    g = gather_to_rank0(torch.from_numpy(obs), total_envs=total)
    t = max_over_ranks(1.0 + rank)
    n = sum_over_ranks(hi - lo)
    if rank == 0:
        q.put((g.numpy(), t, n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process():
    total, ws = 12, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, total, q)) for r in range(ws)]
    for p in procs:
        p.start()
    gathered, tmax, n = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    env = O.OraclePnP(total, seed=13)
    env.reset()
    a = np.random.default_rng(5).uniform(-1, 1, size=(total, 4))
    obs = env.step(a)[0]
    assert np.array_equal(gathered, obs)          # bitwise: sharding does not change any env
    assert tmax == 2.0 and n == total
