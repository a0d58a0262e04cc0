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
    # the per-step form: sizes exchanged once at construction, buffers reused by every call
    from gym_xarm_amd.distributed import Rank0Gather
    gat = Rank0Gather(hi - lo, obs.shape[1:], torch.float64, total_envs=total)
    for k in range(3):
        gk = gat(torch.from_numpy(obs) + k)
        if rank == 0:
            assert torch.equal(gk, g + k)
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


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_rank_plumbing_two_ranks_gloo(tmp_path, scaling):
    """bench.py under torch.distributed.run with two ranks on CPU (gloo, a stand-in env): the launch line the driver
    uses, the barrier / max-over-ranks timing, the env-id sharding of both scaling modes and the one JSON line of rank 0"""
    import json
    import subprocess
    log = str(tmp_path / "stub")
    env = dict(os.environ, XARM_BENCH_DEVICE="cpu", XARM_BENCH_BACKEND="gloo", XARM_BENCH_ENV_FACTORY="bench_stub:make",
               XARM_BENCH_STUB_LOG=log, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "3",
           "--envs-per-gpu", "101", "--scaling", scaling, "--repeats", "2", "--no-cpu-baseline", "--aged-preroll", "25"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 3 and d["scaling"] == scaling and d["higher_is_better"] is True
    shards = [[json.loads(l) for l in open(log + ".%d" % r)] for r in range(2)]
    if scaling == "weak":
        assert d["config"]["total_envs"] == 202 and [s[0]["num_envs"] for s in shards] == [101, 101]
        assert [s[0]["env_id_offset"] for s in shards] == [0, 101]
    else:
        assert d["config"]["total_envs"] == 101 and [s[0]["num_envs"] for s in shards] == [51, 50]      # shard_range
        assert [s[0]["env_id_offset"] for s in shards] == [0, 51]
    k = 1
    if scaling == "weak":
        # world > 1 under weak scaling: the strong-scaling leg splits the configured env count (101) over the ranks
        assert [s[1]["num_envs"] for s in shards] == [51, 50] and [s[1]["env_id_offset"] for s in shards] == [0, 51]
        st = d["strong_scaling"]
        assert st["total_envs"] == 101 and st["envs_per_gpu"] == 51 and st["shard_range_rank0"] == [0, 51] and st["value"] > 0
        assert abs(st["value"] - 101 * 7 / (st["ms_per_step"] * 7e-3)) < 1e-6 * st["value"] and len(st["env_steps_per_sec"]) == 2
        k = 2
    else:
        assert "strong_scaling" not in d
    assert all(s[k]["auto_reset"] == "True" and s[k]["num_envs"] == s[0]["num_envs"] for s in shards)      # the lockstep-phase leg
    assert all(s[k + 1]["auto_reset"] == "lazy" for s in shards)                                         # the lazy leg ran too
    assert d["lockstep_phase"]["steps"] == 20 and d["lockstep_phase"]["value"] > 0
    ag = d["aged_state"]                                     # the same handle after the untimed pre-roll, same windows
    assert ag["preroll_steps"] == 25 + 3 + 2 * 7 and ag["value"] > 0 and len(ag["env_steps_per_sec"]) == 2
    assert ag["step_kernel_ms"] == 0.5 and ag["reset_kernels_ms"] == 0.25
    rf = d["roofline"]
    assert rf["bound"] == "valu/latency" and rf["reported_against"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert {"avg_ms", "share", "traffic", "resets_per_call", "algorithmic_bytes_per_launch", "achieved_GBs", "frac"} <= set(rf["kernels"]["reset"])
    assert abs(d["value"] - d["config"]["total_envs"] * 7 / (d["ms_per_step"] * 7e-3)) < 1e-6 * d["value"]
    assert len(d["repeats"]["env_steps_per_sec"]) == 2 and d["repeats"]["min"] <= d["value"] <= d["repeats"]["max"]
    assert d["roofline"]["kernels"]["reset"]["avg_ms"] == 0.25 and abs(d["roofline"]["kernel_avg_ms"] - 0.75) < 1e-9
    # desynchronised phases, 10-step episodes: each env finishes once per 10 steps
    assert abs(d["config"]["resets_per_step"] - d["config"]["total_envs"] / 10) <= d["config"]["total_envs"] / 10 * 0.5 + 2


def test_rank0_gather_reuses_sizes(tmp_path):
    """Rank0Gather exchanges the shard sizes once; without a process group it is the identity"""
    from gym_xarm_amd.distributed import Rank0Gather
    g = Rank0Gather(5, (3,), torch.float32)
    x = torch.randn(5, 3)
    assert g(x) is x and not g.active


def test_register_with_gym_mirrors_registry(monkeypatch):
    """with a `gym` importable, the reference's ids are registered there with its entry_point / max_episode_steps
    convention (gym_xarm/__init__.py:6-22); without one (this image) the hook is a no-op"""
    import types
    import gym_xarm_amd
    assert gym_xarm_amd._GYM_BACKENDS == [] or set(gym_xarm_amd._GYM_BACKENDS) <= {"gym", "gymnasium"}
    calls = []
    gym = types.ModuleType("gym")
    gym.envs = types.ModuleType("gym.envs")
    gym.envs.registry = {}
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda id, entry_point, max_episode_steps: calls.append((id, entry_point, max_episode_steps))
    gym.envs.registration = reg
    for name, mod in (("gym", gym), ("gym.envs", gym.envs), ("gym.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setitem(sys.modules, "gymnasium", None)        # import of gymnasium fails
    assert gym_xarm_amd.register_with_gym() == ["gym"]
    got = {c[0]: c for c in calls}
    assert got["XarmReach-v0"][1:] == ("gym_xarm_amd.envs:XarmReachEnv", 25)
    assert got["XarmHandover-v0"][1:] == ("gym_xarm_amd.envs:XarmHandover", 100)
    assert got["XarmPickAndPlace-v1"][1:] == ("gym_xarm_amd.envs:XarmPickAndPlace", 50)
    assert set(got) == set(gym_xarm_amd.registered_ids())


def test_reproducible_limits_are_functions_of_the_job_config():
    """the pins of gym_xarm_amd.distributed.reproducible_limits: 'fast' (default above 8 192 envs) = the fast pipeline at every
    shard size + the cooperative reset for every list; every shard of a job gets the same pair"""
    from gym_xarm_amd import distributed as D
    assert D.reproducible_limits(65536) == D.reproducible_limits(65536, "fast") == {"reset_coop_limit": 65536, "step_coop_limit": 1}
    assert D.reproducible_limits(65536, "lane") == {"reset_coop_limit": -1, "step_coop_limit": -1}
    assert D.reproducible_limits(4096) == D.reproducible_limits(4096, "coop") == {"reset_coop_limit": 4096, "step_coop_limit": 4096}
    with pytest.raises(ValueError):
        D.reproducible_limits(10, "warp")
