#!/usr/bin/env python3
"""Headline benchmark: env steps/sec of XarmPDPickAndPlace-v0 (BASELINE.json `metric`).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu E] [--workload pnp|reach|handover|stack]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one env step of EVERY environment of the job (one xarm_step call per rank), including
the automatic resets of finished episodes.  Weak scaling: each rank owns `--envs-per-gpu` (default
65 536, the configuration the metric is quoted on) independent environments, global env ids
rank*E .. (rank+1)*E-1, no data-path collective.  Inputs are synthetic: a ring of
64 pre-generated uniform[-1,1] action tensors resident in HBM, so the timed region contains no RNG
and no host->device traffic.  Rank 0 prints ONE JSON line.

`--workload` selects one of the other BASELINE.json configs for the same measurement (same JSON schema, its own
metric name): reach = config 2 (XarmReach-v0, 4 096 envs), stack = config 4 (XarmPDStackTower-v0, 8 192 envs per
GPU), handover = config 5 (XarmPDHandover-v0, 16 384 envs per GPU).  The default (pnp, 65 536 envs per GPU) is the
configuration BASELINE.json's `metric` is quoted on and the only one the driver runs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, 6.3 TB/s achievable)
VALU_PEAK_TFLOPS = 157.3       # fp32 vector peak (256 CU x 4 SIMD x 16 lanes x 2 pk x 2 fma x 2.4 GHz), the bound that binds
ALGO_BYTES_PER_ENV_STEP = 452  # SURVEY.md 8(d), PnP N=1: state in+out, action in, obs/goals/reward/flags out
# workload -> (env id, default envs per GPU, action width, SURVEY 8(d) algorithmic bytes per env step, step kernel,
#              oracle class, CPU-baseline sample (envs per thread, steps), config dict)
WORKLOADS = {
    "pnp": ("XarmPDPickAndPlace-v0", 65536, 4, ALGO_BYTES_PER_ENV_STEP, "k_step", "OraclePnP", (128, 60),
            dict(GUI=False, num_obj=1, reward_type="sparse", init_grasp_rate=0.0, goal_ground_rate=0.0, goal_shape="air")),
    "reach": ("XarmReach-v0", 4096, 4, 336, "k_reach_step", "OracleReach", (256, 100), None),
    "handover": ("XarmPDHandover-v0", 16384, 8, 648, "k_ho_step", "OracleHandover", (16, 30),
                 dict(GUI=False, num_obj=1, same_side_rate=0.5, goal_shape="ground", use_stand=False)),
    "stack": ("XarmPDStackTower-v0", 8192, 8, 1040, "k_st_step", "OracleStackTower", (64, 60), None),
}


WORKLOAD_NAMES = {
    "pnp": "XarmPDPickAndPlace-v0 (XarmPickAndPlace, num_obj=1, sparse reward, goal_shape=air)",
    "reach": "XarmReach-v0 (XarmReachEnv, sparse reward; BASELINE config 2)",
    "handover": "XarmPDHandover-v0 (XarmHandover, num_obj=1, goal_shape=ground, same_side_rate=0.5; BASELINE config 5)",
    "stack": "XarmPDStackTower-v0 (XarmStackTowerEnv, three cubes, sparse reward; BASELINE config 4)",
}
SUBSTEPS = {"pnp": 15, "reach": 20, "handover": 15, "stack": 15}   # internal substeps (Handover: 15 ticks of one substep)


def cpu_baseline(workload="pnp"):
    """The CPU oracle (a restatement = kind "port"; PyBullet itself is absent) on the host cores:
    every thread steps its own shard through ctypes (the GIL is released inside the C call)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    env_id, _, act_dim, _, _, cls, (sample_envs_per_thread, steps), _ = WORKLOADS[workload]
    cores = max(1, min(os.cpu_count() or 1, 16))
    envs = [getattr(O, cls)(sample_envs_per_thread, seed=0, env_id_offset=k * sample_envs_per_thread) for k in range(cores)]
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(steps, cores, sample_envs_per_thread, act_dim))

    def work(k):
        envs[k].reset()
        t0 = time.perf_counter()
        for s in range(steps):
            envs[k].step(acts[s, k])
        return time.perf_counter() - t0
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(work, range(cores)))
        wall = time.perf_counter() - t0
        # resets are outside the per-thread timers; redo the timed part alone for the rate
        t0 = time.perf_counter()
        list(ex.map(lambda k: [envs[k].step(acts[s, k]) for s in range(steps)], range(cores)))
        wall_steps = time.perf_counter() - t0
    n = cores * sample_envs_per_thread * steps
    return {"value": n / wall_steps, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps of %s on the CPU oracle (float64, gcc -O2), %d threads, %.1f s"
                      % (cores * sample_envs_per_thread, steps, env_id, cores, wall + wall_steps),
            "reference": "unavailable (pybullet not importable)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="pnp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lazy", action="store_true", help="skip the extra measurement of the opt-in lazy auto-reset mode")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import gym_xarm_amd
    from gym_xarm_amd import distributed as D

    rank, local_rank, world = D.env_from_torchrun()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    # rehearsal hooks for a 1-GPU box: XARM_BENCH_DEVICE pins every rank to one device, XARM_BENCH_BACKEND=gloo
    # replaces RCCL (two ranks cannot share a device under RCCL); the driver's real runs use neither
    dev_index = int(os.environ.get("XARM_BENCH_DEVICE", local_rank))
    backend = os.environ.get("XARM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    env_id, default_E, act_dim, algo_bytes_per_step, kernel_name, _, _, env_config = WORKLOADS[args.workload]
    E = args.envs_per_gpu or default_E
    env = gym_xarm_amd.make(env_id, num_envs=E, seed=0, env_id_offset=rank * E, device=dev, config=env_config)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    ring = [torch.rand(E, act_dim, device=dev, generator=gen) * 2 - 1 for _ in range(64)]
    env.reset()
    n_done = torch.zeros((), device=dev)
    for i in range(args.warmup):
        # the warm-up runs exactly what the timed loop runs (torch loads the code object of a kernel at its first
        # launch: round 1 timed the first `done.sum()` inside the window, 50-80 ms of module loading)
        obs, rew, done, info = env.step(ring[i % 64])
        n_done += done.sum()
    torch.cuda.synchronize()
    env.timing_enable(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_done = torch.zeros((), device=dev)
    for i in range(args.steps):
        obs, rew, done, info = env.step(ring[(args.warmup + i) % 64])
        n_done += done.sum()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, device=dev)
    total_envs = int(D.sum_over_ranks(E, device=dev))
    kstep_ms_total, launches = env.timing_read()
    kstep_ms = kstep_ms_total / max(launches, 1)
    kstep_ms = D.max_over_ranks(kstep_ms, device=dev)
    resets = D.sum_over_ranks(float(n_done.item()), device=dev)

    # extra (pnp only): the opt-in lazy auto-reset mode (include/xarm_hip.h XARM_AUTO_RESET_LAZY) - a different contract
    # from the reference's VecEnv, so it never feeds `value`; useful = env steps that are not reset ticks
    lazy = None
    if args.workload == "pnp" and not args.no_lazy:
        env.close()
        lenv = gym_xarm_amd.make(env_id, num_envs=E, seed=0, env_id_offset=rank * E, device=dev, config=env_config, auto_reset="lazy")
        lenv.reset()
        for i in range(args.warmup):
            lenv.step(ring[i % 64])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        useful = torch.zeros((), device=dev)
        for i in range(args.steps):
            _, _, _, linfo = lenv.step(ring[(args.warmup + i) % 64])
            useful += (~linfo["resetting"]).sum()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ldt = D.max_over_ranks(time.perf_counter() - t0, device=dev)
        luse = D.sum_over_ranks(float(useful.item()), device=dev)
        lazy = {"value": luse / ldt, "unit": "useful env steps/s (reset ticks excluded)", "ms_per_step": ldt / args.steps * 1e3,
                "useful_fraction": luse / (total_envs * args.steps),
                "note": "opt-in auto_reset='lazy': a finished env runs the reference's six reset ticks one per step call; "
                        "same reset state, different VecEnv contract - not comparable with `value`"}
        lenv.close()
        env = None
    if rank == 0:
        value = total_envs * args.steps / dt
        algo_bytes = algo_bytes_per_step * E                     # one step-kernel launch processes E env steps
        achieved = algo_bytes / (kstep_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from the rocprofv3 --pmc passes
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("%s_hbm_bytes_per_launch_%d" % (kernel_name, E))
            except Exception:
                traffic = None
        # secondary ceiling (the one that actually binds): fp32 vector issue.  Wave-instruction count per launch
        # from the committed SQ_INSTS_VALU PMC pass, 64 lanes x 2 flop upper bound per instruction.
        valu = None
        if os.path.exists(pmc):
            try:
                n_valu = json.load(open(pmc)).get("%s_valu_wave_insts_per_launch_%d" % (kernel_name, E))
                if n_valu:
                    tf = n_valu * 64 * 2 / (kstep_ms * 1e-3) / 1e12
                    valu = {"achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s (<=2 flop per lane-instruction)",
                            "frac": tf / VALU_PEAK_TFLOPS, "wave_insts_per_launch": n_valu}
            except Exception:
                valu = None
        out = {
            "metric": "env steps/sec (whole node), %s" % env_id, "value": value, "unit": "env steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_NAMES[args.workload],
                       "envs_per_gpu": E, "total_envs": total_envs, "substeps_per_step": SUBSTEPS[args.workload], "solver_iterations": 50,
                       "auto_reset": True, "episodes_reset_in_window": int(resets), "parallelism": "env-shard x%d, no collective" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
                         "kernel_avg_ms": kstep_ms, "kernel_launches": int(launches),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "fused step: HBM is touched once per env step, the kernel is fp32-VALU/latency bound (DESIGN.md)"},
            "kernel_only_env_steps_per_sec_per_gpu": E / (kstep_ms * 1e-3),
        }
        if valu is not None:
            out["roofline"]["valu"] = valu
        if lazy is not None:
            out["lazy_reset"] = lazy
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if env is not None:
        env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
