#!/usr/bin/env python3
"""Headline benchmark: env steps/sec of XarmPDPickAndPlace-v0 (BASELINE.json `metric`).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu E] [--workload pnp|reach|handover|stack|handover2]
                  [--scaling weak|strong] [--repeats R] [--episode-phase desync|lockstep] [--aged-preroll P]
                  [--no-aged] [--no-lockstep] [--no-lazy] [--no-strong] [--no-extras] [--no-cpu-baseline] [--allow-variant-lib]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one env step of EVERY environment of the job (one xarm_step call per rank), including
the automatic resets of finished episodes.  `--scaling weak` (default, what `value` uses): each rank owns
`--envs-per-gpu` (default 65 536, the configuration the metric is quoted on) independent environments, global env
ids rank*E .. (rank+1)*E-1; `--scaling strong`: the same number of environments in TOTAL, split over the ranks
(gym_xarm_amd.distributed.shard_range).  No data-path collective either way.  Inputs are
synthetic: a ring of 64 pre-generated uniform[-1,1] action tensors resident in HBM, so the timed region contains no
RNG and no host->device traffic.  The timed window of K steps is repeated `--repeats` times back to back (default 3);
`value` is the MEDIAN window (max over ranks per window), the others are listed.  `--episode-phase desync` spreads
the per-env step counters uniformly over the episode length before the warm-up, so that time-limit resets arrive at
their steady-state rate (E / max_episode_steps per step) instead of all E at once every 50th step; `lockstep` keeps the
counters as reset() leaves them; the default `auto` picks the workload's own steady state - desync where an episode can
end early by success (pnp, handover), lockstep where every episode has the same fixed length (reach, stack).
Rank 0 prints ONE JSON line.  `library` in it is xarm_version() and the path of the loaded libxarm_hip.so; a TIMING VARIANT
(a development build with fewer solver sweeps) or a library named by XARM_HIP_LIB makes bench.py exit non-zero unless
--allow-variant-lib says the run is a development measurement.

Besides `value` (the K timed steps right after the W warm-up steps) the line carries, each measured by the same
three-window protocol and never feeding `value`:
  aged_state      the same handle after an untimed pre-roll of --aged-preroll (default 1 000) further steps: random actions
                  knock the objects about, reset wavefronts then carry finger-contact rows through their ticks - the rate a
                  long-running user sees (skip: --no-aged);
  strong_scaling  only when world > 1 under weak scaling: the configured env count (65 536) as the TOTAL, split over the
                  ranks by shard_range - the other reading of "65 536 parallel envs at 1/2/4/8 GPUs" (skip: --no-strong);
  lockstep_phase  episodes in lockstep from reset() (skip: --no-lockstep); lazy_reset: the opt-in lazy auto-reset (--no-lazy).
--no-extras skips all four (what the profiling scripts use).

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
#              oracle class, CPU-baseline sample (envs in total - BASELINE.md 3 asks for 4 096 - and steps), config dict)
WORKLOADS = {
    "pnp": ("XarmPDPickAndPlace-v0", 65536, 4, ALGO_BYTES_PER_ENV_STEP, "k_step", "OraclePnP", (4096, 150),
            dict(GUI=False, num_obj=1, reward_type="sparse", init_grasp_rate=0.0, goal_ground_rate=0.0, goal_shape="air")),
    "reach": ("XarmReach-v0", 4096, 4, 336, "k_reach_step", "OracleReach", (4096, 100), None),
    "handover": ("XarmPDHandover-v0", 16384, 8, 648, "k_ho_step", "OracleHandover", (4096, 60),
                 dict(GUI=False, num_obj=1, same_side_rate=0.5, goal_shape="ground", use_stand=False)),
    "stack": ("XarmPDStackTower-v0", 8192, 8, 1040, "k_st_step", "OracleStackTower", (4096, 60), None),
    # not a BASELINE config: the reference's own test.py configuration (num_obj 2, goal_shape 'any'), same per-GPU size as
    # config 5; algorithmic bytes by SURVEY 8(d)'s rule: state (36 + 26 + 6 + 4 = 72 f) x 2 + action 32 B + out (42 + 6 + 6 + 3 = 57 f)
    "handover2": ("XarmHandover-v0", 16384, 8, 836, "k_ho2_step", "OracleHandover", (4096, 30),
                  dict(GUI=False, num_obj=2, same_side_rate=0.5, goal_shape="any", use_stand=False)),
}


COOP_STEP_KERNEL = {"pnp": "k_step_coop", "reach": "k_reach_step_coop", "handover": "k_ho_step_coop_list"}   # batches <= xarm_config.step_coop_limit

WORKLOAD_NAMES = {
    "pnp": "XarmPDPickAndPlace-v0 (XarmPickAndPlace, num_obj=1, sparse reward, goal_shape=air)",
    "reach": "XarmReach-v0 (XarmReachEnv, sparse reward; BASELINE config 2)",
    "handover": "XarmPDHandover-v0 (XarmHandover, num_obj=1, goal_shape=ground, same_side_rate=0.5; BASELINE config 5)",
    "stack": "XarmPDStackTower-v0 (XarmStackTowerEnv, three cubes, sparse reward; BASELINE config 4)",
    "handover2": "XarmHandover-v0 (XarmHandover, num_obj=2, goal_shape=any, same_side_rate=0.5: the reference's test.py:9-15; not a BASELINE config)",
}
SUBSTEPS = {"pnp": 15, "reach": 20, "handover": 15, "stack": 15, "handover2": 15}   # internal substeps (Handover: 15 ticks of one substep);
#                                                                                   the JSON line reports the library's own figure (xarm_dims)


def reference_availability():
    """the reference's arithmetic lives in PyBullet; report whether this box could run it side by side"""
    try:
        import pybullet  # noqa: F401
    except Exception as e:   # ModuleNotFoundError on every box of this project so far
        return "unavailable (pybullet not importable: %s)" % type(e).__name__
    return "pybullet importable - side-by-side harness: tests/tools/pybullet_harness.py (not run by bench.py)"


def cpu_baseline(workload="pnp"):
    """The CPU oracle (a restatement = kind "port"; PyBullet itself is absent) on the host cores.  BASELINE.md 3 sketches
    "OpenMP over 4 096 envs"; what runs here is the same thing by other means: the 4 096 envs are split over `cores` Python
    threads, each stepping its own oracle instance through ctypes (the GIL is released inside the C call; the oracle itself is
    single-threaded C, gcc -O2, float64)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.lib()
    env_id, _, act_dim, _, _, cls, (sample_envs, steps), _ = WORKLOADS[workload]
    cores = max(1, min(os.cpu_count() or 1, 16))
    per = max(1, sample_envs // cores)
    okw = {"num_obj": 2, "goal_shape": "any"} if workload == "handover2" else {}
    envs = [getattr(O, cls)(per, seed=0, env_id_offset=k * per, **okw) for k in range(cores)]
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(steps, cores, per, act_dim))
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda k: envs[k].reset(), range(cores)))
        wall_reset = time.perf_counter() - t0
        t0 = time.perf_counter()
        list(ex.map(lambda k: [envs[k].step(acts[s, k]) for s in range(steps)], range(cores)))
        wall_steps = time.perf_counter() - t0
    n = cores * per * steps
    return {"value": n / wall_steps, "unit": "env steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps of %s on the CPU oracle (single-threaded C, float64, gcc -O2) split over %d Python threads through "
                      "ctypes - not OpenMP; %.1f s of stepping + %.1f s of reset()" % (cores * per, steps, env_id, cores, wall_steps, wall_reset),
            "reference": reference_availability()}


def timed_window(env, ring, first, steps, world, dev, dist, D, torch, sync):
    """time exactly `steps` calls of env.step between two barriers + synchronize; returns max-over-ranks seconds,
    episodes finished, and the HIP-event averages of the step kernel and of the reset kernels (ms per call)"""
    env.timing_enable(True)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    n_done = torch.zeros((), device=dev)
    for i in range(steps):
        obs, rew, done, info = env.step(ring[(first + i) % 64])
        n_done += done.sum()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = D.max_over_ranks(time.perf_counter() - t0, device=dev)
    kstep_ms_total, launches = env.timing_read()
    reset_ms_total, _ = env.timing_read_reset()
    kstep_ms = D.max_over_ranks(kstep_ms_total / max(launches, 1), device=dev)
    reset_ms = D.max_over_ranks(reset_ms_total / max(launches, 1), device=dev)
    resets = D.sum_over_ranks(float(n_done.item()), device=dev)
    return dt, resets, kstep_ms, reset_ms, int(launches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="pnp")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--episode-phase", choices=["auto", "desync", "lockstep"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lazy", action="store_true", help="skip the extra measurement of the opt-in lazy auto-reset mode")
    ap.add_argument("--no-lockstep", action="store_true", help="skip the lockstep-phase leg")
    ap.add_argument("--no-aged", action="store_true", help="skip the aged-state leg (pre-roll + the same windows)")
    ap.add_argument("--aged-preroll", type=int, default=1000, help="untimed steps before the aged-state windows")
    ap.add_argument("--no-strong", action="store_true", help="world > 1: skip the strong-scaling leg")
    ap.add_argument("--no-extras", action="store_true", help="skip every leg that does not feed `value`")
    ap.add_argument("--allow-variant-lib", action="store_true", help="development measurement: accept a TIMING VARIANT build / XARM_HIP_LIB")
    args = ap.parse_args()
    if args.no_extras:
        args.no_lazy = args.no_lockstep = args.no_aged = args.no_strong = True

    import torch
    import torch.distributed as dist
    import gym_xarm_amd
    from gym_xarm_amd import distributed as D

    rank, local_rank, world = D.env_from_torchrun()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    # rehearsal hooks (the driver's real runs use none of them): XARM_BENCH_DEVICE pins every rank to one device of a
    # 1-GPU box, or to "cpu" together with XARM_BENCH_ENV_FACTORY=module:function (a stand-in env: the CPU test of this
    # file's rank plumbing, tests/test_distributed.py); XARM_BENCH_BACKEND=gloo replaces RCCL (two ranks cannot share a
    # device under RCCL)
    dev_spec = os.environ.get("XARM_BENCH_DEVICE", str(local_rank))
    backend = os.environ.get("XARM_BENCH_BACKEND", "nccl")
    on_cpu = dev_spec == "cpu"
    if on_cpu:
        dev = torch.device("cpu")
        if not os.environ.get("XARM_BENCH_ENV_FACTORY"):
            raise SystemExit("XARM_BENCH_DEVICE=cpu needs XARM_BENCH_ENV_FACTORY (the product has no CPU path)")
    else:
        torch.cuda.set_device(int(dev_spec))
        dev = torch.device("cuda", int(dev_spec))
    sync = (lambda: None) if on_cpu else torch.cuda.synchronize
    make = gym_xarm_amd.make
    if os.environ.get("XARM_BENCH_ENV_FACTORY"):
        import importlib
        mod, _, fn = os.environ["XARM_BENCH_ENV_FACTORY"].partition(":")
        make = getattr(importlib.import_module(mod), fn)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and not on_cpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    env_id, default_E, act_dim, algo_bytes_per_step, kernel_name, _, _, env_config = WORKLOADS[args.workload]
    n_cfg = args.envs_per_gpu or default_E
    if args.scaling == "strong":
        lo, hi = D.shard_range(n_cfg, rank, world)     # the configured env count is the TOTAL, split over the ranks
        E, offset = hi - lo, lo
    else:
        E, offset = n_cfg, rank * n_cfg
    env = make(env_id, num_envs=E, seed=0, env_id_offset=offset, device=dev, config=env_config)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    ring = [torch.rand(E, act_dim, device=dev, generator=gen) * 2 - 1 for _ in range(64)]
    env.reset()
    T_ep = env.max_episode_steps
    env_state_dim = getattr(env, "state_dim", 0)
    env_dims = type("Dims", (), {"n_substeps": getattr(env, "n_substeps", None)})
    env_out_floats = getattr(env, "obs_dim", 0) + 2 * getattr(env, "goal_dim", 0)
    # which kernel family this handle runs (include/xarm_hip.h xarm_kernel_limits): the cooperative kernels serve batches
    # / reset lists up to the limits, the one-env-per-lane kernels the rest
    reset_limit, step_limit = env.kernel_limits() if hasattr(env, "kernel_limits") else (0, 0)
    # ... and which launches a step call is made of, from the handle itself (xarm_pipeline_info), with the library's identity
    pipe = env.pipeline_info() if hasattr(env, "pipeline_info") else dict(fast_pipeline=False, reset_overlap=False, eject_coop_cap=0, solver_iterations=None)
    lib_version, lib_path = env.library() if hasattr(env, "library") else ("stand-in env (no library)", None)
    variant = "TIMING VARIANT" in lib_version or bool(os.environ.get("XARM_HIP_LIB"))
    if variant and not args.allow_variant_lib:
        raise SystemExit("bench.py: the loaded library is not the product build (%s, %s); pass --allow-variant-lib for a development measurement"
                         % (lib_version, lib_path))
    if E <= step_limit and args.workload in COOP_STEP_KERNEL:
        kernel_name = COOP_STEP_KERNEL[args.workload]
    elif pipe["fast_pipeline"]:
        # the pad-free fast kernel + the hand-off of the envs with finger-pad rows to the cooperative kernel;
        # the HIP events of the "step kernel" bracket both launches
        kernel_name = {"pnp": "k_step_fast", "handover": "k_ho_step_fast"}[args.workload]
    handoff_kernel = {"k_step_fast": "k_step_coop_list", "k_ho_step_fast": "k_ho_step_coop_list"}.get(kernel_name)
    if args.episode_phase == "auto":
        # steady state of the workload: envs whose episodes can end early (success: PickAndPlace, Handover) drift apart
        # and reset at a uniform rate; fixed-length episodes (Reach: 25 steps, StackTower: 50, never `done` before)
        # start together and stay together for ever - one bulk reset every T_ep-th step IS their steady state
        args.episode_phase = "desync" if args.workload in ("pnp", "handover", "handover2") else "lockstep"
    if args.episode_phase == "desync":
        # steady state: episode phases uniform over the episode length, keyed by the global env id
        env.set_episode_steps((torch.arange(E, device=dev) + offset) * 7919 % T_ep)
    n_done = torch.zeros((), device=dev)
    for i in range(args.warmup):
        # the warm-up runs exactly what the timed loop runs (torch loads the code object of a kernel at its first
        # launch: round 1 timed the first `done.sum()` inside the window, 50-80 ms of module loading)
        obs, rew, done, info = env.step(ring[i % 64])
        n_done += done.sum()
    sync()
    def measure(e, first):
        """`--repeats` back-to-back windows of --steps calls; returns (all windows, the median one)"""
        ws = [timed_window(e, ring_of[id(e)], first + r * args.steps, args.steps, world, dev, dist, D, torch, sync) for r in range(max(1, args.repeats))]
        order = sorted(range(len(ws)), key=lambda k: ws[k][0])
        return ws, ws[order[len(order) // 2]]

    ring_of = {id(env): ring}
    windows, (dt, resets, kstep_ms, reset_ms, launches) = measure(env, args.warmup)
    if pipe["fast_pipeline"] and hasattr(env, "debug_counts"):
        # (outside the timed windows: the read synchronises) how many envs the fast kernel handed to the cooperative kernel(s) in the last call
        pipe = dict(pipe, handed_off_last_call=int(env.debug_counts()[1]))
    if hasattr(env, "stage_info") and args.workload in ("handover", "pnp"):
        pipe = dict(pipe, stage_ticks=env.stage_info())
        if args.workload == "pnp" and pipe["fast_pipeline"] and len(pipe["stage_ticks"]) > 2:
            kernel_name, handoff_kernel = "k_step_fast_stage", "k_step_coop_list_stage"      # the staged step's own kernels (xarm_k_pnp.hip)
    total_envs = int(D.sum_over_ranks(E, device=dev))

    # extra: the same handle, aged.  An untimed pre-roll, then the same windows.  Random actions knock the objects about,
    # so the slowest reset wavefront of a call carries finger-contact rows through its six ticks.
    aged = None
    if not args.no_aged and args.aged_preroll > 0:
        for i in range(args.aged_preroll):
            env.step(ring[i % 64])
        sync()
        aw, (adt, aresets, akstep, areset, _) = measure(env, args.aged_preroll)
        aged = {"value": total_envs * args.steps / adt, "unit": "env steps/s", "preroll_steps": args.aged_preroll + args.warmup + args.steps * len(windows),
                "ms_per_step": adt / args.steps * 1e3, "step_kernel_ms": akstep, "reset_kernels_ms": areset,
                "resets_per_step": aresets / args.steps,
                "env_steps_per_sec": [total_envs * args.steps / w[0] for w in aw],
                "note": "same handle and protocol as `value`, after the untimed pre-roll: the rate of a long run"}

    # extra (world > 1, weak scaling): the configured env count as the TOTAL of the job, split over the ranks
    strong = None
    if world > 1 and args.scaling == "weak" and not args.no_strong:
        slo, shi = D.shard_range(n_cfg, rank, world)
        senv = make(env_id, num_envs=shi - slo, seed=0, env_id_offset=slo, device=dev, config=env_config)
        sring = [r[:shi - slo] for r in ring]
        ring_of[id(senv)] = sring
        senv.reset()
        if args.episode_phase == "desync":
            senv.set_episode_steps((torch.arange(shi - slo, device=dev) + slo) * 7919 % T_ep)
        for i in range(args.warmup):
            senv.step(sring[i % 64])
        sync()
        sw, (sdt, sres, skstep, sreset, _) = measure(senv, args.warmup)
        s_lim = senv.kernel_limits() if hasattr(senv, "kernel_limits") else (0, 0)
        strong = {"value": n_cfg * args.steps / sdt, "unit": "env steps/s", "total_envs": n_cfg, "envs_per_gpu": shi - slo,
                  "shard_range_rank0": [int(slo), int(shi)], "ms_per_step": sdt / args.steps * 1e3,
                  "step_kernel_ms": skstep, "reset_kernels_ms": sreset,
                  "step_kernel": COOP_STEP_KERNEL[args.workload] if args.workload in COOP_STEP_KERNEL and (shi - slo) <= s_lim[1] else WORKLOADS[args.workload][4],
                  "env_steps_per_sec": [n_cfg * args.steps / w[0] for w in sw],
                  "note": "strong scaling: %d envs in TOTAL split by shard_range; a step call is latency-bound, so this grows "
                          "far slower than the weak-scaling `value` (DESIGN.md 7)" % n_cfg}
        senv.close()

    # extra (workloads measured with desynchronised phases): the same env started in lockstep - every episode from
    # reset() together, the time-limit resets arriving as one bulk reset every T_ep-th step, which is what a caller sees
    # until early successes spread the phases.  Never feeds `value`; timed over whole episodes (>= 2 * T_ep steps) so
    # that the window holds its share of bulk resets whatever --steps is.
    lockstep = None
    if args.episode_phase == "desync" and not args.no_lockstep and hasattr(env, "set_episode_steps"):
        env.close()
        kenv = make(env_id, num_envs=E, seed=0, env_id_offset=offset, device=dev, config=env_config)
        kenv.reset()
        for i in range(args.warmup):
            kenv.step(ring[i % 64])
        n_lock = 2 * T_ep
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for i in range(n_lock):
            kenv.step(ring[(args.warmup + i) % 64])
        sync()
        if world > 1:
            dist.barrier()
        sync()
        kdt = D.max_over_ranks(time.perf_counter() - t0, device=dev)
        lockstep = {"value": int(D.sum_over_ranks(E, device=dev)) * n_lock / kdt, "unit": "env steps/s", "steps": n_lock,
                    "ms_per_step": kdt / n_lock * 1e3,
                    "note": "episodes in lockstep from reset(): bulk resets every %d-th step instead of E/%d per step; "
                            "`value` is the desynchronised steady state, the slower of the two" % (T_ep, T_ep)}
        kenv.close()
        env = None

    # extra (pnp only): the opt-in lazy auto-reset mode (include/xarm_hip.h XARM_AUTO_RESET_LAZY) - a different contract
    # from the reference's VecEnv, so it never feeds `value`; useful = env steps that are not reset ticks
    lazy = None
    if args.workload == "pnp" and not args.no_lazy:
        if env is not None:
            env.close()
        lenv = make(env_id, num_envs=E, seed=0, env_id_offset=offset, device=dev, config=env_config, auto_reset="lazy")
        lenv.reset()
        useful = torch.zeros((), device=dev)
        for i in range(args.warmup):
            _, _, _, linfo = lenv.step(ring[i % 64])
            useful += (~linfo["resetting"]).sum()
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        useful = torch.zeros((), device=dev)
        for i in range(args.steps):
            _, _, _, linfo = lenv.step(ring[(args.warmup + i) % 64])
            useful += (~linfo["resetting"]).sum()
        sync()
        if world > 1:
            dist.barrier()
        sync()
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
        algo_bytes = algo_bytes_per_step * E                     # one xarm_step call processes E env steps per rank
        call_ms = kstep_ms + reset_ms                            # the kernels of one xarm_step call
        achieved = algo_bytes / (call_ms * 1e-3) / 1e9
        step_only = algo_bytes / (kstep_ms * 1e-3) / 1e9
        traffic = None
        pmc_data = {}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # written from the rocprofv3 --pmc passes
        if os.path.exists(pmc):
            try:
                pmc_data = json.load(open(pmc))
            except Exception:
                pmc_data = {}
        traffic = pmc_data.get("%s_hbm_bytes_per_launch_%d" % (kernel_name, E))
        # the staged step (Handover, PickAndPlace) launches its fast kernel once per stage: the PMC figure is the average of ONE launch
        step_launches = max(1, len(pipe.get("stage_ticks", [0, 0])) - 1)
        traffic_one_launch = traffic
        if traffic and step_launches > 1:
            traffic = traffic * step_launches
        # secondary ceiling (the one that actually binds): fp32 vector issue.  FLOPs per env step are COUNTED (the
        # kernel core instantiated with a counting scalar type, tools/count_flops.py -> profiles/flop_count.json);
        # the wave-instruction count of the committed SQ_INSTS_VALU pass is kept beside it
        # secondary ceiling (the one that actually binds): fp32 vector issue.  FLOPs are COUNTED (the kernel core
        # instantiated with a counting scalar type, tools/count_flops.py -> profiles/flop_count.json): per env step for the
        # step kernel and, where counted (pnp), per reset for the reset kernels - both over the HIP-event time of the call
        valu = None
        flops = pmc_data.get("%s_counted_flops_per_env_step" % args.workload)
        flops_reset = pmc_data.get("%s_counted_flops_per_env_reset" % args.workload)
        n_valu = pmc_data.get("%s_valu_wave_insts_per_launch_%d" % (kernel_name, E))
        if n_valu and step_launches > 1:
            n_valu = n_valu * step_launches      # (per call: the PMC figure is the average of one of the stage launches)
        resets_per_call = resets / max(args.steps, 1) / world
        if flops:
            tf_step = flops * E / (kstep_ms * 1e-3) / 1e12
            valu = {"unit": "TFLOP/s (counted flops / HIP-event kernel time)", "peak": VALU_PEAK_TFLOPS,
                    "step_kernel": {"achieved": tf_step, "frac": tf_step / VALU_PEAK_TFLOPS, "counted_flops_per_env_step": flops}}
            if flops_reset and reset_ms > 0:
                tf_reset = flops_reset * resets_per_call / (reset_ms * 1e-3) / 1e12
                tf_call = (flops * E + flops_reset * resets_per_call) / (call_ms * 1e-3) / 1e12
                valu["reset_kernels"] = {"achieved": tf_reset, "frac": tf_reset / VALU_PEAK_TFLOPS, "counted_flops_per_env_reset": flops_reset,
                                         "resets_per_call": resets_per_call}
                if pipe["reset_overlap"]:
                    # the reset bracket is a residual (first launch on the side stream): no rate of the reset kernels alone
                    valu["reset_kernels"] = {"counted_flops_per_env_reset": flops_reset, "resets_per_call": resets_per_call,
                                             "note": "no rate: the reset time of a pipelined call is the residual after the hand-off"}
                valu["achieved"], valu["frac"], valu["covers"] = tf_call, tf_call / VALU_PEAK_TFLOPS, "step + reset kernels of one xarm_step call"
            else:
                valu["achieved"], valu["frac"], valu["covers"] = tf_step, tf_step / VALU_PEAK_TFLOPS, "step kernel only (reset flops not counted for this workload)"
            if n_valu:
                valu["wave_insts_per_step_kernel_launch"] = n_valu
        elif n_valu:
            tf = n_valu * 64 * 2 / (kstep_ms * 1e-3) / 1e12
            valu = {"achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s (<=2 flop per lane-instruction, step kernel only)",
                    "frac": tf / VALU_PEAK_TFLOPS, "wave_insts_per_launch": n_valu, "covers": "step kernel only"}
        ho_reset = "k_ho_reset_coop" if reset_limit > 0 else "k_ho_reset"
        reset_kernel_key = {"pnp": "k_reset_coop", "reach": "k_reach_reset_coop" if E <= reset_limit else "k_reach_reset",
                            "handover": ho_reset, "stack": "k_st_reset", "handover2": "k_ho2_reset"}[args.workload]
        reset_kernel = {"pnp": "k_reset_coop (<= %d finished envs per call) / k_reset" % reset_limit,
                        "reach": "k_reach_reset_coop" if E <= reset_limit else "k_reach_reset",
                        "handover": ho_reset + (" (<= %d finished envs per call) / k_ho_reset" % reset_limit if reset_limit > 0 else ""),
                        "stack": "k_st_reset", "handover2": "k_ho2_reset"}[args.workload]
        # reset kernels: algorithmic bytes = state in + out and the fresh obs / goal rows, per finished env
        reset_algo = resets_per_call * (2 * 4 * env_state_dim + 4 * env_out_floats)
        reset_traffic = pmc_data.get("%s_hbm_bytes_per_launch_%d" % (reset_kernel_key, E))
        reset_entry = {"name": reset_kernel, "avg_ms": reset_ms, "share": reset_ms / call_ms, "resets_per_call": resets_per_call,
                       "algorithmic_bytes_per_launch": reset_algo, "traffic": reset_traffic}
        reset_entry["avg_ms_is"] = "reset kernels, on the caller's stream after the step kernel(s)"
        if pipe["reset_overlap"] and reset_traffic:
            reset_entry["launches_per_call"] = 2          # `traffic` is the PMC average of ONE launch that did work
            reset_entry["traffic_per_call"] = 2 * reset_traffic
        if pipe["reset_overlap"]:
            # the first reset launch runs on the handle's side stream beside the hand-off: the bracket holds what is LEFT of the resets
            # after the hand-off, so step + reset no longer partitions the kernels' own time and no rate is derived from it
            reset_entry["avg_ms_is"] = "residual after the hand-off (the first reset launch overlaps it on a side stream; the join is inside the bracket)"
            reset_entry["note"] = ("two launches per call: the episodes that ended in k_step_fast are reset on a side stream while the hand-off "
                                   "runs, those that ended in the hand-off after it (DESIGN.md 4b)")
        elif reset_ms > 0:
            reset_entry["achieved_GBs"] = reset_algo / (reset_ms * 1e-3) / 1e9
            reset_entry["frac"] = reset_entry["achieved_GBs"] / HBM_PEAK_GBS
        out = {
            "metric": "env steps/sec (whole node), %s" % env_id, "value": value, "unit": "env steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOAD_NAMES[args.workload],
                       "envs_per_gpu": E, "total_envs": total_envs, "substeps_per_step": getattr(env_dims, "n_substeps", None) or SUBSTEPS[args.workload],
                       "solver_iterations": pipe["solver_iterations"],
                       "auto_reset": True, "episode_phase": args.episode_phase, "episodes_reset_in_window": int(resets),
                       "resets_per_step": resets / args.steps, "steady_state_time_limit_resets_per_step": total_envs / T_ep,
                       "parallelism": "env-shard x%d, no collective" % world},
            "repeats": {"n": len(windows), "value_is": "median window",
                        "env_steps_per_sec": [total_envs * args.steps / w[0] for w in windows],
                        "min": total_envs * args.steps / max(w[0] for w in windows),
                        "max": total_envs * args.steps / min(w[0] for w in windows)},
            # the unit of work is one xarm_step call = step kernel + the reset kernels that follow it; SURVEY 8(d)'s
            # algorithmic bytes per env step x E env steps per call, over the HIP-event time of those kernels
            # `bound` names what binds (fp32 VALU issue / dependent-instruction latency of lone wavefronts);
            # achieved / peak / frac / traffic are the HBM figures the contract asks for - a fused step touches HBM once
            "roofline": {"bound": "valu/latency", "reported_against": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "%s + %s (one xarm_step call)" % (kernel_name + (" + %s (hand-off)" % handoff_kernel if handoff_kernel else
                                                                                   " (+ k_class_hist, k_class_place: class order)" if kernel_name == "k_st_step" else ""), reset_kernel),
                         "kernel_avg_ms": call_ms, "kernel_launches": int(launches),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernels": {kernel_name: {"avg_ms": kstep_ms, "share": kstep_ms / call_ms,
                                                   "achieved_GBs": step_only, "frac": step_only / HBM_PEAK_GBS, "traffic": traffic},
                                     "reset": reset_entry},
                         "dominant_kernel": kernel_name if kstep_ms >= reset_ms else reset_kernel,
                         "note": "fused step: HBM is touched once per env step, the kernels are fp32-VALU/latency bound (DESIGN.md 5); "
                                 "`traffic` is the PMC figure of the step kernel, kernels.reset.traffic that of the reset kernel"},
            "kernel_only_env_steps_per_sec_per_gpu": E / (call_ms * 1e-3),
            "library": {"version": lib_version, "path": lib_path, "variant": variant, "pipeline": pipe,
                        "reset_coop_limit": reset_limit, "step_coop_limit": step_limit},
        }
        if step_launches > 1:
            out["roofline"]["kernels"][kernel_name].update(launches_per_call=step_launches, traffic_one_launch=traffic_one_launch,
                note="%d fast stages per call (ticks %s); `traffic` = %d x the PMC average of one launch, each stage reads and writes the state" % (
                    step_launches, pipe["stage_ticks"], step_launches))
        if valu is not None:
            out["roofline"]["valu"] = valu
        if aged is not None:
            out["aged_state"] = aged
        if strong is not None:
            out["strong_scaling"] = strong
        if lockstep is not None:
            out["lockstep_phase"] = lockstep
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
