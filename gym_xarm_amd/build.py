"""In-tree build of libxarm_hip.so (hipcc, gfx950 only)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libxarm_hip.so")
# one translation unit per kernel family (the fused kernels take ~30 s each to compile; the units are compiled in parallel)
UNITS = ["xarm_hip.hip", "xarm_k_pnp.hip", "xarm_k_pnp_coop.hip", "xarm_k_reach.hip", "xarm_k_handover.hip", "xarm_k_handover_coop.hip",
         "xarm_k_handover2.hip", "xarm_k_stack.hip"]
HEADERS = ["xarm_dev.h", "xarm_core.h", "xarm7_pd_model.h", "xarm_reach_core.h", "xarm7_reach_model.h", "xarm_handover_core.h", "xarm_handover2_core.h",
           "xarm_stack_core.h", "xarm_coop_core.h", "xarm_reach_coop_core.h", "xarm_handover_coop_core.h"]
SOURCES = UNITS + HEADERS
# per-unit extra flags.  Tried and not shipped: "-ffp-contract=on" for xarm_k_pnp.hip / xarm_k_handover.hip (fused multiply-adds
# only where the source writes them, so that k_step_fast / k_ho_step_fast compute the BITS of k_step / k_ho_step on every env
# they accept, as the host build does) - it holds, and costs 2.6 % of the headline (k_step_fast + hand-off 1.35 -> 1.43 ms,
# k_step 2.02 -> 2.22 ms) and 6 % of Handover: the default `fast` contraction stays, the families agree to float32 rounding.
# Shipped: "-ffp-contract=on" for the cooperative Handover unit.  Its sweeps already spell every fused multiply-add out; what was
# left to the compiler was the glue of xhc::substep (joint update, object impulse, integration), and hipcc fused it differently
# after an unrelated, bit-exact change to the shared sweep (round 4: the commit-free pair step - old and new source are bit for bit
# equal on the host and, built with this flag, on the device; in the default build 45 % of the contact envs differed after one
# step).  With the flag the unit's bits are a function of its source alone.
UNIT_FLAGS = {"xarm_k_handover_coop.hip": ["-ffp-contract=on"]}
# -fno-slp-vectorize: LLVM's SLP pass pairs the scalar fp32 ops of the unrolled solver into v_pk_* instructions,
# which need even-aligned register pairs; in this 400-live-value kernel that costs ~30 % extra v_mov and pushes
# 1.3 KB/lane into scratch.  Without it the step kernel needs 28 B/lane of scratch and 18 % fewer instructions.
# -amdgpu-sched-strategy=iterative-ilp: the kernels run one wavefront per SIMD, so nothing hides the latency between
# dependent instructions except the order of the wavefront's own instruction stream; the ILP-driven scheduler takes
# the contact-path tick from 2.36 to 2.05 ms (tools/variant_time.sh; disabling post-RA scheduling costs +33 %).  Its
# price: it raises register pressure until ~42 dwords of per-env state are parked in scratch across each substep's
# sweep (168 B/lane, outside the Gauss-Seidel loop; +19 MB of L2<->fabric traffic per 65 536-env launch).
# XARM_SCHED=default builds with LLVM's default scheduler instead (0 B scratch, ~13 % slower).  Re-checked in round 4 for the
# lane-pair kernels of StackTower / two-stick Handover (k_st_step 3.95 ms, k_ho2_step 4.55 ms with iterative-ilp): default 4.30 / 5.38,
# max-ilp 4.34 / 5.41, max-memory-clause 4.38 / 5.36, iterative-minreg 4.98 / 5.74 ms - although every one of them emits ~100 fewer
# VALU instructions per sweep: what they add is s_waitcnt stalls (147 -> 243-302 per sweep), and a lone wavefront has nothing to hide them.
_SCHED = [] if os.environ.get("XARM_SCHED", "ilp") == "default" else ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize"] + _SCHED

def find_hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm); set HIPCC=/path/to/hipcc")


def stale():
    if not os.path.exists(LIB):
        return True
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(os.path.dirname(HERE), "include", "xarm_hip.h")]
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(args):
    cmd, verbose = args
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=True, extra_flags=(), lib=LIB, jobs=None):
    """Regenerate the model header, compile csrc/xarm_*.hip (in parallel) and link csrc/libxarm_hip.so."""
    from concurrent.futures import ThreadPoolExecutor
    gen = os.path.join(os.path.dirname(HERE), "tools", "gen_model_header.py")
    if os.path.exists(gen):
        subprocess.check_call([sys.executable, gen], stdout=subprocess.DEVNULL)
    if not force and lib == LIB and not stale():
        return lib
    hipcc = find_hipcc()
    objdir = os.path.join(CSRC, "build" if lib == LIB else "build_" + os.path.splitext(os.path.basename(lib))[0])
    os.makedirs(objdir, exist_ok=True)
    objs, jobs_l = [], []
    for u in UNITS:
        o = os.path.join(objdir, os.path.splitext(u)[0] + ".o")
        objs.append(o)
        jobs_l.append(([hipcc] + HIPCC_FLAGS + UNIT_FLAGS.get(u, []) + list(extra_flags) + ["-c", "-o", o, os.path.join(CSRC, u)], verbose))
    with ThreadPoolExecutor(jobs or min(len(UNITS), os.cpu_count() or 1)) as ex:
        list(ex.map(_compile, jobs_l))
    _compile(([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, verbose))
    return lib


def build_example(verbose=True):
    """examples/abi_step_loop: a plain-C host program on include/xarm_hip.h (gcc, no torch, no Python)."""
    root = os.path.dirname(HERE)
    src, out = os.path.join(root, "examples", "abi_step_loop.c"), os.path.join(root, "examples", "abi_step_loop")
    if not os.path.exists(src):
        return None
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(os.path.join(root, "include", "xarm_hip.h"))):
        return out
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-O2", "-std=c11", "-I", os.path.join(root, "include"), "-I", os.path.join(rocm, "include"), src,
           "-L", CSRC, "-lxarm_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../gym_xarm_amd/csrc", "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
