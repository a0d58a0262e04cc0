// xarm_hip.hip - gfx950 kernels and the C ABI (include/xarm_hip.h) of the batched Xarm environments.
//
// PickAndPlace launches (cores: xarm_core.h, xarm_coop_core.h):
//   k_step        one thread per environment, one 64-lane wavefront per workgroup.  The fused step keeps an env's whole
//                 working set on chip for all 15 substeps x 50 solver sweeps: ~450 VGPRs of per-env state / solver blocks
//                 (hence 1 wave per SIMD, __launch_bounds__(64)) plus 149 floats per env of LDS (hand Jacobian S, T =
//                 M^-1 S^T, A_hh, table slots: lane-private columns, 38 KB per workgroup).  65 536 envs = 1024 workgroups =
//                 4 per CU; workgroups never communicate, so no XCD-aware remap is needed.
//   k_step_coop / k_reset_coop   one environment per DPP row of 16 lanes (4 per wavefront), impulse-space sweep spread
//                 over the row: the latency-optimal form for small batches and for the resets that follow a step.
//   k_reset       the one-env-per-lane reset, for bulk resets (> coop_limit finished envs in one call).
// HBM is touched once per env step: 54 state floats in, 54 out (structure-of-arrays, lane = env, every load/store a fully
// coalesced 256-B wave access), 4 action floats in and the 24+3+3+1 output floats + 2 flag bytes out (row-major at the
// API edge, 16-B vector stores).  Episodes that end are compacted into a list (one atomic per finished env) and
// re-initialised by the reset kernel inside the same xarm_step call.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "../../include/xarm_hip.h"
#include "xarm_core.h"
#include "xarm_reach_core.h"
#include "xarm_handover_core.h"
#include "xarm_handover2_core.h"
#include "xarm_stack_core.h"
#include "xarm_coop_core.h"
#include "xarm_handover_coop_core.h"
#include "xarm_reach_coop_core.h"

namespace {

constexpr int WG = 64;

struct DevLds {
    float *base;
    __device__ __forceinline__ float &operator[](int i) const { return base[i * WG]; }
};

struct KParams {
    float *state;      // [STATE_DIM][stride]
    int64_t stride;    // padded env count
    int64_t num_envs;
    xk::EnvCfg cfg;
    int auto_reset;
    int state_dim;
    int coop_limit;    // resets of at most this many envs run on the cooperative kernel (0: never)
    int eject_coop_cap; // fast-step pipeline: hand-offs of at most this many envs step on the cooperative kernel, more on k_step
    xr::EnvCfg rcfg;
    xh::EnvCfg hcfg;
};

// Output addresses are per-lane 64-bit values that LLVM would otherwise compute in the prologue and keep (spill)
// across the whole simulation; re-deriving the env index through an opaque move pins them to the epilogue.
__device__ __forceinline__ int64_t late_index(int64_t e) {
    int lo = (int)e, hi = (int)(e >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi) : : "memory");
    return ((int64_t)hi << 32) | (uint32_t)lo;
}

__device__ __forceinline__ void load_state(const KParams &P, int64_t e, xk::EnvState<float> &s) {
    const float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) { s.q[i] = S[(xk::S_Q + i) * n]; s.qd[i] = S[(xk::S_QD + i) * n]; }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        s.bp[i] = S[(xk::S_BP + i) * n]; s.bv[i] = S[(xk::S_BV + i) * n];
        s.bw[i] = S[(xk::S_BW + i) * n]; s.goal[i] = S[(xk::S_GOAL + i) * n];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) s.bq[i] = S[(xk::S_BQ + i) * n];
#pragma unroll
    for (int i = 0; i < 8; i++) { s.lam_t[i] = S[(xk::S_LT + i) * n]; s.lam_p[i] = S[(xk::S_LP + i) * n]; }
    s.touch = S[xk::S_TOUCH * n]; s.mug = S[xk::S_MUG * n]; s.steps = S[xk::S_STEPS * n]; s.episode = S[xk::S_EPISODE * n];
}
__device__ __forceinline__ void store_state(const KParams &P, int64_t e, const xk::EnvState<float> &s) {
    float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) { S[(xk::S_Q + i) * n] = s.q[i]; S[(xk::S_QD + i) * n] = s.qd[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        S[(xk::S_BP + i) * n] = s.bp[i]; S[(xk::S_BV + i) * n] = s.bv[i];
        S[(xk::S_BW + i) * n] = s.bw[i]; S[(xk::S_GOAL + i) * n] = s.goal[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) S[(xk::S_BQ + i) * n] = s.bq[i];
#pragma unroll
    for (int i = 0; i < 8; i++) { S[(xk::S_LT + i) * n] = s.lam_t[i]; S[(xk::S_LP + i) * n] = s.lam_p[i]; }
    S[xk::S_TOUCH * n] = s.touch; S[xk::S_MUG * n] = s.mug; S[xk::S_STEPS * n] = s.steps; S[xk::S_EPISODE * n] = s.episode;
}
__device__ __forceinline__ void write_obs(const float (&obs)[xk::OBS_DIM], const xk::EnvState<float> &s, int64_t e,
                                          float *obs_out, float *ag_out, float *dg_out) {
    float4 *o = reinterpret_cast<float4 *>(obs_out + e * xk::OBS_DIM);
#pragma unroll
    for (int k = 0; k < xk::OBS_DIM / 4; k++) o[k] = make_float4(obs[4 * k], obs[4 * k + 1], obs[4 * k + 2], obs[4 * k + 3]);
#pragma unroll
    for (int k = 0; k < 3; k++) { ag_out[e * 3 + k] = s.bp[k]; dg_out[e * 3 + k] = s.goal[k]; }
}

__global__ __launch_bounds__(WG) void k_init(KParams P) {
    const int64_t e = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e >= P.num_envs) return;
    xk::EnvState<float> s;
    xk::env_init<float>(P.cfg, e, s);
    store_state(P, e, s);
}

// XarmPickAndPlace.step for every env (list == null) or for the envs list[0 .. *count) (the hand-off of k_step_fast when it
// is too long for the cooperative kernel); finished episodes are appended to done_list
__global__ __launch_bounds__(WG) void k_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                             float *__restrict__ ag_out, float *__restrict__ dg_out,
                                             float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                             uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                             int *__restrict__ done_list, int *__restrict__ done_count,
                                             const int *__restrict__ list, const int *__restrict__ count) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t i_in = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (count && n <= P.eject_coop_cap) return;     // k_step_coop_list's range
    if (i_in >= n) return;
    const int64_t e_in = list ? (int64_t)list[i_in] : i_in;
    DevLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xk::OBS_DIM], reward;
    bool done, success;
    xk::env_step<float, DevLds>(P.cfg, s, act, obs, reward, done, success, lds);
    const int64_t e = late_index(e_in);
    store_state(P, e, s);
    write_obs(obs, s, e, obs_out, ag_out, dg_out);
    rew_out[e] = reward;
    done_out[e] = done ? 1 : 0;
    succ_out[e] = success ? 1 : 0;
    if (done && P.auto_reset) {
        if (term_obs) {
            float4 *o = reinterpret_cast<float4 *>(term_obs + e * xk::OBS_DIM);
#pragma unroll
            for (int k = 0; k < xk::OBS_DIM / 4; k++) o[k] = make_float4(obs[4 * k], obs[4 * k + 1], obs[4 * k + 2], obs[4 * k + 3]);
        }
        const int pos = atomicAdd(done_count, 1);
        done_list[pos] = (int)e;
    }
}

// The fast step: XarmPickAndPlace.step on the pad-free substep (xk::substep<.., FAST>) for every env.  An env none of
// whose finger pads comes within the solver margin of the object during the step - ~98 % of them - is finished here
// with the arithmetic of k_step (same bits in the host build / with -ffp-contract=on; float32 last bits apart otherwise).  An env with an active pad row stores NOTHING and is appended to eject_list: it is
// stepped again from its untouched state by k_step_coop_list (or k_step when the list is long).  Why: a wavefront with ONE
// such lane sweeps the pad blocks for all 64 lanes, and with ~2 % of the envs in contact that is most wavefronts - k_step
// takes 1.88 ms where a contact-free batch takes 0.74 ms (tools/fastpath_probe.py).  Only the table-slot columns live in
// LDS (8 KB per workgroup instead of 38 KB).
constexpr int FAST_LDS_FLOATS = xk::LDS_FLOATS - xk::LDS_TBL;
struct FastLds {
    float *base;
    __device__ __forceinline__ float &operator[](int i) const { return base[(i - xk::LDS_TBL) * WG]; }
};
__global__ __launch_bounds__(WG) void k_step_fast(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                  float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                  float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                  uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                  int *__restrict__ done_list, int *__restrict__ done_count,
                                                  int *__restrict__ eject_list, int *__restrict__ eject_count) {
    __shared__ float smem[FAST_LDS_FLOATS * WG];
    const int64_t e_in = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e_in >= P.num_envs) return;
    FastLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xk::OBS_DIM], reward;
    bool done, success;
    const bool ok = xk::env_step_fast<float, FastLds>(P.cfg, s, act, obs, reward, done, success, lds);
    const int64_t e = late_index(e_in);
    if (!ok) {
        const int pos = atomicAdd(eject_count, 1);
        eject_list[pos] = (int)e;
        return;
    }
    store_state(P, e, s);
    write_obs(obs, s, e, obs_out, ag_out, dg_out);
    rew_out[e] = reward;
    done_out[e] = done ? 1 : 0;
    succ_out[e] = success ? 1 : 0;
    if (done && P.auto_reset) {
        if (term_obs) {
            float4 *o = reinterpret_cast<float4 *>(term_obs + e * xk::OBS_DIM);
#pragma unroll
            for (int k = 0; k < xk::OBS_DIM / 4; k++) o[k] = make_float4(obs[4 * k], obs[4 * k + 1], obs[4 * k + 2], obs[4 * k + 3]);
        }
        const int pos = atomicAdd(done_count, 1);
        done_list[pos] = (int)e;
    }
}

// lazy auto-reset (xk::env_step_lazy): a finished env runs its reset ticks in its next six step calls; no reset launch,
// no done list.  done_out carries the phase: 0 ordinary step, 1 the episode ended in this call, 2 reset tick.
__global__ __launch_bounds__(WG) void k_step_lazy(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                  float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                  float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                  uint8_t *__restrict__ succ_out) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t e_in = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e_in >= P.num_envs) return;
    DevLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xk::OBS_DIM], reward;
    bool done, success;
    int phase;
    xk::env_step_lazy<float, DevLds>(P.cfg, e_in, s, act, obs, reward, done, success, phase, lds);
    const int64_t e = late_index(e_in);
    store_state(P, e, s);
    write_obs(obs, s, e, obs_out, ag_out, dg_out);
    rew_out[e] = reward;
    done_out[e] = (uint8_t)phase;
    succ_out[e] = success ? 1 : 0;
}

// XarmPickAndPlace.reset for the envs in list[0 .. *count): thread i handles env list[i].  Counts of at most
// P.coop_limit belong to k_reset_coop (launched beside this kernel; exactly one of the two does the work).
__global__ __launch_bounds__(WG) void k_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                              float *__restrict__ obs_out, float *__restrict__ ag_out,
                                              float *__restrict__ dg_out) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (n <= P.coop_limit) return;
    if (i >= n) return;
    const int64_t e_in = list ? (int64_t)list[i] : i;
    DevLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    xk::env_reset<float, DevLds>(P.cfg, e_in, s, lds);
    const int64_t e = late_index(e_in);
    store_state(P, e, s);
    if (obs_out) {
        float obs[xk::OBS_DIM];
        xk::get_obs(s, obs);
        write_obs(obs, s, e, obs_out, ag_out, dg_out);
    }
}

// The same reset with one environment per DPP row of 16 lanes (xarm_coop_core.h): 4 environments per wavefront, the
// Gauss-Seidel sweep spread over the row.  This is the latency-optimal form for the usual case - a handful to a few
// thousand finished episodes per step - where k_reset would keep one wavefront busy for six sequential ticks while
// the rest of the GPU idles.  Rows beyond the list shadow its last entry (the wavefront stays convergent) and store
// nothing; lane 0 of a row writes the environment back.
constexpr int COOP_ENVS = WG / xc::GL;
// A cooperative workgroup owns 4 consecutive envs = 16 B of every state column, an HBM line holds 16-32 envs.  Under the
// default round-robin of workgroups over the 8 XCDs (one private L2 each) every XCD fetched - and partially wrote -
// every line: 5x the algorithmic bytes measured at 4 096 envs (profiles/r02e_reach_pmc_summary.json).  This bijective
// remap (valid for any grid size) gives the workgroups that share an XCD one contiguous env range instead.
__device__ __forceinline__ int64_t xcd_contiguous_block() {
    const unsigned b = blockIdx.x, nwg = gridDim.x, xcd = b & 7u, q = nwg >> 3, r = nwg & 7u;
    return (int64_t)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3));
}
__global__ __launch_bounds__(WG) void k_reset_coop(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                   float *__restrict__ obs_out, float *__restrict__ ag_out,
                                                   float *__restrict__ dg_out) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (n > P.coop_limit) return;
    // a reset list covers the front of a grid sized for coop_limit entries: the remap would pile its workgroups on one or
    // two XCDs, and the listed envs are scattered anyway - only the full-batch reset (no list) is remapped
    const int64_t i0 = (list ? (int64_t)blockIdx.x : xcd_contiguous_block()) * COOP_ENVS;
    if (i0 >= n) return;
    const int64_t i_raw = i0 + threadIdx.x / xc::GL;
    const bool live = i_raw < n;
    const int64_t i = live ? i_raw : n - 1;
    const int64_t e_in = list ? (int64_t)list[i] : i;
    const xc::Grp G{(int)(threadIdx.x & (xc::GL - 1))};
    DevLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    xc::env_reset<float, DevLds>(G, P.cfg, e_in, s, lds);
    if (!live || G.l != 0) return;
    const int64_t e = late_index(e_in);
    store_state(P, e, s);
    if (obs_out) {
        float obs[xk::OBS_DIM];
        xk::get_obs(s, obs);
        write_obs(obs, s, e, obs_out, ag_out, dg_out);
    }
}

// XarmPickAndPlace.step with one environment per 16-lane row (xarm_coop_core.h) - the launch for batches that leave
// most SIMDs without a wavefront under the one-env-per-lane mapping (num_envs <= kp.coop_step_limit): 16x the
// wavefronts and a ~3x shorter tick.  Same outputs and done list as k_step.
__global__ __launch_bounds__(WG) void k_step_coop(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                  float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                  float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                  uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                  int *__restrict__ done_list, int *__restrict__ done_count) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t e_raw = xcd_contiguous_block() * COOP_ENVS + threadIdx.x / xc::GL;
    const bool live = e_raw < P.num_envs;
    const int64_t e_in = live ? e_raw : P.num_envs - 1;
    const xc::Grp G{(int)(threadIdx.x & (xc::GL - 1))};
    DevLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xk::OBS_DIM], reward;
    bool done, success;
    xc::env_step<float, DevLds>(G, P.cfg, s, act, obs, reward, done, success, lds);
    if (!live || G.l != 0) return;
    const int64_t e = late_index(e_in);
    store_state(P, e, s);
    write_obs(obs, s, e, obs_out, ag_out, dg_out);
    rew_out[e] = reward;
    done_out[e] = done ? 1 : 0;
    succ_out[e] = success ? 1 : 0;
    if (done && P.auto_reset) {
        if (term_obs) {
            float4 *o = reinterpret_cast<float4 *>(term_obs + e * xk::OBS_DIM);
#pragma unroll
            for (int k = 0; k < xk::OBS_DIM / 4; k++) o[k] = make_float4(obs[4 * k], obs[4 * k + 1], obs[4 * k + 2], obs[4 * k + 3]);
        }
        const int pos = atomicAdd(done_count, 1);
        done_list[pos] = (int)e;
    }
}

// XarmPickAndPlace.step of the envs list[0 .. *count) on the cooperative core: the hand-off of k_step_fast (envs with an
// active finger-pad row).  The grid is fixed (the count lives on the device); a workgroup walks the list with a grid stride.
// Lists longer than P.eject_coop_cap belong to k_step (launched beside this kernel; exactly one of the two does the work).
__global__ __launch_bounds__(WG) void k_step_coop_list(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                       float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                       float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                       uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                       int *__restrict__ done_list, int *__restrict__ done_count,
                                                       const int *__restrict__ list, const int *__restrict__ count) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t n = (int64_t)*count;
    if (n > P.eject_coop_cap) return;
    const xc::Grp G{(int)(threadIdx.x & (xc::GL - 1))};
    DevLds lds{smem + threadIdx.x};
#pragma unroll 1
    for (int64_t i0 = (int64_t)blockIdx.x * COOP_ENVS; i0 < n; i0 += (int64_t)gridDim.x * COOP_ENVS) {
        const int64_t i_raw = i0 + threadIdx.x / xc::GL;
        const bool live = i_raw < n;
        const int64_t e_in = (int64_t)list[live ? i_raw : n - 1];
        xk::EnvState<float> s;
        load_state(P, e_in, s);
        const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in];
        const float act[4] = {a4.x, a4.y, a4.z, a4.w};
        float obs[xk::OBS_DIM], reward;
        bool done, success;
        xc::env_step<float, DevLds>(G, P.cfg, s, act, obs, reward, done, success, lds);
        if (live && G.l == 0) {
            const int64_t e = late_index(e_in);
            store_state(P, e, s);
            write_obs(obs, s, e, obs_out, ag_out, dg_out);
            rew_out[e] = reward;
            done_out[e] = done ? 1 : 0;
            succ_out[e] = success ? 1 : 0;
            if (done && P.auto_reset) {
                if (term_obs) {
                    float4 *o = reinterpret_cast<float4 *>(term_obs + e * xk::OBS_DIM);
#pragma unroll
                    for (int k = 0; k < xk::OBS_DIM / 4; k++) o[k] = make_float4(obs[4 * k], obs[4 * k + 1], obs[4 * k + 2], obs[4 * k + 3]);
                }
                const int pos = atomicAdd(done_count, 1);
                done_list[pos] = (int)e;
            }
        }
    }
}

// test hook: n internal substeps toward fixed joint targets (no action / IK / obs logic)
__global__ __launch_bounds__(WG) void k_substeps(KParams P, const float *__restrict__ qt_in, int n) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t e = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e >= P.num_envs) return;
    DevLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e, s);
    float qt[9];
#pragma unroll
    for (int k = 0; k < 9; k++) qt[k] = qt_in[e * 9 + k];
    const float dt = (float)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
#pragma unroll 1
    for (int k = 0; k < n; k++) xk::substep<float, DevLds>(s, qt, dt, lds);
    store_state(P, e, s);
}

__global__ void k_compact_mask(const uint8_t *__restrict__ mask, int64_t n, int *__restrict__ list, int *__restrict__ count) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n || !mask[e]) return;
    const int pos = atomicAdd(count, 1);
    list[pos] = (int)e;
}

// row-major [E, STATE_DIM] <-> structure-of-arrays [STATE_DIM][stride]
__global__ void k_get_state(KParams P, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.num_envs * P.state_dim) return;
    const int64_t e = i / P.state_dim, f = i % P.state_dim;
    out[i] = P.state[f * P.stride + e];
}
__global__ void k_set_state(KParams P, const float *__restrict__ in) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.num_envs * P.state_dim) return;
    const int64_t e = i / P.state_dim, f = i % P.state_dim;
    P.state[f * P.stride + e] = in[i];
}

__global__ void k_compute_reward(int reward_type, const float *__restrict__ ag, const float *__restrict__ g, int64_t n,
                                 float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = ag[i * 3] - g[i * 3], dy = ag[i * 3 + 1] - g[i * 3 + 1], dz = ag[i * 3 + 2] - g[i * 3 + 2];
    out[i] = xk::reward_of<float>(reward_type, sqrtf(dx * dx + dy * dy + dz * dz));
}

// ------------------------------------------------------------------------------ XarmReach-v0
__device__ __forceinline__ void reach_load(const KParams &P, int64_t e, xr::EnvState<float> &s) {
    const float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < xr::ND; i++) { s.q[i] = S[(xr::R_Q + i) * n]; s.qd[i] = S[(xr::R_QD + i) * n]; s.qt[i] = S[(xr::R_QT + i) * n]; }
#pragma unroll
    for (int i = 0; i < 3; i++) s.goal[i] = S[(xr::R_GOAL + i) * n];
    s.d_old = S[xr::R_DOLD * n]; s.steps = S[xr::R_STEPS * n]; s.episode = S[xr::R_EPISODE * n];
}
__device__ __forceinline__ void reach_store(const KParams &P, int64_t e, const xr::EnvState<float> &s) {
    float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < xr::ND; i++) { S[(xr::R_Q + i) * n] = s.q[i]; S[(xr::R_QD + i) * n] = s.qd[i]; S[(xr::R_QT + i) * n] = s.qt[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) S[(xr::R_GOAL + i) * n] = s.goal[i];
    S[xr::R_DOLD * n] = s.d_old; S[xr::R_STEPS * n] = s.steps; S[xr::R_EPISODE * n] = s.episode;
}
__device__ __forceinline__ void reach_write_obs(const float (&obs)[xr::OBS_DIM], const xr::EnvState<float> &s, int64_t e,
                                                float *obs_out, float *ag_out, float *dg_out) {
    float4 *o = reinterpret_cast<float4 *>(obs_out + e * xr::OBS_DIM);
    o[0] = make_float4(obs[0], obs[1], obs[2], obs[3]);
    o[1] = make_float4(obs[4], obs[5], obs[6], obs[7]);
#pragma unroll
    for (int k = 0; k < 3; k++) { ag_out[e * 3 + k] = obs[k]; dg_out[e * 3 + k] = s.goal[k]; }
}
__global__ __launch_bounds__(WG) void k_reach_init(KParams P) {
    const int64_t e = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e >= P.num_envs) return;
    xr::EnvState<float> s;
    xr::env_init<float>(P.rcfg, e, s);
    reach_store(P, e, s);
}
__global__ __launch_bounds__(WG) void k_reach_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                   float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                   float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                   uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                   int *__restrict__ done_list, int *__restrict__ done_count) {
    const int64_t e = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e >= P.num_envs) return;
    xr::EnvState<float> s;
    reach_load(P, e, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xr::OBS_DIM], reward;
    bool done, success;
    int fut;
    xr::env_step<float>(P.rcfg, s, act, obs, reward, done, success, fut);
    reach_store(P, e, s);
    reach_write_obs(obs, s, e, obs_out, ag_out, dg_out);
    rew_out[e] = reward;
    done_out[e] = done ? 1 : 0;
    succ_out[e] = success ? 1 : 0;
    if (done && P.auto_reset) {
        if (term_obs) {
            float4 *o = reinterpret_cast<float4 *>(term_obs + e * xr::OBS_DIM);
            o[0] = make_float4(obs[0], obs[1], obs[2], obs[3]);
            o[1] = make_float4(obs[4], obs[5], obs[6], obs[7]);
        }
        const int pos = atomicAdd(done_count, 1);
        done_list[pos] = (int)e;
    }
}
__global__ __launch_bounds__(WG) void k_reach_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                    float *__restrict__ obs_out, float *__restrict__ ag_out,
                                                    float *__restrict__ dg_out) {
    const int64_t i = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (n <= P.coop_limit) return;          // k_reach_reset_coop's range
    if (i >= n) return;
    const int64_t e = list ? (int64_t)list[i] : i;
    xr::EnvState<float> s;
    reach_load(P, e, s);
    float obs[xr::OBS_DIM];
    xr::env_reset<float>(P.rcfg, e, s, obs);
    reach_store(P, e, s);
    if (obs_out) reach_write_obs(obs, s, e, obs_out, ag_out, dg_out);
}
// XarmReach-v0 with one environment per DPP row of 16 lanes (xarm_reach_coop_core.h): lane l = body l = dof l.  At the
// BASELINE size (4 096 envs) the one-env-per-lane kernels above fill 64 of 1 024 SIMDs; these fill all of them and halve
// the dependent instructions of a substep.  Same outputs and done list.
__global__ __launch_bounds__(WG) void k_reach_step_coop(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                        float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                        float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                        uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                        int *__restrict__ done_list, int *__restrict__ done_count) {
    const int64_t e_raw = xcd_contiguous_block() * COOP_ENVS + threadIdx.x / xc::GL;
    const bool live = e_raw < P.num_envs;
    const int64_t e = live ? e_raw : P.num_envs - 1;
    const xc::Grp G{(int)(threadIdx.x & (xc::GL - 1))};
    xr::EnvState<float> s;
    reach_load(P, e, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xr::OBS_DIM], reward;
    bool done, success;
    int fut;
    xrc::env_step<float>(G, P.rcfg, s, act, obs, reward, done, success, fut);
    if (!live || G.l != 0) return;
    reach_store(P, e, s);
    reach_write_obs(obs, s, e, obs_out, ag_out, dg_out);
    rew_out[e] = reward;
    done_out[e] = done ? 1 : 0;
    succ_out[e] = success ? 1 : 0;
    if (done && P.auto_reset) {
        if (term_obs) {
            float4 *o = reinterpret_cast<float4 *>(term_obs + e * xr::OBS_DIM);
            o[0] = make_float4(obs[0], obs[1], obs[2], obs[3]);
            o[1] = make_float4(obs[4], obs[5], obs[6], obs[7]);
        }
        const int pos = atomicAdd(done_count, 1);
        done_list[pos] = (int)e;
    }
}
__global__ __launch_bounds__(WG) void k_reach_reset_coop(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                         float *__restrict__ obs_out, float *__restrict__ ag_out,
                                                         float *__restrict__ dg_out) {
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (n > P.coop_limit) return;
    // a reset list covers the front of a grid sized for coop_limit entries: the remap would pile its workgroups on one or
    // two XCDs, and the listed envs are scattered anyway - only the full-batch reset (no list) is remapped
    const int64_t i0 = (list ? (int64_t)blockIdx.x : xcd_contiguous_block()) * COOP_ENVS;
    if (i0 >= n) return;
    const int64_t i_raw = i0 + threadIdx.x / xc::GL;
    const bool live = i_raw < n;
    const int64_t i = live ? i_raw : n - 1;
    const int64_t e = list ? (int64_t)list[i] : i;
    const xc::Grp G{(int)(threadIdx.x & (xc::GL - 1))};
    xr::EnvState<float> s;
    reach_load(P, e, s);
    float obs[xr::OBS_DIM];
    xrc::env_reset<float>(G, P.rcfg, e, s, obs);
    if (!live || G.l != 0) return;
    reach_store(P, e, s);
    if (obs_out) reach_write_obs(obs, s, e, obs_out, ag_out, dg_out);
}
__global__ void k_reach_compute_reward(int reward_type, const float *__restrict__ ag, const float *__restrict__ g, int64_t n,
                                       float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = ag[i * 3] - g[i * 3], dy = ag[i * 3 + 1] - g[i * 3 + 1], dz = ag[i * 3 + 2] - g[i * 3 + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    out[i] = reward_type == 0 ? (d < (float)xmr::DISTANCE_THRESHOLD ? 1.f : 0.f) : -d;
}
// --------------------------------------------------------------------- XarmHandover-v0 (two lanes per env)
// lane-pair exchange by DPP quad permutes: lanes (2k, 2k+1) are arm 0 / arm 1 of one environment
struct DppXchg {
    __device__ __forceinline__ float from0(float v) const { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xA0, 0xF, 0xF, true)); }   // quad_perm [0,0,2,2]
    __device__ __forceinline__ float from1(float v) const { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xF5, 0xF, 0xF, true)); }   // quad_perm [1,1,3,3]
    __device__ __forceinline__ float partner(float v) const { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); } // quad_perm [1,0,3,2]
};
__device__ __forceinline__ void ho_load(const KParams &P, int64_t e, int arm, xh::Lane<float> &L) {
    const float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) { L.st.q[i] = S[(xh::H_Q + 9 * arm + i) * n]; L.st.qd[i] = S[(xh::H_QD + 9 * arm + i) * n]; }
    L.ft = S[(xh::H_FT + arm) * n];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        L.st.bp[i] = S[(xh::H_BP + i) * n]; L.st.bv[i] = S[(xh::H_BV + i) * n];
        L.st.bw[i] = S[(xh::H_BW + i) * n]; L.st.goal[i] = S[(xh::H_GOAL + i) * n];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { L.st.bq[i] = S[(xh::H_BQ + i) * n]; L.st.lam_p[i] = S[(xh::H_LP + 4 * arm + i) * n]; L.st.lam_p[4 + i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; i++) L.st.lam_t[i] = S[(xh::H_LT + i) * n];
    L.st.touch = S[(xh::H_TOUCH + arm) * n]; L.st.mug = S[(xh::H_MUG + arm) * n];
    L.st.steps = S[xh::H_STEPS * n]; L.st.episode = S[xh::H_EPISODE * n];
}
__device__ __forceinline__ void ho_store(const KParams &P, int64_t e, int arm, const xh::Lane<float> &L) {
    float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) { S[(xh::H_Q + 9 * arm + i) * n] = L.st.q[i]; S[(xh::H_QD + 9 * arm + i) * n] = L.st.qd[i]; }
    S[(xh::H_FT + arm) * n] = L.ft;
#pragma unroll
    for (int i = 0; i < 4; i++) S[(xh::H_LP + 4 * arm + i) * n] = L.st.lam_p[i];
    S[(xh::H_TOUCH + arm) * n] = L.st.touch; S[(xh::H_MUG + arm) * n] = L.st.mug;
    if (arm == 0) { // shared fields are bit-identical in both lanes
#pragma unroll
        for (int i = 0; i < 3; i++) {
            S[(xh::H_BP + i) * n] = L.st.bp[i]; S[(xh::H_BV + i) * n] = L.st.bv[i];
            S[(xh::H_BW + i) * n] = L.st.bw[i]; S[(xh::H_GOAL + i) * n] = L.st.goal[i];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) S[(xh::H_BQ + i) * n] = L.st.bq[i];
#pragma unroll
        for (int i = 0; i < 8; i++) S[(xh::H_LT + i) * n] = L.st.lam_t[i];
        S[xh::H_STEPS * n] = L.st.steps; S[xh::H_EPISODE * n] = L.st.episode;
    }
}
__device__ __forceinline__ void ho_write_obs(const xh::Lane<float> &L, int64_t e, int arm, float *obs_out, float *ag_out, float *dg_out) {
    float o8[8];
    xh::arm_obs(L, arm, o8);
    float *o = obs_out + e * xh::OBS_DIM;
#pragma unroll
    for (int k = 0; k < 8; k++) o[13 + 8 * arm + k] = o8[k];
    if (arm == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { o[k] = L.st.bp[k]; o[7 + k] = L.st.bv[k]; o[10 + k] = L.st.bw[k]; ag_out[e * 3 + k] = L.st.bp[k]; dg_out[e * 3 + k] = L.st.goal[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) o[3 + k] = L.st.bq[k];
    }
}
__global__ __launch_bounds__(WG) void k_ho_init(KParams P) {
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, e = t >> 1;
    const int arm = (int)(t & 1);
    if (e >= P.num_envs) return;
    xh::Lane<float> L;
    xh::lane_init<float>(P.hcfg, e, L);
    ho_store(P, e, arm, L);
}
// XarmHandover.step for every env (list == null) or for the envs list[0 .. *count) (the hand-off of k_ho_step_fast when it
// is too long for the cooperative kernel)
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                int *__restrict__ done_list, int *__restrict__ done_count,
                                                const int *__restrict__ list, const int *__restrict__ count) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, i_in = t >> 1;
    const int arm = (int)(t & 1);
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (count && n <= P.eject_coop_cap) return;     // k_ho_step_coop_list's range
    if (i_in >= n) return;
    const int64_t e_in = list ? (int64_t)list[i_in] : i_in;
    DevLds lds{smem + threadIdx.x};
    xh::Lane<float> L;
    ho_load(P, e_in, arm, L);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in * 2 + arm];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float reward;
    bool done, success;
    xh::lane_step<float, DevLds, DppXchg, Scene>(L, arm, act, reward, done, success, lds, DppXchg(), P.hcfg.reward_type);
    const int64_t e = late_index(e_in);
    ho_store(P, e, arm, L);
    ho_write_obs(L, e, arm, obs_out, ag_out, dg_out);
    if (done && P.auto_reset && term_obs) ho_write_obs(L, e, arm, term_obs, ag_out, dg_out);
    if (arm == 0) {
        rew_out[e] = reward;
        done_out[e] = done ? 1 : 0;
        succ_out[e] = success ? 1 : 0;
        if (done && P.auto_reset) {
            const int pos = atomicAdd(done_count, 1);
            done_list[pos] = (int)e;
        }
    }
}
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                 float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, i = t >> 1;
    const int arm = (int)(t & 1);
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (n <= P.coop_limit) return;                  // k_ho_reset_coop's range
    if (i >= n) return;
    const int64_t e_in = list ? (int64_t)list[i] : i;
    DevLds lds{smem + threadIdx.x};
    xh::Lane<float> L;
    ho_load(P, e_in, arm, L);
    xh::lane_reset<float, DevLds, DppXchg, Scene>(P.hcfg, e_in, L, arm, lds, DppXchg());
    const int64_t e = late_index(e_in);
    ho_store(P, e, arm, L);
    if (obs_out) ho_write_obs(L, e, arm, obs_out, ag_out, dg_out);
}
__global__ void k_ho_compute_reward(const float *__restrict__ ag, const float *__restrict__ g, int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = ag[i * 3] - g[i * 3], dy = ag[i * 3 + 1] - g[i * 3 + 1], dz = ag[i * 3 + 2] - g[i * 3 + 2];
    out[i] = sqrtf(dx * dx + dy * dy + dz * dz) > (float)xm::HO_DISTANCE_THRESHOLD ? -1.f : 0.f;
}

// The fast Handover step: XarmHandover.step on the pad-free lane-pair substep for every env (xh::lane_step_fast).  An env none
// of whose finger pads comes within the solver margin of the stick during the step is finished here; an env with an active
// pad row on either arm stores NOTHING and is appended to eject_list: it is stepped again, from its untouched state, by
// k_ho_step_coop_list (or k_ho_step when the list is long).  Why: a wavefront of k_ho_step with ONE such lane sweeps the pad
// blocks for all 32 of its envs (2.3 ms against 0.96 ms for a contact-free batch, tools/ho_time.py).  Only the support-slot
// columns live in LDS.
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho_step_fast(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                     float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                     float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                     uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                     int *__restrict__ done_list, int *__restrict__ done_count,
                                                     int *__restrict__ eject_list, int *__restrict__ eject_count) {
    __shared__ float smem[FAST_LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, e_in = t >> 1;
    const int arm = (int)(t & 1);
    if (e_in >= P.num_envs) return;
    FastLds lds{smem + threadIdx.x};
    xh::Lane<float> L;
    ho_load(P, e_in, arm, L);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in * 2 + arm];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float reward;
    bool done, success;
    const bool ok = xh::lane_step_fast<float, FastLds, DppXchg, Scene>(L, arm, act, reward, done, success, lds, DppXchg(), P.hcfg.reward_type);
    const int64_t e = late_index(e_in);
    if (!ok) {
        if (arm == 0) {
            const int pos = atomicAdd(eject_count, 1);
            eject_list[pos] = (int)e;
        }
        return;
    }
    ho_store(P, e, arm, L);
    ho_write_obs(L, e, arm, obs_out, ag_out, dg_out);
    if (done && P.auto_reset && term_obs) ho_write_obs(L, e, arm, term_obs, ag_out, dg_out);
    if (arm == 0) {
        rew_out[e] = reward;
        done_out[e] = done ? 1 : 0;
        succ_out[e] = success ? 1 : 0;
        if (done && P.auto_reset) {
            const int pos = atomicAdd(done_count, 1);
            done_list[pos] = (int)e;
        }
    }
}

// Row exchange of the cooperative Handover kernels (xarm_handover_coop_core.h): an environment owns two DPP rows of one
// wavefront, rows 0 / 1 = arm 0 of env slots 0 / 1, rows 2 / 3 = arm 1, i.e. lane l and lane l + 32 are the same lane of the
// two arms of one environment and ONE v_permlane32_swap_b32 (gfx950) hands a register across in both directions.
struct SwapXchg {
    int arm;
    __device__ __forceinline__ void pair(float v, float &v0, float &v1) const {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v0 = __uint_as_float(r[0]);   // lanes 0-31 (arm 0) everywhere
        v1 = __uint_as_float(r[1]);   // lanes 32-63 (arm 1) everywhere
    }
    __device__ __forceinline__ float from0(float v) const { float a, b; pair(v, a, b); return a; }
    __device__ __forceinline__ float from1(float v) const { float a, b; pair(v, a, b); return b; }
    __device__ __forceinline__ float partner(float v) const { float a, b; pair(v, a, b); return arm == 0 ? b : a; }
    __device__ __forceinline__ void both(xc::LV<float> v, xc::LV<float> &v0, xc::LV<float> &v1) const { pair(v.v[0], v0.v[0], v1.v[0]); }
};
constexpr int HO_COOP_LDS_FLOATS = xk::LDS_T;     // only the joint motion axes S are staged by the cooperative core
// XarmHandover.step of the envs list[0 .. *count) (null: all) on the cooperative rows: the hand-off of k_ho_step_fast.  The grid
// is fixed (the count lives on the device); a workgroup walks the list with a grid stride, two envs per wavefront.  Lists
// longer than P.eject_coop_cap belong to k_ho_step (launched beside this kernel; exactly one of the two does the work).
template <typename Scene, bool FORCE_COUPLED>
__global__ __launch_bounds__(WG) void k_ho_step_coop_list(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                          float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                          float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                          uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                          int *__restrict__ done_list, int *__restrict__ done_count,
                                                          const int *__restrict__ list, const int *__restrict__ count) {
    __shared__ float smem[HO_COOP_LDS_FLOATS * WG];
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (count && n > P.eject_coop_cap) return;
    const int row = (int)(threadIdx.x >> 4), slot = row & 1;
    const xc::Grp G{(int)(threadIdx.x & (xc::GL - 1))};
    const SwapXchg x{row >> 1};
    DevLds lds{smem + threadIdx.x};
#pragma unroll 1
    for (int64_t i0 = (int64_t)blockIdx.x * xhc::ROW_ENVS; i0 < n; i0 += (int64_t)gridDim.x * xhc::ROW_ENVS) {
        const int64_t i_raw = i0 + slot;
        const bool live = i_raw < n;
        const int64_t i = live ? i_raw : n - 1;
        const int64_t e_in = list ? (int64_t)list[i] : i;
        xh::Lane<float> L;
        ho_load(P, e_in, x.arm, L);
        const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in * 2 + x.arm];
        const float act[4] = {a4.x, a4.y, a4.z, a4.w};
        float reward;
        bool done, success;
        xhc::env_step<float, DevLds, SwapXchg, Scene, FORCE_COUPLED>(G, x, L, act, reward, done, success, lds, P.hcfg.reward_type);
        if (live && G.l == 0) {
            const int64_t e = late_index(e_in);
            ho_store(P, e, x.arm, L);
            ho_write_obs(L, e, x.arm, obs_out, ag_out, dg_out);
            if (done && P.auto_reset && term_obs) ho_write_obs(L, e, x.arm, term_obs, ag_out, dg_out);
            if (x.arm == 0) {
                rew_out[e] = reward;
                done_out[e] = done ? 1 : 0;
                succ_out[e] = success ? 1 : 0;
                if (done && P.auto_reset) {
                    const int pos = atomicAdd(done_count, 1);
                    done_list[pos] = (int)e;
                }
            }
        }
    }
}
// XarmHandover.reset on the cooperative rows for the envs list[0 .. *count) (null: all), counts up to P.coop_limit (more:
// k_ho_reset, launched beside this kernel): six ticks of latency for the handful of envs that finish in a step
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho_reset_coop(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                      float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out) {
    __shared__ float smem[HO_COOP_LDS_FLOATS * WG];
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (n > P.coop_limit) return;
    const int row = (int)(threadIdx.x >> 4), slot = row & 1;
    const xc::Grp G{(int)(threadIdx.x & (xc::GL - 1))};
    const SwapXchg x{row >> 1};
    DevLds lds{smem + threadIdx.x};
#pragma unroll 1
    for (int64_t i0 = (int64_t)blockIdx.x * xhc::ROW_ENVS; i0 < n; i0 += (int64_t)gridDim.x * xhc::ROW_ENVS) {
        const int64_t i_raw = i0 + slot;
        const bool live = i_raw < n;
        const int64_t i = live ? i_raw : n - 1;
        const int64_t e_in = list ? (int64_t)list[i] : i;
        xh::Lane<float> L;
        ho_load(P, e_in, x.arm, L);
        xhc::env_reset<float, DevLds, SwapXchg, Scene>(G, x, P.hcfg, e_in, L, lds);
        if (live && G.l == 0) {
            const int64_t e = late_index(e_in);
            ho_store(P, e, x.arm, L);
            if (obs_out) ho_write_obs(L, e, x.arm, obs_out, ag_out, dg_out);
        }
    }
}

// --------------------------------------------------------------------- XarmHandover-v0, num_obj = 2 (two lanes per env)
// the reference's test.py configuration (test.py:9-15); core xarm_handover2_core.h.  411 LDS floats per lane = 105 KB per
// wavefront: one wavefront per CU.
__device__ __forceinline__ void h2_load(const KParams &P, int64_t e, int arm, xh2::Lane<float> &L) {
    const float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) { L.q[i] = S[(xh2::G_Q + 9 * arm + i) * n]; L.qd[i] = S[(xh2::G_QD + 9 * arm + i) * n]; }
    L.ft = S[(xh2::G_FT + arm) * n];
#pragma unroll
    for (int o = 0; o < xh2::NOBJ; o++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            L.bp[o][k] = S[(xh2::G_BP + 3 * o + k) * n]; L.bv[o][k] = S[(xh2::G_BV + 3 * o + k) * n];
            L.bw[o][k] = S[(xh2::G_BW + 3 * o + k) * n]; L.goal[o][k] = S[(xh2::G_GOAL + 3 * o + k) * n];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) L.bq[o][k] = S[(xh2::G_BQ + 4 * o + k) * n];
#pragma unroll
        for (int k = 0; k < 8; k++) L.lam_t[o][k] = S[(xh2::G_LT + 8 * o + k) * n];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) L.lam_p[k] = S[(xh2::G_LP + 4 * arm + k) * n];
    L.touch = S[(xh2::G_TOUCH + arm) * n]; L.mug = S[(xh2::G_MUG + arm) * n];
    L.steps = S[xh2::G_STEPS * n]; L.episode = S[xh2::G_EPISODE * n];
}
__device__ __forceinline__ void h2_store(const KParams &P, int64_t e, int arm, const xh2::Lane<float> &L) {
    float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) { S[(xh2::G_Q + 9 * arm + i) * n] = L.q[i]; S[(xh2::G_QD + 9 * arm + i) * n] = L.qd[i]; }
    S[(xh2::G_FT + arm) * n] = L.ft;
#pragma unroll
    for (int k = 0; k < 4; k++) S[(xh2::G_LP + 4 * arm + k) * n] = L.lam_p[k];
    S[(xh2::G_TOUCH + arm) * n] = L.touch; S[(xh2::G_MUG + arm) * n] = L.mug;
    if (arm == 0) { // shared fields are bit-identical in both lanes
#pragma unroll
        for (int o = 0; o < xh2::NOBJ; o++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                S[(xh2::G_BP + 3 * o + k) * n] = L.bp[o][k]; S[(xh2::G_BV + 3 * o + k) * n] = L.bv[o][k];
                S[(xh2::G_BW + 3 * o + k) * n] = L.bw[o][k]; S[(xh2::G_GOAL + 3 * o + k) * n] = L.goal[o][k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) S[(xh2::G_BQ + 4 * o + k) * n] = L.bq[o][k];
#pragma unroll
            for (int k = 0; k < 8; k++) S[(xh2::G_LT + 8 * o + k) * n] = L.lam_t[o][k];
        }
        S[xh2::G_STEPS * n] = L.steps; S[xh2::G_EPISODE * n] = L.episode;
    }
}
// observation (:314-329): stick pos 6, quat 8, v 6, w 6, then per arm grip pos 3, hand vel 3, finger q, qd
__device__ __forceinline__ void h2_write_obs(const xh2::Lane<float> &L, int64_t e, int arm, float *obs_out, float *ag_out, float *dg_out) {
    float o8[8];
    xh2::arm_obs(L, arm, o8);
    float *o = obs_out + e * xh2::OBS_DIM;
#pragma unroll
    for (int k = 0; k < 8; k++) o[26 + 8 * arm + k] = o8[k];
    if (arm == 0) {
#pragma unroll
        for (int ob = 0; ob < xh2::NOBJ; ob++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                o[3 * ob + k] = L.bp[ob][k]; o[14 + 3 * ob + k] = L.bv[ob][k]; o[20 + 3 * ob + k] = L.bw[ob][k];
                if (ag_out) { ag_out[e * 6 + 3 * ob + k] = L.bp[ob][k]; dg_out[e * 6 + 3 * ob + k] = L.goal[ob][k]; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) o[6 + 4 * ob + k] = L.bq[ob][k];
        }
    }
}
__global__ __launch_bounds__(WG) void k_ho2_init(KParams P) {
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, e = t >> 1;
    const int arm = (int)(t & 1);
    if (e >= P.num_envs) return;
    xh2::Lane<float> L;
    xh2::lane_init<float>(P.hcfg, e, L);
    h2_store(P, e, arm, L);
}
__global__ __launch_bounds__(WG) void k_ho2_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                 float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                 float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                 uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                 int *__restrict__ done_list, int *__restrict__ done_count) {
    __shared__ float smem[xh2::LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, e_in = t >> 1;
    const int arm = (int)(t & 1);
    if (e_in >= P.num_envs) return;
    DevLds lds{smem + threadIdx.x};
    xh2::Lane<float> L;
    h2_load(P, e_in, arm, L);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in * 2 + arm];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float reward;
    bool done, success;
    xh2::lane_step<float, DevLds, DppXchg>(L, arm, act, reward, done, success, lds, DppXchg());
    const int64_t e = late_index(e_in);
    h2_store(P, e, arm, L);
    h2_write_obs(L, e, arm, obs_out, ag_out, dg_out);
    if (done && P.auto_reset && term_obs) h2_write_obs(L, e, arm, term_obs, nullptr, nullptr);
    if (arm == 0) {
        rew_out[e] = reward;
        done_out[e] = done ? 1 : 0;
        succ_out[e] = success ? 1 : 0;
        if (done && P.auto_reset) {
            const int pos = atomicAdd(done_count, 1);
            done_list[pos] = (int)e;
        }
    }
}
__global__ __launch_bounds__(WG) void k_ho2_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                  float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out) {
    __shared__ float smem[xh2::LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, i = t >> 1;
    const int arm = (int)(t & 1);
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (i >= n) return;
    const int64_t e_in = list ? (int64_t)list[i] : i;
    DevLds lds{smem + threadIdx.x};
    xh2::Lane<float> L;
    h2_load(P, e_in, arm, L);
    xh2::lane_reset<float, DevLds, DppXchg>(P.hcfg, e_in, L, arm, lds, DppXchg());
    const int64_t e = late_index(e_in);
    h2_store(P, e, arm, L);
    if (obs_out) h2_write_obs(L, e, arm, obs_out, ag_out, dg_out);
}
// xarm_handover.py:177-183 over n rows of 6: -sum_i [|ag_i - g_i| > thr]
__global__ void k_ho2_compute_reward(const float *__restrict__ ag, const float *__restrict__ g, int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.f;
#pragma unroll
    for (int o = 0; o < 2; o++) {
        const float dx = ag[i * 6 + 3 * o] - g[i * 6 + 3 * o], dy = ag[i * 6 + 3 * o + 1] - g[i * 6 + 3 * o + 1], dz = ag[i * 6 + 3 * o + 2] - g[i * 6 + 3 * o + 2];
        r += sqrtf(dx * dx + dy * dy + dz * dz) > (float)xm::HO_DISTANCE_THRESHOLD ? 1.f : 0.f;
    }
    out[i] = -r;
}

// --------------------------------------------------------------------- XarmPDStackTower-v0 (two lanes per env)
// 603 LDS floats per lane = 151 KB per wavefront: one wavefront per CU, which is this scene's BASELINE size
// (8192 envs per GPU = 256 wavefronts)
__device__ __forceinline__ void st_load(const KParams &P, int64_t e, int arm, xs::Lane<float> &L) {
    const float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        L.q[i] = S[(xs::K_Q + 9 * arm + i) * n]; L.qd[i] = S[(xs::K_QD + 9 * arm + i) * n]; L.qt[i] = S[(xs::K_QT + 9 * arm + i) * n];
    }
#pragma unroll
    for (int o = 0; o < xs::NOBJ; o++) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            L.bp[o][k] = S[(xs::K_BP + 3 * o + k) * n]; L.bv[o][k] = S[(xs::K_BV + 3 * o + k) * n];
            L.bw[o][k] = S[(xs::K_BW + 3 * o + k) * n]; L.goal[o][k] = S[(xs::K_GOAL + 3 * o + k) * n];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) L.bq[o][k] = S[(xs::K_BQ + 4 * o + k) * n];
#pragma unroll
        for (int k = 0; k < 8; k++) L.lam_t[o][k] = S[(xs::K_LT + 8 * o + k) * n];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) L.lam_p[k] = S[(xs::K_LP + 4 * arm + k) * n];
    L.steps = S[xs::K_STEPS * n]; L.episode = S[xs::K_EPISODE * n];
    L.cls = 0;
}
__device__ __forceinline__ void st_store(const KParams &P, int64_t e, int arm, const xs::Lane<float> &L) {
    float *S = P.state + e;
    const int64_t n = P.stride;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        S[(xs::K_Q + 9 * arm + i) * n] = L.q[i]; S[(xs::K_QD + 9 * arm + i) * n] = L.qd[i]; S[(xs::K_QT + 9 * arm + i) * n] = L.qt[i];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) S[(xs::K_LP + 4 * arm + k) * n] = L.lam_p[k];
    if (arm == 0) { // shared fields are bit-identical in both lanes
#pragma unroll
        for (int o = 0; o < xs::NOBJ; o++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                S[(xs::K_BP + 3 * o + k) * n] = L.bp[o][k]; S[(xs::K_BV + 3 * o + k) * n] = L.bv[o][k];
                S[(xs::K_BW + 3 * o + k) * n] = L.bw[o][k]; S[(xs::K_GOAL + 3 * o + k) * n] = L.goal[o][k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) S[(xs::K_BQ + 4 * o + k) * n] = L.bq[o][k];
#pragma unroll
            for (int k = 0; k < 8; k++) S[(xs::K_LT + 8 * o + k) * n] = L.lam_t[o][k];
        }
        S[xs::K_STEPS * n] = L.steps; S[xs::K_EPISODE * n] = L.episode;
    }
}
// observation (:190-199): cube pos 9, quat 12, v 9, w 9, then per arm hand COM pos 3, vel 3, finger q, qd
__device__ __forceinline__ void st_write_obs(const xs::Lane<float> &L, int64_t e, int arm, float *obs_out, float *ag_out, float *dg_out) {
    float o8[8];
    xs::arm_obs(L, arm, o8);
    float *o = obs_out + e * xs::OBS_DIM;
#pragma unroll
    for (int k = 0; k < 8; k++) o[39 + 8 * arm + k] = o8[k];
    if (arm == 0) {
#pragma unroll
        for (int ob = 0; ob < xs::NOBJ; ob++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                o[3 * ob + k] = L.bp[ob][k]; o[21 + 3 * ob + k] = L.bv[ob][k]; o[30 + 3 * ob + k] = L.bw[ob][k];
                if (ag_out) { ag_out[e * 9 + 3 * ob + k] = L.bp[ob][k]; dg_out[e * 9 + 3 * ob + k] = L.goal[ob][k]; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) o[9 + 4 * ob + k] = L.bq[ob][k];
        }
    }
}
__global__ __launch_bounds__(WG) void k_st_init(KParams P) {
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, e = t >> 1;
    const int arm = (int)(t & 1);
    if (e >= P.num_envs) return;
    xs::Lane<float> L;
    xs::lane_init<float>(P.cfg, e, L);
    st_store(P, e, arm, L);
}
// class-homogeneous wavefronts (xarm_stack_core.h class_layout): histogram of the per-env class keys, then every env takes
// the next slot of its class; order[slot] = env is the order k_st_step visits the envs in.  The arrival order inside a
// class comes from an atomic counter and differs from run to run - it decides which wavefront an env shares, never its
// result (an env is bitwise independent of its neighbours).
__global__ void k_class_hist(const uint8_t *__restrict__ key, int64_t n, int *__restrict__ hist) {
    __shared__ int h[xs::NCLS];
    if (threadIdx.x < xs::NCLS) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) atomicAdd(&h[key[e] & (xs::NCLS - 1)], 1);
    __syncthreads();
    if (threadIdx.x < xs::NCLS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_class_place(const uint8_t *__restrict__ key, int64_t n, const int *__restrict__ hist, int *__restrict__ cursor,
                              int *__restrict__ order, int group) {
    __shared__ xs::ClassLayout Y;
    __shared__ int cnt[xs::NCLS], base[xs::NCLS];
    if (threadIdx.x < xs::NCLS) cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        int hh[xs::NCLS];
        for (int c = 0; c < xs::NCLS; c++) hh[c] = hist[c];
        xs::class_layout(hh, group, Y);
    }
    __syncthreads();
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = e < n ? (int)(key[e] & (xs::NCLS - 1)) : 0;
    // arrival number inside the class: rank inside the block (LDS counter), one global atomic per block and class -
    // thousands of class-0 envs on one global counter cost 90 us per call
    const int local = e < n ? atomicAdd(&cnt[c], 1) : 0;
    __syncthreads();
    if (threadIdx.x < xs::NCLS && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]);
    __syncthreads();
    if (e >= n) return;
    const int slot = xs::class_slot(Y, c, base[c] + local);
    if (slot >= 0 && slot < n) order[slot] = (int)e;   // always true for a histogram of these keys; never write outside
}
__global__ __launch_bounds__(WG) void k_st_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                int *__restrict__ done_list, int *__restrict__ done_count,
                                                const int *__restrict__ order, uint8_t *__restrict__ key) {
    __shared__ float smem[xs::LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, slot = t >> 1;
    const int arm = (int)(t & 1);
    if (slot >= P.num_envs) return;
    const int64_t e_in = order ? (int64_t)order[slot] : slot;
    DevLds lds{smem + threadIdx.x};
    xs::Lane<float> L;
    st_load(P, e_in, arm, L);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in * 2 + arm];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float reward;
    bool done, success;
    xs::lane_step<float, DevLds, DppXchg>(P.cfg, L, arm, act, reward, done, success, lds, DppXchg());
    const int64_t e = late_index(e_in);
    st_store(P, e, arm, L);
    st_write_obs(L, e, arm, obs_out, ag_out, dg_out);
    if (done && P.auto_reset && term_obs) st_write_obs(L, e, arm, term_obs, nullptr, nullptr);
    if (arm == 0) {
        if (key) key[e] = (uint8_t)L.cls;
        rew_out[e] = reward;
        done_out[e] = done ? 1 : 0;
        succ_out[e] = success ? 1 : 0;
        if (done && P.auto_reset) {
            const int pos = atomicAdd(done_count, 1);
            done_list[pos] = (int)e;
        }
    }
}
__global__ __launch_bounds__(WG) void k_st_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                 float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                 uint8_t *__restrict__ key) {
    __shared__ float smem[xs::LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, i = t >> 1;
    const int arm = (int)(t & 1);
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (i >= n) return;
    const int64_t e_in = list ? (int64_t)list[i] : i;
    DevLds lds{smem + threadIdx.x};
    xs::Lane<float> L;
    st_load(P, e_in, arm, L);
    xs::lane_reset<float, DevLds, DppXchg>(P.cfg, e_in, L, arm, lds, DppXchg());
    const int64_t e = late_index(e_in);
    st_store(P, e, arm, L);
    if (key && arm == 0) key[e] = (uint8_t)L.cls;
    if (obs_out) st_write_obs(L, e, arm, obs_out, ag_out, dg_out);
}
// xarm_stack_tower.py:124-129 over n rows of 9
__global__ void k_st_compute_reward(int reward_type, const float *__restrict__ ag, const float *__restrict__ g, int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < 9; k++) { const float d = ag[i * 9 + k] - g[i * 9 + k]; d2 += d * d; }
    const float d = sqrtf(d2);
    out[i] = reward_type == 0 ? (d > (float)xm::ST_DISTANCE_THRESHOLD ? -1.f : 0.f) : -d;
}

// number of steps taken in the current episode (info['future_length'] = max_episode_steps - steps, :90)
__global__ void k_episode_steps(KParams P, int steps_field, int32_t *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.num_envs) return;
    out[e] = (int32_t)P.state[(int64_t)steps_field * P.stride + e];
}

} // namespace

// ------------------------------------------------------------------------------------- C ABI
struct xarm_handle {
    xarm_config cfg;
    KParams kp;
    int *done_list;   // [E]
    int *counters;    // [3 + 2 * NCLS], zeroed by ONE memset at the start of every step call (no host-side state: a captured
                      // step can be replayed): done_count = +0, eject_count = +1 (2), class_hist = +3
    int *done_count;  // [1] episodes that ended in this call (list A)
    int *mask_count;  // [1]
    int coop_step_limit; // PickAndPlace: batches of at most this many envs step on k_step_coop
    int fast_pipeline;   // PickAndPlace, larger batches: k_step_fast + hand-off of the envs with finger-pad rows (1) or k_step (0)
    int *eject_list;     // [E] envs handed off by k_step_fast
    int *eject_count;    // [2]: hand-off count, count of episodes that ended in the hand-off kernels
    // the two reset launches of a pipelined step (xarm_step): episodes that ended in k_step_fast are reset on `side`
    // while the hand-off still runs on the caller's stream, the few that end in the hand-off after it
    int *done_list_b;    // [E] episodes that ended in the hand-off kernels
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    int reset_overlap;
    int ho_force_coupled; // test hook (XARM_HO_FORCE_COUPLED=1): every substep of the cooperative Handover step through the coupled sweep
    // StackTower: class-homogeneous wavefronts (xarm_stack_core.h class_layout); null when XARM_ST_CLASS_ORDER=0
    uint8_t *class_key;  // [E] row-set class of each env's last substep
    int *class_hist;     // [2 * NCLS] histogram, then the per-class arrival counters
    int *class_order;    // [E] slot -> env
    char err[512];
    // timing
    int timing;
    static constexpr int NEV = 1024;
    hipEvent_t ev0[NEV], ev1[NEV], ev2[NEV];   // before the step kernel, after it, after the reset kernels
    int ev_n;
    double ev_ms, ev_reset_ms;
    int64_t ev_launches;
    bool ev_created;
};

static char g_err[512] = "";

static int fail(xarm_handle *h, int code, const char *fmt, const char *detail) {
    char *dst = h ? h->err : g_err;
    snprintf(dst, 512, fmt, detail);
    return code;
}
#define HIPCHK(h, call)                                                          \
    do {                                                                         \
        hipError_t _e = (call);                                                  \
        if (_e != hipSuccess) return fail(h, XARM_E_HIP, #call ": %s", hipGetErrorString(_e)); \
    } while (0)

// grid of a cooperative Handover launch over at most `cap` envs: two envs per wavefront, grid stride beyond 2 048 workgroups
static unsigned ho_coop_grid(int64_t cap) {
    const int64_t g = (cap + xhc::ROW_ENVS - 1) / xhc::ROW_ENVS;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
// Handover reset of the envs in list[0 .. *count) (null: all): the cooperative rows take counts up to kp.coop_limit, the lane-pair
// kernel the rest; both are launched, the one out of its range exits at once (the count lives on the device)
static void launch_ho_reset(xarm_handle *h, unsigned grid2, const int *list, const int *count, float *obs_dev, float *ag_dev, float *dg_dev,
                            hipStream_t st) {
    if (h->cfg.num_obj == 2) { k_ho2_reset<<<dim3(grid2), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev); return; }
    const int64_t cap = h->kp.num_envs < (int64_t)h->kp.coop_limit ? h->kp.num_envs : (int64_t)h->kp.coop_limit;
    if (cap > 0) {
        if (h->kp.hcfg.use_stand) k_ho_reset_coop<xh::HandoverStandScene><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
        else k_ho_reset_coop<xh::HandoverScene><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    }
    if (h->kp.num_envs > cap) {
        if (h->kp.hcfg.use_stand) k_ho_reset<xh::HandoverStandScene><<<dim3(grid2), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
        else k_ho_reset<xh::HandoverScene><<<dim3(grid2), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    }
}

static void launch_reach_reset(xarm_handle *h, const int *list, const int *count, float *obs_dev, float *ag_dev, float *dg_dev, hipStream_t st) {
    const int64_t cap = h->kp.num_envs < (int64_t)h->kp.coop_limit ? h->kp.num_envs : (int64_t)h->kp.coop_limit;
    if (cap > 0)
        k_reach_reset_coop<<<dim3((unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    if (h->kp.num_envs > cap)
        k_reach_reset<<<dim3((unsigned)(h->kp.stride / WG)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
}

// PickAndPlace reset of the envs in list[0 .. *count) (null: all): the cooperative kernel takes counts up to
// kp.coop_limit, the one-env-per-lane kernel the rest; both are launched, the one out of its range exits at once.
static void launch_pnp_reset(xarm_handle *h, const int *list, const int *count, float *obs_dev, float *ag_dev, float *dg_dev,
                             hipStream_t st) {
    const int64_t cap = h->kp.num_envs < (int64_t)h->kp.coop_limit ? h->kp.num_envs : (int64_t)h->kp.coop_limit;
    if (cap > 0)
        k_reset_coop<<<dim3((unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    if (h->kp.num_envs > cap)
        k_reset<<<dim3((unsigned)(h->kp.stride / WG)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
}

// The handle's kernels, events and buffers live on cfg.device.  Every entry point makes that device current for its
// own duration and restores the caller's (torch's) current device on return, so a handle can be created and used
// while another device is current, and several handles on different GPUs can share a process.
struct DeviceGuard {
    int prev;
    bool switched;
    explicit DeviceGuard(int dev) : prev(-1), switched(false) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) hipSetDevice(prev); }
};
#define DEVGUARD(h) DeviceGuard _guard((h)->cfg.device)

static void timing_flush(xarm_handle *h) {
    for (int i = 0; i < h->ev_n; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(h->ev2[i]) == hipSuccess && hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]) == hipSuccess) {
            h->ev_ms += ms;
            h->ev_launches++;
            if (hipEventElapsedTime(&ms, h->ev1[i], h->ev2[i]) == hipSuccess) h->ev_reset_ms += ms;
        }
    }
    h->ev_n = 0;
}

extern "C" {

// a timing variant built with -DXC_SWEEP_ITERS=n (tools/coop_split.sh) runs fewer solver sweeps: it must never pass for the product
#define XARM_STR2(x) #x
#define XARM_STR(x) XARM_STR2(x)
#ifdef XARM_SWEEP_VARIANT
const char *xarm_version(void) { return "xarm_hip 0.1 (gfx950) TIMING VARIANT sweeps=" XARM_STR(XARM_SWEEP_VARIANT); }
#else
const char *xarm_version(void) { return "xarm_hip 0.1 (gfx950)"; }
#endif

const char *xarm_last_error(const xarm_handle *h) { return h ? h->err : g_err; }

int xarm_create(const xarm_config *cfg, xarm_handle **out) {
    if (!cfg || !out) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: null argument");
    const bool reach = cfg->env_kind == XARM_ENV_REACH, handover = cfg->env_kind == XARM_ENV_HANDOVER, stack = cfg->env_kind == XARM_ENV_STACK_TOWER;
    if (cfg->env_kind != XARM_ENV_PICK_AND_PLACE && !reach && !handover && !stack) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: unsupported env_kind");
    if (stack && cfg->num_obj != 3) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmStackTower has num_obj == 3 (xarm_stack_tower.py:19)");
    if (stack && cfg->reward_type > 1) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmStackTower reward_type is 0 (sparse) or 1 (-d)");
    if (handover && cfg->num_obj != 1 && cfg->num_obj != 2) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmHandover supports num_obj 1 or 2");
    if (handover && cfg->num_obj == 2 && (cfg->reward_type != 0 || cfg->use_stand))
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmHandover with num_obj == 2 takes the sparse reward and no stand (the reference's dense branch raises a broadcast error there, xarm_handover.py:187-188)");
    if (!reach && !stack && !handover && cfg->num_obj != 1)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmPickAndPlace supports num_obj == 1 (with more the reference's own step raises, xarm_pick_and_place.py:289-291)");
    if (handover && cfg->reward_type != 0 && cfg->reward_type != XARM_REWARD_DENSE)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmHandover reward_type is sparse (hard-wired in the reference, xarm_handover.py:40) or dense (:184-199)");
    if (cfg->use_stand && !handover) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: use_stand belongs to XarmHandover (xarm_handover.py:391-392)");
    if (cfg->auto_reset < 0 || cfg->auto_reset > XARM_AUTO_RESET_LAZY) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: auto_reset must be 0, 1 or XARM_AUTO_RESET_LAZY");
    if (cfg->auto_reset == XARM_AUTO_RESET_LAZY && cfg->env_kind != XARM_ENV_PICK_AND_PLACE)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: lazy auto-reset is implemented for XarmPickAndPlace only");
    if (cfg->num_envs <= 0 || cfg->num_envs > (int64_t)1 << 30) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: num_envs out of range");
    if (cfg->reward_type < 0 || cfg->reward_type > 2) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: unsupported reward_type");
    if (!reach && !stack && cfg->goal_shape != XARM_GOAL_AIR && cfg->goal_shape != XARM_GOAL_GROUND)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: unsupported goal_shape");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, XARM_E_NODEVICE, "%s", "xarm_create: no HIP device");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: bad device ordinal");
    DeviceGuard _guard(cfg->device);
    xarm_handle *h = new (std::nothrow) xarm_handle();
    if (!h) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: out of host memory");
    memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    const int64_t E = cfg->num_envs, stride = (E + WG - 1) / WG * WG;
    h->kp.stride = stride;
    h->kp.num_envs = E;
    h->kp.cfg.seed = cfg->seed;
    h->kp.cfg.env_id_offset = cfg->env_id_offset;
    h->kp.cfg.init_grasp_rate = cfg->init_grasp_rate;
    h->kp.cfg.goal_ground_rate = cfg->goal_ground_rate;
    h->kp.cfg.goal_shape = cfg->goal_shape;
    h->kp.cfg.reward_type = cfg->reward_type;
    h->kp.auto_reset = cfg->auto_reset;
    // cooperative reset kernel (PickAndPlace): default cross-over measured on MI355X; 0 disables it
    h->kp.coop_limit = 0;
    if (cfg->env_kind == XARM_ENV_PICK_AND_PLACE || cfg->env_kind == XARM_ENV_REACH) {
        h->kp.coop_limit = cfg->reset_coop_limit > 0 ? cfg->reset_coop_limit : (cfg->reset_coop_limit < 0 ? 0 : XARM_RESET_COOP_LIMIT_DEFAULT);
        // the environment variable replaces the DEFAULT only: an explicit xarm_config value (incl. "< 0 = never") wins
        const char *ev = getenv("XARM_RESET_COOP_LIMIT");
        if (ev && *ev && cfg->reset_coop_limit == 0) h->kp.coop_limit = atoi(ev) > 0 ? atoi(ev) : 0;
    }
    // cooperative step kernel: pays while the one-env-per-lane launch would leave SIMDs empty (measured cross-over,
    // DESIGN.md 5); XARM_STEP_COOP_LIMIT overrides, 0 disables
    h->coop_step_limit = (cfg->env_kind != XARM_ENV_PICK_AND_PLACE && cfg->env_kind != XARM_ENV_REACH) || cfg->step_coop_limit < 0 ? 0 :
                         (cfg->step_coop_limit > 0 ? cfg->step_coop_limit : XARM_STEP_COOP_LIMIT_DEFAULT);
    {
        const char *ev = getenv("XARM_STEP_COOP_LIMIT");
        if (ev && *ev && cfg->step_coop_limit == 0 && (cfg->env_kind == XARM_ENV_PICK_AND_PLACE || cfg->env_kind == XARM_ENV_REACH))
            h->coop_step_limit = atoi(ev) > 0 ? atoi(ev) : 0;
    }
    const bool handover2 = handover && cfg->num_obj == 2;
    const bool handover1 = handover && !handover2;
    // cooperative reset of Handover (one stick): two rows per env, measured cross-over against the lane-pair reset
    if (handover1) {
        h->kp.coop_limit = cfg->reset_coop_limit > 0 ? cfg->reset_coop_limit : (cfg->reset_coop_limit < 0 ? 0 : XARM_HO_RESET_COOP_LIMIT_DEFAULT);
        const char *ev = getenv("XARM_RESET_COOP_LIMIT");
        if (ev && *ev && cfg->reset_coop_limit == 0) h->kp.coop_limit = atoi(ev) > 0 ? atoi(ev) : 0;
    }
    // fast-step pipeline (PickAndPlace batches above the cooperative limit, Handover with one stick): on unless the caller
    // pinned the one-env-per-lane family (step_coop_limit < 0: gym_xarm_amd.distributed.reproducible_limits('lane')) or
    // XARM_STEP_PIPELINE=0 asks for the plain k_step / k_ho_step
    h->fast_pipeline = ((cfg->env_kind == XARM_ENV_PICK_AND_PLACE || handover1) && cfg->step_coop_limit >= 0 && cfg->auto_reset != XARM_AUTO_RESET_LAZY) ? 1 : 0;
    {
        const char *ev = getenv("XARM_STEP_PIPELINE");
        if (ev && *ev) h->fast_pipeline = h->fast_pipeline && atoi(ev) != 0;
        ev = getenv("XARM_HO_FORCE_COUPLED");
        h->ho_force_coupled = ev && *ev && atoi(ev) != 0;
    }
    h->kp.eject_coop_cap = handover1 ? XARM_HO_EJECT_COOP_CAP : XARM_EJECT_COOP_CAP;
    h->kp.state_dim = reach ? xr::STATE_DIM : (handover2 ? xh2::STATE_DIM : (handover ? xh::STATE_DIM : (stack ? xs::STATE_DIM : xk::STATE_DIM)));
    h->kp.hcfg.seed = cfg->seed;
    h->kp.hcfg.env_id_offset = cfg->env_id_offset;
    h->kp.hcfg.same_side_rate = cfg->same_side_rate;
    h->kp.hcfg.goal_shape = cfg->goal_shape;
    h->kp.hcfg.use_stand = handover && cfg->use_stand ? 1 : 0;
    h->kp.hcfg.reward_type = cfg->reward_type == XARM_REWARD_DENSE ? 1 : 0;
    h->kp.rcfg.seed = cfg->seed;
    h->kp.rcfg.env_id_offset = cfg->env_id_offset;
    h->kp.rcfg.reward_type = cfg->reward_type;
    hipError_t e1 = hipMalloc(&h->kp.state, sizeof(float) * h->kp.state_dim * stride);
    hipError_t e2 = hipMalloc(&h->done_list, sizeof(int) * stride);
    hipError_t e3 = hipMalloc(&h->counters, sizeof(int) * (3 + 2 * xs::NCLS));
    h->done_count = h->counters; h->eject_count = h->counters + 1; h->class_hist = h->counters + 3;
    hipError_t e4 = hipMalloc(&h->mask_count, sizeof(int));
    if (e4 == hipSuccess && h->fast_pipeline) {
        e4 = hipMalloc(&h->eject_list, sizeof(int) * stride);
        if (e4 == hipSuccess) e4 = hipMalloc(&h->done_list_b, sizeof(int) * stride);
        const char *ev = getenv("XARM_RESET_OVERLAP");
        if (e4 == hipSuccess && cfg->auto_reset && !(ev && *ev && atoi(ev) == 0)) {
            e4 = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
            if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
            if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
            h->reset_overlap = e4 == hipSuccess;
        }
    }
    if (e4 == hipSuccess && stack) {
        const char *ev = getenv("XARM_ST_CLASS_ORDER");
        if (!(ev && *ev && atoi(ev) == 0)) {
            e4 = hipMalloc(&h->class_key, stride);
            if (e4 == hipSuccess) e4 = hipMalloc(&h->class_order, sizeof(int) * stride);
            if (e4 == hipSuccess) e4 = hipMemset(h->class_key, 0, stride);
        }
    }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        fail(nullptr, XARM_E_HIP, "xarm_create: hipMalloc failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : (e3 != hipSuccess ? e3 : e4))));
        xarm_destroy(h);
        return XARM_E_HIP;
    }
    hipMemset(h->kp.state, 0, sizeof(float) * h->kp.state_dim * stride);
    hipMemset(h->counters, 0, sizeof(int) * (3 + 2 * xs::NCLS));
    hipMemset(h->mask_count, 0, sizeof(int));
    if (reach) k_reach_init<<<dim3((unsigned)(stride / WG)), dim3(WG)>>>(h->kp);
    else if (handover2) k_ho2_init<<<dim3((unsigned)(2 * stride / WG)), dim3(WG)>>>(h->kp);
    else if (handover) k_ho_init<<<dim3((unsigned)(2 * stride / WG)), dim3(WG)>>>(h->kp);
    else if (stack) k_st_init<<<dim3((unsigned)(2 * stride / WG)), dim3(WG)>>>(h->kp);
    else k_init<<<dim3((unsigned)(stride / WG)), dim3(WG)>>>(h->kp);
    hipError_t e5 = hipDeviceSynchronize();
    if (e5 != hipSuccess) {
        fail(nullptr, XARM_E_HIP, "xarm_create: k_init: %s", hipGetErrorString(e5));
        xarm_destroy(h);
        return XARM_E_HIP;
    }
    *out = h;
    return XARM_OK;
}

int xarm_destroy(xarm_handle *h) {
    if (!h) return XARM_OK;
    DEVGUARD(h);
    hipDeviceSynchronize();
    if (h->ev_created)
        for (int i = 0; i < xarm_handle::NEV; i++) { hipEventDestroy(h->ev0[i]); hipEventDestroy(h->ev1[i]); hipEventDestroy(h->ev2[i]); }
    if (h->kp.state) hipFree(h->kp.state);
    if (h->done_list) hipFree(h->done_list);
    if (h->counters) hipFree(h->counters);
    if (h->mask_count) hipFree(h->mask_count);
    if (h->eject_list) hipFree(h->eject_list);
    if (h->done_list_b) hipFree(h->done_list_b);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->side) hipStreamDestroy(h->side);
    if (h->class_key) hipFree(h->class_key);
    if (h->class_order) hipFree(h->class_order);
    delete h;
    return XARM_OK;
}

int xarm_dims(const xarm_handle *h, xarm_dims_t *out) {
    if (!out) return XARM_E_INVALID;
    const bool reach = h && h->cfg.env_kind == XARM_ENV_REACH, handover = h && h->cfg.env_kind == XARM_ENV_HANDOVER;
    const bool stack = h && h->cfg.env_kind == XARM_ENV_STACK_TOWER;
    const bool handover2 = handover && h->cfg.num_obj == 2;
    out->obs_dim = reach ? xr::OBS_DIM : (handover2 ? xh2::OBS_DIM : (handover ? xh::OBS_DIM : (stack ? xs::OBS_DIM : xk::OBS_DIM)));
    out->goal_dim = stack ? xs::GOAL_DIM : (handover2 ? xh2::GOAL_DIM : xk::GOAL_DIM);
    out->act_dim = handover ? xh::ACT_DIM : (stack ? xs::ACT_DIM : xk::ACT_DIM);
    out->state_dim = reach ? xr::STATE_DIM : (handover2 ? xh2::STATE_DIM : (handover ? xh::STATE_DIM : (stack ? xs::STATE_DIM : xk::STATE_DIM)));
    out->max_episode_steps = reach ? xmr::MAX_EPISODE_STEPS : (handover ? xm::HO_MAX_EPISODE_STEPS : (stack ? xm::ST_MAX_EPISODE_STEPS : xm::PNP_MAX_EPISODE_STEPS));
    out->n_substeps = reach ? xmr::N_SUBSTEPS : (handover ? xm::HO_N_TICKS : (stack ? xm::ST_N_SUBSTEPS : xm::PNP_N_SUBSTEPS));
    return XARM_OK;
}

int xarm_reset(xarm_handle *h, const uint8_t *mask_dev, float *obs_dev, float *ag_dev, float *dg_dev, void *stream) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (obs_dev && (!ag_dev || !dg_dev)) return fail(h, XARM_E_INVALID, "%s", "xarm_reset: goal buffers required with obs");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(h->kp.stride / WG);
    if (mask_dev) {
        HIPCHK(h, hipMemsetAsync(h->mask_count, 0, sizeof(int), st));
        k_compact_mask<<<dim3((unsigned)((h->kp.num_envs + 255) / 256)), dim3(256), 0, st>>>(mask_dev, h->kp.num_envs, h->done_list, h->mask_count);
        if (h->cfg.env_kind == XARM_ENV_REACH) launch_reach_reset(h, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_HANDOVER) launch_ho_reset(h, 2 * grid, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_STACK_TOWER) k_st_reset<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, h->class_key);
        else launch_pnp_reset(h, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, st);
    } else {
        if (h->cfg.env_kind == XARM_ENV_REACH) launch_reach_reset(h, nullptr, nullptr, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_HANDOVER) launch_ho_reset(h, 2 * grid, nullptr, nullptr, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_STACK_TOWER) k_st_reset<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, nullptr, nullptr, obs_dev, ag_dev, dg_dev, h->class_key);
        else launch_pnp_reset(h, nullptr, nullptr, obs_dev, ag_dev, dg_dev, st);
    }
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_step(xarm_handle *h, const float *actions_dev, float *obs_dev, float *ag_dev, float *dg_dev, float *reward_dev,
              uint8_t *done_dev, uint8_t *success_dev, float *terminal_obs_dev, void *stream) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (!actions_dev || !obs_dev || !ag_dev || !dg_dev || !reward_dev || !done_dev || !success_dev)
        return fail(h, XARM_E_INVALID, "%s", "xarm_step: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(h->kp.stride / WG);
    int *cnt = h->done_count;
    const bool timed = h->timing && h->ev_created;
    bool pipelined = false;
    const bool overlap = h->reset_overlap && h->kp.auto_reset;
    if (timed && h->ev_n == xarm_handle::NEV) timing_flush(h);
    if (timed) HIPCHK(h, hipEventRecord(h->ev0[h->ev_n], st));
    const bool reach = h->cfg.env_kind == XARM_ENV_REACH, handover = h->cfg.env_kind == XARM_ENV_HANDOVER;
    const bool stack = h->cfg.env_kind == XARM_ENV_STACK_TOWER;
    if (h->kp.auto_reset == XARM_AUTO_RESET_LAZY) {
        k_step_lazy<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev);
        if (timed) { HIPCHK(h, hipEventRecord(h->ev1[h->ev_n], st)); HIPCHK(h, hipEventRecord(h->ev2[h->ev_n], st)); h->ev_n++; }
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    // the call's device-side counters (ended episodes, hand-offs, class histogram): zeroed here, in stream order - the
    // handle keeps no host-side per-step state, so a captured step call replays correctly
    HIPCHK(h, hipMemsetAsync(h->counters, 0, sizeof(int) * ((stack && h->class_key) ? 3 + 2 * xs::NCLS : (h->fast_pipeline ? 3 : 1)), st));
    if (stack) {
        if (h->class_key) {
            const unsigned cg = (unsigned)((h->kp.num_envs + 255) / 256);
            k_class_hist<<<dim3(cg), dim3(256), 0, st>>>(h->class_key, h->kp.num_envs, h->class_hist);
            k_class_place<<<dim3(cg), dim3(256), 0, st>>>(h->class_key, h->kp.num_envs, h->class_hist, h->class_hist + xs::NCLS, h->class_order, WG / 2);
        }
        k_st_step<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                       terminal_obs_dev, h->done_list, cnt, h->class_order, h->class_key);
    }
    else if (handover && h->cfg.num_obj == 2)
        k_ho2_step<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                        terminal_obs_dev, h->done_list, cnt);
    else if (handover && h->fast_pipeline) {
        // as for PickAndPlace below: every env on the pad-free fast lane-pair step, the ones with an active finger-pad row
        // handed off, untouched, to the cooperative rows (lists of at most eject_coop_cap envs) or to k_ho_step (longer)
        pipelined = true;
        const bool stand = h->kp.hcfg.use_stand != 0;
        if (stand) k_ho_step_fast<xh::HandoverStandScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                             terminal_obs_dev, h->done_list, cnt, h->eject_list, h->eject_count);
        else k_ho_step_fast<xh::HandoverScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                  terminal_obs_dev, h->done_list, cnt, h->eject_list, h->eject_count);
        if (overlap) {
            HIPCHK(h, hipEventRecord(h->ev_fork, st));
            HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
            launch_ho_reset(h, 2 * grid, h->done_list, cnt, obs_dev, ag_dev, dg_dev, h->side);
            HIPCHK(h, hipEventRecord(h->ev_join, h->side));
        }
        const int64_t cap = h->kp.num_envs < (int64_t)h->kp.eject_coop_cap ? h->kp.num_envs : (int64_t)h->kp.eject_coop_cap;
        int *list_b = overlap ? h->done_list_b : h->done_list, *cnt_b = overlap ? h->eject_count + 1 : cnt;
        if (stand) k_ho_step_coop_list<xh::HandoverStandScene, false><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(
            h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
        else if (h->ho_force_coupled) k_ho_step_coop_list<xh::HandoverScene, true><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(
            h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
        else k_ho_step_coop_list<xh::HandoverScene, false><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(
            h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
        if (h->kp.num_envs > cap) {
            if (stand) k_ho_step<xh::HandoverStandScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                            terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
            else k_ho_step<xh::HandoverScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                 terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
        }
    }
    else if (handover && h->kp.hcfg.use_stand)
        k_ho_step<xh::HandoverStandScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev,
                                                                           success_dev, terminal_obs_dev, h->done_list, cnt, nullptr, nullptr);
    else if (handover)
        k_ho_step<xh::HandoverScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev,
                                                                      success_dev, terminal_obs_dev, h->done_list, cnt, nullptr, nullptr);
    else if (reach && h->kp.num_envs <= (int64_t)h->coop_step_limit)
        k_reach_step_coop<<<dim3((unsigned)((h->kp.num_envs + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(
            h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, h->done_list, cnt);
    else if (reach)
        k_reach_step<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                      terminal_obs_dev, h->done_list, cnt);
    else if (h->kp.num_envs <= (int64_t)h->coop_step_limit)
        k_step_coop<<<dim3((unsigned)((h->kp.num_envs + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(
            h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, h->done_list, cnt);
    else if (h->fast_pipeline) {
        // every env on the pad-free fast step; the ones with an active finger-pad row are handed off, untouched, to the
        // cooperative kernel (lists of at most eject_coop_cap envs) or to k_step (longer lists) - both launched, the one
        // out of its range exits at once (the count lives on the device).  Episodes that end in the hand-off go to a
        // list of their own (done_list_b): the reset of the ~98 % that ended in k_step_fast need not wait for it.
        pipelined = true;
        k_step_fast<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                     terminal_obs_dev, h->done_list, cnt, h->eject_list, h->eject_count);
        if (overlap) {
            // a reset is six sequential ticks of latency on a few hundred wavefronts (2.9 ms), the hand-off 0.55 ms on a
            // few hundred others: started now on the side stream, the first reset overlaps the hand-off.  The call still
            // waits for the second one - a few dozen envs, but the ones with a finger contact, whose reset carries the pad
            // rows through the homing ticks (worth 0.16 ms per call at 16 384 envs, nothing at 65 536)
            HIPCHK(h, hipEventRecord(h->ev_fork, st));
            HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
            launch_pnp_reset(h, h->done_list, cnt, obs_dev, ag_dev, dg_dev, h->side);
            HIPCHK(h, hipEventRecord(h->ev_join, h->side));
        }
        const int64_t cap = h->kp.num_envs < (int64_t)h->kp.eject_coop_cap ? h->kp.num_envs : (int64_t)h->kp.eject_coop_cap;
        const unsigned cgrid = (unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS) < 1024u ? (unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS) : 1024u;
        // without the side stream (XARM_RESET_OVERLAP=0): one list, one reset after the hand-off
        int *list_b = overlap ? h->done_list_b : h->done_list, *cnt_b = overlap ? h->eject_count + 1 : cnt;
        k_step_coop_list<<<dim3(cgrid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                           terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
        if (h->kp.num_envs > cap)
            k_step<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                    terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
    } else
        k_step<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                terminal_obs_dev, h->done_list, cnt, nullptr, nullptr);
    if (timed) HIPCHK(h, hipEventRecord(h->ev1[h->ev_n], st));
    if (h->kp.auto_reset && pipelined && overlap) {
        if (handover) launch_ho_reset(h, 2 * grid, h->done_list_b, h->eject_count + 1, obs_dev, ag_dev, dg_dev, st);
        else launch_pnp_reset(h, h->done_list_b, h->eject_count + 1, obs_dev, ag_dev, dg_dev, st);
        HIPCHK(h, hipStreamWaitEvent(st, h->ev_join, 0));
    } else if (h->kp.auto_reset) {
        if (reach) launch_reach_reset(h, h->done_list, cnt, obs_dev, ag_dev, dg_dev, st);
        else if (handover) launch_ho_reset(h, 2 * grid, h->done_list, cnt, obs_dev, ag_dev, dg_dev, st);
        else if (stack) k_st_reset<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, h->done_list, cnt, obs_dev, ag_dev, dg_dev, h->class_key);
        else launch_pnp_reset(h, h->done_list, cnt, obs_dev, ag_dev, dg_dev, st);
    }
    if (timed) { HIPCHK(h, hipEventRecord(h->ev2[h->ev_n], st)); h->ev_n++; }
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_compute_reward(xarm_handle *h, const float *ag_dev, const float *g_dev, int64_t n, float *out_dev, void *stream) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (n < 0 || (n > 0 && (!ag_dev || !g_dev || !out_dev))) return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: bad argument");
    if (h->cfg.env_kind == XARM_ENV_STACK_TOWER) {
        if (n == 0) return XARM_OK;
        k_st_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->cfg.reward_type, ag_dev, g_dev, n, out_dev);
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    if (h->cfg.env_kind == XARM_ENV_HANDOVER) {
        if (h->cfg.reward_type == XARM_REWARD_DENSE)
            return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: reward_type 'dense' depends on the grasp flags and gripper positions and cannot be relabelled");
        if (n == 0) return XARM_OK;
        if (h->cfg.num_obj == 2) k_ho2_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(ag_dev, g_dev, n, out_dev);
        else k_ho_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(ag_dev, g_dev, n, out_dev);
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    if (h->cfg.env_kind == XARM_ENV_REACH) {
        if (h->cfg.reward_type == XARM_REACH_REWARD_DENSE_DIFF)
            return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: reward_type 'dense_diff' is stateful (d_old) and cannot be relabelled");
        if (n == 0) return XARM_OK;
        k_reach_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->cfg.reward_type, ag_dev, g_dev, n, out_dev);
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    if (h->cfg.reward_type == XARM_REWARD_DENSE)
        return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: reward_type 'dense' depends on the contact state and cannot be relabelled");
    if (n == 0) return XARM_OK;
    k_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->cfg.reward_type, ag_dev, g_dev, n, out_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_get_state(xarm_handle *h, float *state_dev, void *stream) {
    if (!h || !state_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    const int64_t n = h->kp.num_envs * h->kp.state_dim;
    k_get_state<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->kp, state_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}
int xarm_set_state(xarm_handle *h, const float *state_dev, void *stream) {
    if (!h || !state_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    const int64_t n = h->kp.num_envs * h->kp.state_dim;
    k_set_state<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->kp, state_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_episode_steps(xarm_handle *h, int32_t *steps_dev, void *stream) {
    if (!h || !steps_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    const int field = h->cfg.env_kind == XARM_ENV_REACH ? (int)xr::R_STEPS : (h->cfg.env_kind == XARM_ENV_HANDOVER ? (h->cfg.num_obj == 2 ? (int)xh2::G_STEPS : (int)xh::H_STEPS) :
                      (h->cfg.env_kind == XARM_ENV_STACK_TOWER ? (int)xs::K_STEPS : (int)xk::S_STEPS));
    k_episode_steps<<<dim3((unsigned)((h->kp.num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->kp, field, steps_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_debug_substeps(xarm_handle *h, const float *qtarget_dev, int32_t n, void *stream) {
    if (!h || !qtarget_dev || n < 0) return XARM_E_INVALID;
    DEVGUARD(h);
    if (h->cfg.env_kind != XARM_ENV_PICK_AND_PLACE) return fail(h, XARM_E_INVALID, "%s", "xarm_debug_substeps: PickAndPlace only");
    k_substeps<<<dim3((unsigned)(h->kp.stride / WG)), dim3(WG), 0, (hipStream_t)stream>>>(h->kp, qtarget_dev, n);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_timing_enable(xarm_handle *h, int32_t enable) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (enable && !h->ev_created) {
        for (int i = 0; i < xarm_handle::NEV; i++) {
            HIPCHK(h, hipEventCreate(&h->ev0[i]));
            HIPCHK(h, hipEventCreate(&h->ev1[i]));
            HIPCHK(h, hipEventCreate(&h->ev2[i]));
        }
        h->ev_created = true;
    }
    if (enable) { h->ev_n = 0; h->ev_ms = 0; h->ev_reset_ms = 0; h->ev_launches = 0; }
    h->timing = enable;
    return XARM_OK;
}
int xarm_timing_read(xarm_handle *h, double *ms_total, int64_t *launches) {
    if (!h || !ms_total || !launches) return XARM_E_INVALID;
    DEVGUARD(h);
    timing_flush(h);
    *ms_total = h->ev_ms;
    *launches = h->ev_launches;
    return XARM_OK;
}
int xarm_timing_read_reset(xarm_handle *h, double *reset_ms_total, int64_t *launches) {
    if (!h || !reset_ms_total || !launches) return XARM_E_INVALID;
    DEVGUARD(h);
    timing_flush(h);
    *reset_ms_total = h->ev_reset_ms;
    *launches = h->ev_launches;
    return XARM_OK;
}
int xarm_class_keys(xarm_handle *h, uint8_t *keys_dev, void *stream) {
    if (!h || !keys_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    if (!h->class_key) return fail(h, XARM_E_INVALID, "%s", "xarm_class_keys: StackTower handles with the class order enabled only");
    HIPCHK(h, hipMemcpyAsync(keys_dev, h->class_key, (size_t)h->kp.num_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return XARM_OK;
}
int xarm_kernel_limits(const xarm_handle *h, int32_t *reset_coop_limit, int32_t *step_coop_limit) {
    if (!h || !reset_coop_limit || !step_coop_limit) return XARM_E_INVALID;
    *reset_coop_limit = h->kp.coop_limit;
    *step_coop_limit = h->coop_step_limit;
    return XARM_OK;
}

} // extern "C"
