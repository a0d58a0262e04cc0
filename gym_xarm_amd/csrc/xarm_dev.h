// xarm_dev.h - what the kernel translation units of libxarm_hip.so share: the launch parameter block, the LDS accessors, the
// state load / store helpers and the prototypes of every kernel.  The kernels are defined in xarm_k_*.hip (one translation
// unit per kernel family, compiled in parallel by gym_xarm_amd/build.py: the fused kernels take ~30 s each to compile) and
// launched from the C ABI in xarm_hip.hip; HIP host stubs have external linkage, so a kernel is launched from a translation
// unit that only sees its prototype.
#pragma once
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

namespace xd {

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

// The same reset with one environment per DPP row of 16 lanes (xarm_coop_core.h): 4 environments per wavefront, the
// Gauss-Seidel sweep spread over the row.  This is the latency-optimal form for the usual case - a handful to a few
// thousand finished episodes per step - where k_reset would keep one wavefront busy for six sequential ticks while
// the rest of the GPU idles.  Rows beyond the list shadow its last entry (the wavefront stays convergent) and store
// nothing; lane 0 of a row writes the environment back.
constexpr int COOP_ENVS = WG / xc::GL;
constexpr int HO_COOP_LDS_FLOATS = xk::LDS_T;     // Handover cooperative rows: only the joint motion axes S are staged

// A cooperative workgroup owns 4 consecutive envs = 16 B of every state column, an HBM line holds 16-32 envs.  Under the
// default round-robin of workgroups over the 8 XCDs (one private L2 each) every XCD fetched - and partially wrote -
// every line: 5x the algorithmic bytes measured at 4 096 envs (profiles/r02e_reach_pmc_summary.json).  This bijective
// remap (valid for any grid size) gives the workgroups that share an XCD one contiguous env range instead.
__device__ __forceinline__ int64_t xcd_contiguous_block() {
    const unsigned b = blockIdx.x, nwg = gridDim.x, xcd = b & 7u, q = nwg >> 3, r = nwg & 7u;
    return (int64_t)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3));
}

// --------------------------------------------------------------------- XarmHandover-v0 (two lanes per env)
// lane-pair exchange by DPP quad permutes: lanes (2k, 2k+1) are arm 0 / arm 1 of one environment
struct DppXchg {
    __device__ __forceinline__ float from0(float v) const { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xA0, 0xF, 0xF, true)); }   // quad_perm [0,0,2,2]
    __device__ __forceinline__ float from1(float v) const { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xF5, 0xF, 0xF, true)); }   // quad_perm [1,1,3,3]
    __device__ __forceinline__ float partner(float v) const { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); } // quad_perm [1,0,3,2]
};

// One stage of the staged Handover step (xarm_step, xarm_hip.hip): the fast kernel runs the ticks [tick0, tick1) of the step,
// the hand-off kernels the ticks [tick0, HO_N_TICKS).  tick0 == 0 opens the step (xh::step_begin) and leaves the joint targets in
// qt[18][stride] for the later stages; flag[e] != 0: env e was handed off in an earlier stage of this call, the fast kernel
// passes it by.  The unstaged step is {0, HO_N_TICKS, null, null}.
struct HoStage {
    int tick0, tick1;
    float *qt;
    uint8_t *flag;
};
__device__ __forceinline__ void ho_load_qt(const KParams &P, const HoStage &S, int64_t e, int arm, float (&qt)[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) qt[i] = S.qt[(9 * arm + i) * P.stride + e];
}
__device__ __forceinline__ void ho_store_qt(const KParams &P, const HoStage &S, int64_t e, int arm, const float (&qt)[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) S.qt[(9 * arm + i) * P.stride + e] = qt[i];
}
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

// ---------------------------------------------------------------------------------------------
// kernels (definitions: xarm_k_*.hip)
__global__ __launch_bounds__(WG) void k_init(KParams P);
__global__ __launch_bounds__(WG) void k_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                             float *__restrict__ ag_out, float *__restrict__ dg_out,
                                             float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                             uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                             int *__restrict__ done_list, int *__restrict__ done_count,
                                             const int *__restrict__ list, const int *__restrict__ count);
__global__ __launch_bounds__(WG) void k_step_fast(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                  float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                  float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                  uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                  int *__restrict__ done_list, int *__restrict__ done_count,
                                                  int *__restrict__ eject_list, int *__restrict__ eject_count);
__global__ __launch_bounds__(WG) void k_step_lazy(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                  float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                  float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                  uint8_t *__restrict__ succ_out);
__global__ __launch_bounds__(WG) void k_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                              float *__restrict__ obs_out, float *__restrict__ ag_out,
                                              float *__restrict__ dg_out);
__global__ __launch_bounds__(WG) void k_reset_coop(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                   float *__restrict__ obs_out, float *__restrict__ ag_out,
                                                   float *__restrict__ dg_out);
__global__ __launch_bounds__(WG) void k_step_coop(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                  float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                  float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                  uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                  int *__restrict__ done_list, int *__restrict__ done_count);
__global__ __launch_bounds__(WG) void k_step_coop_list(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                       float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                       float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                       uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                       int *__restrict__ done_list, int *__restrict__ done_count,
                                                       const int *__restrict__ list, const int *__restrict__ count);
// the staged PickAndPlace step (xarm_k_pnp.hip, xarm_k_pnp_coop.hip)
__global__ __launch_bounds__(WG) void k_step_fast_stage(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                        float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                        float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                        uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                        int *__restrict__ done_list, int *__restrict__ done_count,
                                                        int *__restrict__ eject_list, int *__restrict__ eject_count, HoStage stage);
__global__ __launch_bounds__(WG) void k_step_from_stage(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                        float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                        float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                        uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                        int *__restrict__ done_list, int *__restrict__ done_count,
                                                        const int *__restrict__ list, const int *__restrict__ count, HoStage stage);
__global__ __launch_bounds__(WG) void k_step_coop_list_stage(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                             float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                             float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                             uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                             int *__restrict__ done_list, int *__restrict__ done_count,
                                                             const int *__restrict__ list, const int *__restrict__ count, HoStage stage);
__global__ __launch_bounds__(WG) void k_substeps(KParams P, const float *__restrict__ qt_in, int n);
__global__ void k_compact_mask(const uint8_t *__restrict__ mask, int64_t n, int *__restrict__ list, int *__restrict__ count);
__global__ void k_get_state(KParams P, float *__restrict__ out);
__global__ void k_set_state(KParams P, const float *__restrict__ in);
__global__ void k_compute_reward(int reward_type, const float *__restrict__ ag, const float *__restrict__ g, int64_t n,
                                 float *__restrict__ out);
__global__ __launch_bounds__(WG) void k_reach_init(KParams P);
__global__ __launch_bounds__(WG) void k_reach_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                   float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                   float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                   uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                   int *__restrict__ done_list, int *__restrict__ done_count);
__global__ __launch_bounds__(WG) void k_reach_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                    float *__restrict__ obs_out, float *__restrict__ ag_out,
                                                    float *__restrict__ dg_out);
__global__ __launch_bounds__(WG) void k_reach_step_coop(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                        float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                        float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                        uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                        int *__restrict__ done_list, int *__restrict__ done_count);
__global__ __launch_bounds__(WG) void k_reach_reset_coop(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                         float *__restrict__ obs_out, float *__restrict__ ag_out,
                                                         float *__restrict__ dg_out);
__global__ void k_reach_compute_reward(int reward_type, const float *__restrict__ ag, const float *__restrict__ g, int64_t n,
                                       float *__restrict__ out);
__global__ __launch_bounds__(WG) void k_ho_init(KParams P);
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                int *__restrict__ done_list, int *__restrict__ done_count,
                                                const int *__restrict__ list, const int *__restrict__ count, HoStage stage);
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                 float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);
__global__ void k_ho_compute_reward(const float *__restrict__ ag, const float *__restrict__ g, int64_t n, float *__restrict__ out);
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho_step_fast(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                     float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                     float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                     uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                     int *__restrict__ done_list, int *__restrict__ done_count,
                                                     int *__restrict__ eject_list, int *__restrict__ eject_count, HoStage stage);
template <typename Scene, bool FORCE_COUPLED>
__global__ __launch_bounds__(WG) void k_ho_step_coop_list(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                          float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                          float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                          uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                          int *__restrict__ done_list, int *__restrict__ done_count,
                                                          const int *__restrict__ list, const int *__restrict__ count, HoStage stage);
template <typename Scene, bool FORCE_COUPLED>
__global__ __launch_bounds__(WG) void k_ho_reset_coop(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                      float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);
__global__ __launch_bounds__(WG) void k_ho2_init(KParams P);
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho2_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                 float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                 float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                 uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                 int *__restrict__ done_list, int *__restrict__ done_count);
template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho2_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                  float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);
__global__ void k_ho2_compute_reward(const float *__restrict__ ag, const float *__restrict__ g, int64_t n, float *__restrict__ out);
__global__ __launch_bounds__(WG) void k_st_init(KParams P);
__global__ void k_class_hist(const uint8_t *__restrict__ key, int64_t n, int *__restrict__ hist);
__global__ void k_class_place(const uint8_t *__restrict__ key, int64_t n, const int *__restrict__ hist, int *__restrict__ cursor,
                              int *__restrict__ order, int group);
__global__ __launch_bounds__(WG) void k_st_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                int *__restrict__ done_list, int *__restrict__ done_count,
                                                const int *__restrict__ order, uint8_t *__restrict__ key);
__global__ __launch_bounds__(WG) void k_st_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                 float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                 uint8_t *__restrict__ key);
__global__ void k_st_compute_reward(int reward_type, const float *__restrict__ ag, const float *__restrict__ g, int64_t n, float *__restrict__ out);
__global__ void k_episode_steps(KParams P, int steps_field, int32_t *__restrict__ out);

} // namespace xd
