// xarm_k_reach.hip - XarmReach-v0.
// Part of libxarm_hip.so (gfx950); shared declarations: xarm_dev.h, C ABI: xarm_hip.hip.
#include "xarm_dev.h"

namespace xd {

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

} // namespace xd
