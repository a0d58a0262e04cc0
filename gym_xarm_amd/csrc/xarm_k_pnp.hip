// xarm_k_pnp.hip - PickAndPlace, one environment per lane (k_step / k_step_fast / k_step_lazy / k_reset) and the small generic kernels.
// Part of libxarm_hip.so (gfx950); shared declarations: xarm_dev.h, C ABI: xarm_hip.hip.
#include "xarm_dev.h"

namespace xd {

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

// ---- the STAGED step (XARM_PNP_STAGES > 1; xarm_step): stage kernels beside the unstaged ones above, which stay as they are.
// qt[9][stride]: the joint targets the opening stage computed; flag[e] != 0: env e was handed off in an earlier stage of this call.
__device__ __forceinline__ void pnp_finish(const KParams &P, int64_t e, const xk::EnvState<float> &s, const float (&obs)[xk::OBS_DIM], float reward,
                                           bool done, bool success, float *obs_out, float *ag_out, float *dg_out, float *rew_out, uint8_t *done_out,
                                           uint8_t *succ_out, float *term_obs, int *done_list, int *done_count) {
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
// substeps [stage.tick0, stage.tick1) of the step on the pad-free fast substep; an env with an active pad row in them stores nothing,
// is flagged and appended to eject_list (its hand-off re-runs the substeps from stage.tick0 on)
__global__ __launch_bounds__(WG) void k_step_fast_stage(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                        float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                        float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                        uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                        int *__restrict__ done_list, int *__restrict__ done_count,
                                                        int *__restrict__ eject_list, int *__restrict__ eject_count, HoStage stage) {
    __shared__ float smem[FAST_LDS_FLOATS * WG];
    const int64_t e_in = (int64_t)blockIdx.x * WG + threadIdx.x;
    if (e_in >= P.num_envs) return;
    if (stage.tick0 > 0 && stage.flag[e_in]) return;
    FastLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xk::OBS_DIM], reward, qt[9];
    bool done, success;
    if (stage.tick0 > 0) ho_load_qt(P, stage, e_in, 0, qt);
    const bool ok = xk::env_step_fast_range<float, FastLds>(P.cfg, s, act, qt, stage.tick0, stage.tick1, obs, reward, done, success, lds);
    const int64_t e = late_index(e_in);
    if (!ok) {
        const int pos = atomicAdd(eject_count, 1);
        eject_list[pos] = (int)e;
        stage.flag[e] = 1;
        return;
    }
    store_state(P, e, s);
    if (stage.tick1 < xm::PNP_N_SUBSTEPS) {
        if (stage.tick0 == 0) { ho_store_qt(P, stage, e, 0, qt); stage.flag[e] = 0; }
        return;
    }
    pnp_finish(P, e, s, obs, reward, done, success, obs_out, ag_out, dg_out, rew_out, done_out, succ_out, term_obs, done_list, done_count);
}
// the long-list fall-back of a staged hand-off: substeps [stage.tick0, 15) of the envs list[0 .. *count) on the full one-env-per-lane substep
__global__ __launch_bounds__(WG) void k_step_from_stage(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                        float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                        float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                        uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                        int *__restrict__ done_list, int *__restrict__ done_count,
                                                        const int *__restrict__ list, const int *__restrict__ count, HoStage stage) {
    __shared__ float smem[xk::LDS_FLOATS * WG];
    const int64_t i_in = (int64_t)blockIdx.x * WG + threadIdx.x;
    const int64_t n = (int64_t)*count;
    if (n <= P.eject_coop_cap) return;     // k_step_coop_list_stage's range
    if (i_in >= n) return;
    const int64_t e_in = (int64_t)list[i_in];
    DevLds lds{smem + threadIdx.x};
    xk::EnvState<float> s;
    load_state(P, e_in, s);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float obs[xk::OBS_DIM], reward, qt[9];
    bool done, success;
    if (stage.tick0 > 0) ho_load_qt(P, stage, e_in, 0, qt);
    xk::env_step_from<float, DevLds>(P.cfg, s, act, qt, stage.tick0, obs, reward, done, success, lds);
    const int64_t e = late_index(e_in);
    store_state(P, e, s);
    pnp_finish(P, e, s, obs, reward, done, success, obs_out, ag_out, dg_out, rew_out, done_out, succ_out, term_obs, done_list, done_count);
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

// number of steps taken in the current episode (info['future_length'] = max_episode_steps - steps, :90)
__global__ void k_episode_steps(KParams P, int steps_field, int32_t *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.num_envs) return;
    out[e] = (int32_t)P.state[(int64_t)steps_field * P.stride + e];
}

} // namespace xd
