// xarm_k_pnp_coop.hip - PickAndPlace on the cooperative core (16 lanes per environment): k_reset_coop / k_step_coop / k_step_coop_list.
// Part of libxarm_hip.so (gfx950); shared declarations: xarm_dev.h, C ABI: xarm_hip.hip.
#include "xarm_dev.h"

namespace xd {

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

// the hand-off of a staged step: substeps [stage.tick0, 15) of the envs list[0 .. *count) on the cooperative core
__global__ __launch_bounds__(WG) void k_step_coop_list_stage(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                             float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                             float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                             uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                             int *__restrict__ done_list, int *__restrict__ done_count,
                                                             const int *__restrict__ list, const int *__restrict__ count, HoStage stage) {
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
        float obs[xk::OBS_DIM], reward, qt[9];
        bool done, success;
        if (stage.tick0 > 0) ho_load_qt(P, stage, e_in, 0, qt);
        xc::env_step_from<float, DevLds>(G, P.cfg, s, act, qt, stage.tick0, obs, reward, done, success, lds);
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

} // namespace xd
