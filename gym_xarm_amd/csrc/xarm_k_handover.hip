// xarm_k_handover.hip - XarmHandover-v0 (one stick), two lanes per environment: k_ho_step / k_ho_reset and the pad-free fast step k_ho_step_fast.
// Part of libxarm_hip.so (gfx950); shared declarations: xarm_dev.h, C ABI: xarm_hip.hip.
#include "xarm_dev.h"

namespace xd {

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
                                                const int *__restrict__ list, const int *__restrict__ count, HoStage stage) {
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
    float qt[9];
    if (stage.tick0 > 0) ho_load_qt(P, stage, e_in, arm, qt);   // the step was opened by an earlier fast stage
    xh::lane_step_range<float, DevLds, DppXchg, Scene>(L, arm, act, qt, stage.tick0, xm::HO_N_TICKS, reward, done, success, lds, DppXchg(), P.hcfg.reward_type);
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

template __global__ void k_ho_step<xh::HandoverScene>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                int *__restrict__ done_list, int *__restrict__ done_count,
                                                const int *__restrict__ list, const int *__restrict__ count, HoStage stage);
template __global__ void k_ho_step<xh::HandoverStandScene>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                int *__restrict__ done_list, int *__restrict__ done_count,
                                                const int *__restrict__ list, const int *__restrict__ count, HoStage stage);

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

template __global__ void k_ho_reset<xh::HandoverScene>(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                 float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);
template __global__ void k_ho_reset<xh::HandoverStandScene>(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                 float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);

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
                                                     int *__restrict__ eject_list, int *__restrict__ eject_count, HoStage stage) {
    __shared__ float smem[FAST_LDS_FLOATS * WG];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, e_in = t >> 1;
    const int arm = (int)(t & 1);
    if (e_in >= P.num_envs) return;
    if (stage.tick0 > 0 && stage.flag[e_in]) return;   // handed off in an earlier stage of this call
    FastLds lds{smem + threadIdx.x};
    xh::Lane<float> L;
    ho_load(P, e_in, arm, L);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in * 2 + arm];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float reward;
    bool done, success;
    float qt[9];
    if (stage.tick0 > 0) ho_load_qt(P, stage, e_in, arm, qt);
    const bool ok = xh::lane_step_fast_range<float, FastLds, DppXchg, Scene>(L, arm, act, qt, stage.tick0, stage.tick1, reward, done, success, lds,
                                                                            DppXchg(), P.hcfg.reward_type);
    const int64_t e = late_index(e_in);
    if (!ok) {
        if (arm == 0) {
            const int pos = atomicAdd(eject_count, 1);
            eject_list[pos] = (int)e;
            if (stage.flag) stage.flag[e] = 1;
        }
        return;
    }
    ho_store(P, e, arm, L);
    if (stage.tick1 < xm::HO_N_TICKS) {   // the step goes on in the next stage
        if (stage.tick0 == 0) {
            ho_store_qt(P, stage, e, arm, qt);
            if (arm == 0) stage.flag[e] = 0;
        }
        return;
    }
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

template __global__ void k_ho_step_fast<xh::HandoverScene>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                     float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                     float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                     uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                     int *__restrict__ done_list, int *__restrict__ done_count,
                                                     int *__restrict__ eject_list, int *__restrict__ eject_count, HoStage stage);
template __global__ void k_ho_step_fast<xh::HandoverStandScene>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                     float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                     float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                     uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                     int *__restrict__ done_list, int *__restrict__ done_count,
                                                     int *__restrict__ eject_list, int *__restrict__ eject_count, HoStage stage);

} // namespace xd
