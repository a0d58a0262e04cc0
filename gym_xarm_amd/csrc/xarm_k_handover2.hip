// xarm_k_handover2.hip - XarmHandover-v0 with num_obj = 2 (the reference's test.py configuration).
// Part of libxarm_hip.so (gfx950); shared declarations: xarm_dev.h, C ABI: xarm_hip.hip.
#include "xarm_dev.h"

namespace xd {

// --------------------------------------------------------------------- XarmHandover-v0, num_obj = 2 (two lanes per env)
// the reference's test.py configuration (test.py:9-15); core xarm_handover2_core.h.
// LDS: the arm's columns S | T | A_hh (117 floats) are per lane; the sticks' contact columns (support slots, stick/stick
// manifold, clipping scratch: 294 floats) are the SAME numbers in an env's two lanes - both compute the object rows
// redundantly and bit-identically, in lockstep - so the two lanes share one copy: 117 x 64 + 294 x 32 floats = 66 KB per
// wavefront, two wavefronts per CU (one column set per lane was 105 KB: one wavefront per CU, a quarter of the SIMDs).
constexpr int H2_OBJ_FLOATS = xh2::LDS_FLOATS - xk::LDS_TBL;
constexpr int H2_LDS_FLOATS = xk::LDS_TBL * WG + H2_OBJ_FLOATS * (WG / 2);
struct Ho2Lds {
    float *arm;   // + lane, column stride WG
    float *obj;   // + lane / 2, column stride WG / 2
    __device__ __forceinline__ float &operator[](int i) const { return i < xk::LDS_TBL ? arm[i * WG] : obj[(i - xk::LDS_TBL) * (WG / 2)]; }
};
__device__ __forceinline__ Ho2Lds ho2_lds(float *smem) { return Ho2Lds{smem + threadIdx.x, smem + xk::LDS_TBL * WG + (threadIdx.x >> 1)}; }
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

template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho2_step(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                 float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                 float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                 uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                 int *__restrict__ done_list, int *__restrict__ done_count) {
    __shared__ float smem[H2_LDS_FLOATS];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, e_in = t >> 1;
    const int arm = (int)(t & 1);
    if (e_in >= P.num_envs) return;
    const Ho2Lds lds = ho2_lds(smem);
    xh2::Lane<float> L;
    h2_load(P, e_in, arm, L);
    const float4 a4 = reinterpret_cast<const float4 *>(actions)[e_in * 2 + arm];
    const float act[4] = {a4.x, a4.y, a4.z, a4.w};
    float reward;
    bool done, success;
    xh2::lane_step<float, Ho2Lds, DppXchg, Scene>(L, arm, act, reward, done, success, lds, DppXchg());
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

template <typename Scene>
__global__ __launch_bounds__(WG) void k_ho2_reset(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                  float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out) {
    __shared__ float smem[H2_LDS_FLOATS];
    const int64_t t = (int64_t)blockIdx.x * WG + threadIdx.x, i = t >> 1;
    const int arm = (int)(t & 1);
    const int64_t n = count ? (int64_t)*count : P.num_envs;
    if (i >= n) return;
    const int64_t e_in = list ? (int64_t)list[i] : i;
    const Ho2Lds lds = ho2_lds(smem);
    xh2::Lane<float> L;
    h2_load(P, e_in, arm, L);
    xh2::lane_reset<float, Ho2Lds, DppXchg, Scene>(P.hcfg, e_in, L, arm, lds, DppXchg());
    const int64_t e = late_index(e_in);
    h2_store(P, e, arm, L);
    if (obs_out) h2_write_obs(L, e, arm, obs_out, ag_out, dg_out);
}

// the two scenes the C ABI selects from (xarm_config.use_stand)
template __global__ void k_ho2_step<xh::HandoverScene>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                 float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                 float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                 uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                 int *__restrict__ done_list, int *__restrict__ done_count);
template __global__ void k_ho2_reset<xh::HandoverScene>(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                  float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);
template __global__ void k_ho2_step<xh::HandoverStandScene>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                 float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                 float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                 uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                 int *__restrict__ done_list, int *__restrict__ done_count);
template __global__ void k_ho2_reset<xh::HandoverStandScene>(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                  float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);

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

} // namespace xd
