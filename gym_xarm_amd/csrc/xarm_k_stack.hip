// xarm_k_stack.hip - XarmPDStackTower-v0.
// Part of libxarm_hip.so (gfx950); shared declarations: xarm_dev.h, C ABI: xarm_hip.hip.
#include "xarm_dev.h"

namespace xd {

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

} // namespace xd
