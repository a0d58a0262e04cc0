// xarm_k_handover_coop.hip - XarmHandover-v0 (one stick): the cooperative rows (two 16-lane rows per environment): the hand-off of k_ho_step_fast (xarm_k_handover.hip) and the reset.
// Part of libxarm_hip.so (gfx950); shared declarations: xarm_dev.h, C ABI: xarm_hip.hip.
#include "xarm_dev.h"

namespace xd {

// Row exchange of the cooperative Handover kernels (xarm_handover_coop_core.h): an environment owns two DPP rows of one
// wavefront, rows 0 / 1 = arm 0 of env slots 0 / 1, rows 2 / 3 = arm 1, i.e. lane l and lane l + 32 are the same lane of the
// two arms of one environment and ONE v_permlane32_swap_b32 (gfx950) hands a register across in both directions.
struct SwapXchg {
    int arm;
    __device__ __forceinline__ void pair(float v, float &v0, float &v1) const {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        unsigned r0 = r[0], r1 = r[1];
        // HAZARD (gfx950, hipcc 7.2): a VALU instruction that reads a register v_permlane32_swap_b32 has just written gets
        // STALE data when it issues right behind the swap - the compiler pads the other direction (VALU write -> swap read,
        // s_nop 1) and nothing here.  Found as run-to-run differences of the cooperative reset (a v_fmac reading the swapped
        // impulse one instruction after the swap; tools/ho_reset_probe.py: 281 of 3 277 envs wrong without the pad, 0 with it).
        // The asm ties both results, so every consumer sits behind the two wait states.
        asm volatile("s_nop 1" : "+v"(r0), "+v"(r1));
        v0 = __uint_as_float(r0);   // lanes 0-31 (arm 0) everywhere
        v1 = __uint_as_float(r1);   // lanes 32-63 (arm 1) everywhere
    }
    __device__ __forceinline__ float from0(float v) const { float a, b; pair(v, a, b); return a; }
    __device__ __forceinline__ float from1(float v) const { float a, b; pair(v, a, b); return b; }
    __device__ __forceinline__ float partner(float v) const { float a, b; pair(v, a, b); return arm == 0 ? b : a; }
    __device__ __forceinline__ void both(xc::LV<float> v, xc::LV<float> &v0, xc::LV<float> &v1) const { pair(v.v[0], v0.v[0], v1.v[0]); }
};

// XarmHandover.step of the envs list[0 .. *count) (null: all) on the cooperative rows: the hand-off of k_ho_step_fast.  The grid
// is fixed (the count lives on the device); a workgroup walks the list with a grid stride, two envs per wavefront.  Lists
// longer than P.eject_coop_cap belong to k_ho_step (launched beside this kernel; exactly one of the two does the work).
template <typename Scene, bool FORCE_COUPLED>
__global__ __launch_bounds__(WG) void k_ho_step_coop_list(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                          float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                          float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                          uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                          int *__restrict__ done_list, int *__restrict__ done_count,
                                                          const int *__restrict__ list, const int *__restrict__ count, HoStage stage) {
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
        float qt[9];
        if (stage.tick0 > 0) ho_load_qt(P, stage, e_in, x.arm, qt);   // the step was opened by an earlier fast stage
        xhc::env_step_from<float, DevLds, SwapXchg, Scene, FORCE_COUPLED>(G, x, L, act, qt, stage.tick0, reward, done, success, lds, P.hcfg.reward_type);
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

template __global__ void k_ho_step_coop_list<xh::HandoverScene, false>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                          float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                          float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                          uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                          int *__restrict__ done_list, int *__restrict__ done_count,
                                                          const int *__restrict__ list, const int *__restrict__ count, HoStage stage);
template __global__ void k_ho_step_coop_list<xh::HandoverScene, true>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                          float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                          float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                          uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                          int *__restrict__ done_list, int *__restrict__ done_count,
                                                          const int *__restrict__ list, const int *__restrict__ count, HoStage stage);
template __global__ void k_ho_step_coop_list<xh::HandoverStandScene, false>(KParams P, const float *__restrict__ actions, float *__restrict__ obs_out,
                                                          float *__restrict__ ag_out, float *__restrict__ dg_out,
                                                          float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                          uint8_t *__restrict__ succ_out, float *__restrict__ term_obs,
                                                          int *__restrict__ done_list, int *__restrict__ done_count,
                                                          const int *__restrict__ list, const int *__restrict__ count, HoStage stage);

// XarmHandover.reset on the cooperative rows for the envs list[0 .. *count) (null: all), counts up to P.coop_limit (more:
// k_ho_reset, launched beside this kernel): six ticks of latency for the handful of envs that finish in a step
template <typename Scene, bool FORCE_COUPLED>
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
        xhc::env_reset<float, DevLds, SwapXchg, Scene, FORCE_COUPLED>(G, x, P.hcfg, e_in, L, lds);
        if (live && G.l == 0) {
            const int64_t e = late_index(e_in);
            ho_store(P, e, x.arm, L);
            if (obs_out) ho_write_obs(L, e, x.arm, obs_out, ag_out, dg_out);
        }
    }
}

template __global__ void k_ho_reset_coop<xh::HandoverScene, false>(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                      float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);
template __global__ void k_ho_reset_coop<xh::HandoverScene, true>(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                      float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);
template __global__ void k_ho_reset_coop<xh::HandoverStandScene, false>(KParams P, const int *__restrict__ list, const int *__restrict__ count,
                                                      float *__restrict__ obs_out, float *__restrict__ ag_out, float *__restrict__ dg_out);

} // namespace xd
