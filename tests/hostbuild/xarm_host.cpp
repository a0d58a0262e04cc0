// Host (g++) instantiation of gym_xarm_amd/csrc/xarm_core.h for CPU-side unit tests and
// sanitizer runs ONLY.  It lives under tests/ and is never loaded by the product package:
// gym_xarm_amd binds libxarm_hip.so (HIP, gfx950) or raises.  The point of this build is to
// check the kernel's restructured algorithm (CRBA + Cholesky + operational-space block PGS,
// lane-masked execution) against the oracle in both float and double before any GPU time
// is spent, and to run it under -fsanitize=address,undefined.
#define XARM_HOST_BUILD 1
#include "../../gym_xarm_amd/csrc/xarm_core.h"
#include "../../gym_xarm_amd/csrc/xarm_reach_core.h"
#include <atomic>
#include <sched.h>
#include "../../gym_xarm_amd/csrc/xarm_handover_core.h"
#include "../../gym_xarm_amd/csrc/xarm_handover2_core.h"
#include "../../gym_xarm_amd/csrc/xarm_stack_core.h"
#include "../../gym_xarm_amd/csrc/xarm_coop_core.h"
#include "../../gym_xarm_amd/csrc/xarm_handover_coop_core.h"
#include "../../gym_xarm_amd/csrc/xarm_reach_coop_core.h"
#include <pthread.h>
#include <string.h>

namespace {
template <typename T> struct HostLds {
    T *base;
    T &operator[](int i) const { return base[i]; }
};
template <typename T> void load(const double *row, xk::EnvState<T> &s) {
    for (int i = 0; i < 9; i++) { s.q[i] = (T)row[xk::S_Q + i]; s.qd[i] = (T)row[xk::S_QD + i]; }
    for (int i = 0; i < 3; i++) { s.bp[i] = (T)row[xk::S_BP + i]; s.bv[i] = (T)row[xk::S_BV + i]; s.bw[i] = (T)row[xk::S_BW + i]; s.goal[i] = (T)row[xk::S_GOAL + i]; }
    for (int i = 0; i < 4; i++) s.bq[i] = (T)row[xk::S_BQ + i];
    for (int i = 0; i < 8; i++) { s.lam_t[i] = (T)row[xk::S_LT + i]; s.lam_p[i] = (T)row[xk::S_LP + i]; }
    s.touch = (T)row[xk::S_TOUCH]; s.mug = (T)row[xk::S_MUG]; s.steps = (T)row[xk::S_STEPS]; s.episode = (T)row[xk::S_EPISODE];
}
template <typename T> void store(const xk::EnvState<T> &s, double *row) {
    for (int i = 0; i < 9; i++) { row[xk::S_Q + i] = s.q[i]; row[xk::S_QD + i] = s.qd[i]; }
    for (int i = 0; i < 3; i++) { row[xk::S_BP + i] = s.bp[i]; row[xk::S_BV + i] = s.bv[i]; row[xk::S_BW + i] = s.bw[i]; row[xk::S_GOAL + i] = s.goal[i]; }
    for (int i = 0; i < 4; i++) row[xk::S_BQ + i] = s.bq[i];
    for (int i = 0; i < 8; i++) { row[xk::S_LT + i] = s.lam_t[i]; row[xk::S_LP + i] = s.lam_p[i]; }
    row[xk::S_TOUCH] = s.touch; row[xk::S_MUG] = s.mug; row[xk::S_STEPS] = s.steps; row[xk::S_EPISODE] = s.episode;
}
xk::EnvCfg mkcfg(uint64_t seed, int64_t off, double igr, double ggr, int gs, int rt) {
    xk::EnvCfg c; c.seed = seed; c.env_id_offset = off; c.init_grasp_rate = (float)igr; c.goal_ground_rate = (float)ggr; c.goal_shape = gs; c.reward_type = rt;
    return c;
}
template <typename T>
void do_step(const xk::EnvCfg &cfg, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    for (int64_t e = 0; e < E; e++) {
        xk::EnvState<T> s; T lds[xk::LDS_FLOATS]; HostLds<T> L{lds};
        load(state + e * xk::STATE_DIM, s);
        T a[4], o[xk::OBS_DIM], r; bool d, su;
        for (int k = 0; k < 4; k++) a[k] = (T)act[e * 4 + k];
        xk::env_step<T>(cfg, s, a, o, r, d, su, L);
        store(s, state + e * xk::STATE_DIM);
        for (int k = 0; k < xk::OBS_DIM; k++) obs[e * xk::OBS_DIM + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = s.bp[k]; dg[e * 3 + k] = s.goal[k]; }
        rew[e] = r; done[e] = d; succ[e] = su;
    }
}
// the pad-free fast step (xk::env_step_fast): ok[e] = 0 when a finger-pad row was active (outputs meaningless, state untouched)
template <typename T> void do_step_fast(const xk::EnvCfg &c, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, uint8_t *ok) {
    for (int64_t e = 0; e < E; e++) {
        xk::EnvState<T> s; load(state + e * xk::STATE_DIM, s);
        T lds[xk::LDS_FLOATS]; HostLds<T> hl{lds};
        T a[4], o[xk::OBS_DIM], r; bool d, su;
        for (int k = 0; k < 4; k++) a[k] = (T)act[e * 4 + k];
        ok[e] = xk::env_step_fast<T>(c, s, a, o, r, d, su, hl) ? 1 : 0;
        if (!ok[e]) continue;
        store(s, state + e * xk::STATE_DIM);
        for (int k = 0; k < xk::OBS_DIM; k++) obs[e * xk::OBS_DIM + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = s.bp[k]; dg[e * 3 + k] = s.goal[k]; }
        rew[e] = r; done[e] = d; succ[e] = su;
    }
}
template <typename T>
void do_step_lazy(const xk::EnvCfg &cfg, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    for (int64_t e = 0; e < E; e++) {
        xk::EnvState<T> s; T lds[xk::LDS_FLOATS]; HostLds<T> L{lds};
        load(state + e * xk::STATE_DIM, s);
        T a[4], o[xk::OBS_DIM], r; bool d, su; int phase;
        for (int k = 0; k < 4; k++) a[k] = (T)act[e * 4 + k];
        xk::env_step_lazy<T>(cfg, e, s, a, o, r, d, su, phase, L);
        store(s, state + e * xk::STATE_DIM);
        for (int k = 0; k < xk::OBS_DIM; k++) obs[e * xk::OBS_DIM + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = s.bp[k]; dg[e * 3 + k] = s.goal[k]; }
        rew[e] = r; done[e] = (uint8_t)phase; succ[e] = su;
    }
}
template <typename T>
void do_reset(const xk::EnvCfg &cfg, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        xk::EnvState<T> s; T lds[xk::LDS_FLOATS]; HostLds<T> L{lds};
        load(state + e * xk::STATE_DIM, s);
        xk::env_reset<T>(cfg, e, s, L);
        store(s, state + e * xk::STATE_DIM);
        T o[xk::OBS_DIM];
        xk::get_obs(s, o);
        for (int k = 0; k < xk::OBS_DIM; k++) obs[e * xk::OBS_DIM + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = s.bp[k]; dg[e * 3 + k] = s.goal[k]; }
    }
}
template <typename T> void do_init(const xk::EnvCfg &cfg, int64_t E, double *state) {
    for (int64_t e = 0; e < E; e++) { xk::EnvState<T> s; xk::env_init<T>(cfg, e, s); store(s, state + e * xk::STATE_DIM); }
}
template <typename T> void do_substep(int64_t E, double *state, const double *qt, int n) {
    for (int64_t e = 0; e < E; e++) {
        xk::EnvState<T> s; T lds[xk::LDS_FLOATS]; HostLds<T> L{lds};
        load(state + e * xk::STATE_DIM, s);
        T t[9]; for (int k = 0; k < 9; k++) t[k] = (T)qt[e * 9 + k];
        const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
        for (int k = 0; k < n; k++) xk::substep<T>(s, t, dt, L);
        store(s, state + e * xk::STATE_DIM);
    }
}
// cooperative (16 lanes per env) core: the host LV<T> carries the 16 lane values, so one call runs the whole row
template <typename T>
void coop_step(const xk::EnvCfg &cfg, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    for (int64_t e = 0; e < E; e++) {
        xk::EnvState<T> s; T lds[xk::LDS_FLOATS]; HostLds<T> L{lds};
        load(state + e * xk::STATE_DIM, s);
        T a[4], o[xk::OBS_DIM], r; bool d, su;
        for (int k = 0; k < 4; k++) a[k] = (T)act[e * 4 + k];
        xc::env_step<T>(xc::Grp(), cfg, s, a, o, r, d, su, L);
        store(s, state + e * xk::STATE_DIM);
        for (int k = 0; k < xk::OBS_DIM; k++) obs[e * xk::OBS_DIM + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = s.bp[k]; dg[e * 3 + k] = s.goal[k]; }
        rew[e] = r; done[e] = d; succ[e] = su;
    }
}
template <typename T>
void coop_reset(const xk::EnvCfg &cfg, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        xk::EnvState<T> s; T lds[xk::LDS_FLOATS]; HostLds<T> L{lds};
        load(state + e * xk::STATE_DIM, s);
        xc::env_reset<T>(xc::Grp(), cfg, e, s, L);
        store(s, state + e * xk::STATE_DIM);
        T o[xk::OBS_DIM];
        xk::get_obs(s, o);
        for (int k = 0; k < xk::OBS_DIM; k++) obs[e * xk::OBS_DIM + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = s.bp[k]; dg[e * 3 + k] = s.goal[k]; }
    }
}
template <typename T> void coop_substep(int64_t E, double *state, const double *qt, int n) {
    for (int64_t e = 0; e < E; e++) {
        xk::EnvState<T> s; T lds[xk::LDS_FLOATS]; HostLds<T> L{lds};
        load(state + e * xk::STATE_DIM, s);
        T t[9]; for (int k = 0; k < 9; k++) t[k] = (T)qt[e * 9 + k];
        const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
        const xc::ArmLane<T> C = xc::arm_lane_consts<T>(xc::Grp());
        for (int k = 0; k < n; k++) xc::substep<T>(xc::Grp(), C, s, t, dt, L);
        store(s, state + e * xk::STATE_DIM);
    }
}
template <typename T> void do_ik(const double *q, const double *target, double *out) {
    T qi[9], qo[9]; for (int k = 0; k < 9; k++) qi[k] = (T)q[k];
    xk::ik_solve<T>(qi, xk::mk<T>((T)target[0], (T)target[1], (T)target[2]), qo);
    for (int k = 0; k < 9; k++) out[k] = qo[k];
}
}

namespace {
template <typename T> void rload(const double *r, xr::EnvState<T> &s) {
    for (int i = 0; i < 13; i++) { s.q[i] = (T)r[xr::R_Q + i]; s.qd[i] = (T)r[xr::R_QD + i]; s.qt[i] = (T)r[xr::R_QT + i]; }
    for (int i = 0; i < 3; i++) s.goal[i] = (T)r[xr::R_GOAL + i];
    s.d_old = (T)r[xr::R_DOLD]; s.steps = (T)r[xr::R_STEPS]; s.episode = (T)r[xr::R_EPISODE];
}
template <typename T> void rstore(const xr::EnvState<T> &s, double *r) {
    for (int i = 0; i < 13; i++) { r[xr::R_Q + i] = s.q[i]; r[xr::R_QD + i] = s.qd[i]; r[xr::R_QT + i] = s.qt[i]; }
    for (int i = 0; i < 3; i++) r[xr::R_GOAL + i] = s.goal[i];
    r[xr::R_DOLD] = s.d_old; r[xr::R_STEPS] = s.steps; r[xr::R_EPISODE] = s.episode;
}
template <typename T> void reach_coop_step(const xr::EnvCfg &c, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, int32_t *fut) {
    for (int64_t e = 0; e < E; e++) {
        xr::EnvState<T> s; rload(state + e * xr::STATE_DIM, s);
        T a[4], o[xr::OBS_DIM], r; bool d, su; int f;
        for (int k = 0; k < 4; k++) a[k] = (T)act[e * 4 + k];
        xrc::env_step<T>(xc::Grp(), c, s, a, o, r, d, su, f);
        rstore(s, state + e * xr::STATE_DIM);
        for (int k = 0; k < 8; k++) obs[e * 8 + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = o[k]; dg[e * 3 + k] = s.goal[k]; }
        rew[e] = r; done[e] = d; succ[e] = su; fut[e] = f;
    }
}
template <typename T> void reach_coop_reset(const xr::EnvCfg &c, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        xr::EnvState<T> s; rload(state + e * xr::STATE_DIM, s);
        T o[xr::OBS_DIM];
        xrc::env_reset<T>(xc::Grp(), c, e, s, o);
        rstore(s, state + e * xr::STATE_DIM);
        for (int k = 0; k < 8; k++) obs[e * 8 + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = o[k]; dg[e * 3 + k] = s.goal[k]; }
    }
}
template <typename T> void reach_step(const xr::EnvCfg &c, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, int32_t *fut) {
    for (int64_t e = 0; e < E; e++) {
        xr::EnvState<T> s; rload(state + e * xr::STATE_DIM, s);
        T a[4], o[xr::OBS_DIM], r; bool d, su; int f;
        for (int k = 0; k < 4; k++) a[k] = (T)act[e * 4 + k];
        xr::env_step<T>(c, s, a, o, r, d, su, f);
        rstore(s, state + e * xr::STATE_DIM);
        for (int k = 0; k < 8; k++) obs[e * 8 + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = o[k]; dg[e * 3 + k] = s.goal[k]; }
        rew[e] = r; done[e] = d; succ[e] = su; fut[e] = f;
    }
}
template <typename T> void reach_reset(const xr::EnvCfg &c, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        xr::EnvState<T> s; rload(state + e * xr::STATE_DIM, s);
        T o[xr::OBS_DIM];
        xr::env_reset<T>(c, e, s, o);
        rstore(s, state + e * xr::STATE_DIM);
        for (int k = 0; k < 8; k++) obs[e * 8 + k] = o[k];
        for (int k = 0; k < 3; k++) { ag[e * 3 + k] = o[k]; dg[e * 3 + k] = s.goal[k]; }
    }
}
template <typename T> void reach_init(const xr::EnvCfg &c, int64_t E, double *state) {
    for (int64_t e = 0; e < E; e++) { xr::EnvState<T> s; xr::env_init<T>(c, e, s); rstore(s, state + e * xr::STATE_DIM); }
}
}

// ---- Handover: the two lanes of an environment run as two host threads, the lane-pair exchange is a slot + barrier
namespace {
// two-party sense-reversing spin barrier: the lane pair exchanges thousands of values per tick, and a futex-based
// pthread barrier made the emulation ~50x slower than the arithmetic it synchronises
struct SpinBarrier {
    std::atomic<int> count{0}, gen{0};
    void wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == 2) {
            count.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
        } else {
            int spins = 0;
            while (gen.load(std::memory_order_acquire) == g)
                if (++spins > 4096) { sched_yield(); spins = 0; }
        }
    }
};
struct PairShared { SpinBarrier bar; double slot[2]; };
struct PairXchg {
    PairShared *sh; int arm;
    template <typename T> T get(int src, T v) const {
        sh->slot[arm] = (double)v;
        sh->bar.wait();
        T r = (T)sh->slot[src];
        sh->bar.wait();
        return r;
    }
    template <typename T> T from0(T v) const { return get(0, v); }
    template <typename T> T from1(T v) const { return get(1, v); }
    template <typename T> T partner(T v) const { return get(1 - arm, v); }
};
template <typename T> void hload(const double *r, int arm, xh::Lane<T> &L) {
    for (int i = 0; i < 9; i++) { L.st.q[i] = (T)r[xh::H_Q + 9 * arm + i]; L.st.qd[i] = (T)r[xh::H_QD + 9 * arm + i]; }
    L.ft = (T)r[xh::H_FT + arm];
    for (int i = 0; i < 3; i++) { L.st.bp[i] = (T)r[xh::H_BP + i]; L.st.bv[i] = (T)r[xh::H_BV + i]; L.st.bw[i] = (T)r[xh::H_BW + i]; L.st.goal[i] = (T)r[xh::H_GOAL + i]; }
    for (int i = 0; i < 4; i++) L.st.bq[i] = (T)r[xh::H_BQ + i];
    for (int i = 0; i < 8; i++) { L.st.lam_t[i] = (T)r[xh::H_LT + i]; L.st.lam_p[i] = i < 4 ? (T)r[xh::H_LP + 4 * arm + i] : (T)0; }
    L.st.touch = (T)r[xh::H_TOUCH + arm]; L.st.mug = (T)r[xh::H_MUG + arm]; L.st.steps = (T)r[xh::H_STEPS]; L.st.episode = (T)r[xh::H_EPISODE];
}
template <typename T> void hstore(const xh::Lane<T> &L, int arm, double *r) {
    for (int i = 0; i < 9; i++) { r[xh::H_Q + 9 * arm + i] = L.st.q[i]; r[xh::H_QD + 9 * arm + i] = L.st.qd[i]; }
    r[xh::H_FT + arm] = L.ft;
    for (int i = 0; i < 4; i++) r[xh::H_LP + 4 * arm + i] = L.st.lam_p[i];
    r[xh::H_TOUCH + arm] = L.st.touch; r[xh::H_MUG + arm] = L.st.mug;
    if (arm == 0) {
        for (int i = 0; i < 3; i++) { r[xh::H_BP + i] = L.st.bp[i]; r[xh::H_BV + i] = L.st.bv[i]; r[xh::H_BW + i] = L.st.bw[i]; r[xh::H_GOAL + i] = L.st.goal[i]; }
        for (int i = 0; i < 4; i++) r[xh::H_BQ + i] = L.st.bq[i];
        for (int i = 0; i < 8; i++) r[xh::H_LT + i] = L.st.lam_t[i];
        r[xh::H_STEPS] = L.st.steps; r[xh::H_EPISODE] = L.st.episode;
    }
}
template <typename T> struct HoJob {
    int mode; xh::EnvCfg cfg; int64_t E; double *state; const double *act; const uint8_t *mask;
    double *obs, *ag, *dg, *rew; uint8_t *done, *succ; PairShared *sh; int arm;
};
template <typename T> void *ho_thread(void *p) {
    HoJob<T> &J = *(HoJob<T> *)p;
    PairXchg x{J.sh, J.arm};
    for (int64_t e = 0; e < J.E; e++) {
        if (J.mode == 1 && J.mask && !J.mask[e]) continue;
        xh::Lane<T> L; T lds[xk::LDS_FLOATS]; HostLds<T> hl{lds};
        hload(J.state + e * xh::STATE_DIM, J.arm, L);
        T r = 0; bool d = false, su = false;
        if (J.mode == 0) {
            T a[4]; for (int k = 0; k < 4; k++) a[k] = (T)J.act[e * 8 + 4 * J.arm + k];
            if (J.cfg.use_stand) xh::lane_step<T, HostLds<T>, PairXchg, xh::HandoverStandScene>(L, J.arm, a, r, d, su, hl, x, J.cfg.reward_type);
            else xh::lane_step<T>(L, J.arm, a, r, d, su, hl, x, J.cfg.reward_type);
        } else if (J.cfg.use_stand) xh::lane_reset<T, HostLds<T>, PairXchg, xh::HandoverStandScene>(J.cfg, e, L, J.arm, hl, x);
        else xh::lane_reset<T>(J.cfg, e, L, J.arm, hl, x);
        T o8[8];
        xh::arm_obs(L, J.arm, o8);
        J.sh->bar.wait();   // both lanes finished computing before anyone overwrites the row
        hstore(L, J.arm, J.state + e * xh::STATE_DIM);
        double *o = J.obs + e * xh::OBS_DIM;
        for (int k = 0; k < 8; k++) o[13 + 8 * J.arm + k] = o8[k];
        if (J.arm == 0) {
            for (int k = 0; k < 3; k++) { o[k] = L.st.bp[k]; o[7 + k] = L.st.bv[k]; o[10 + k] = L.st.bw[k]; J.ag[e * 3 + k] = L.st.bp[k]; J.dg[e * 3 + k] = L.st.goal[k]; }
            for (int k = 0; k < 4; k++) o[3 + k] = L.st.bq[k];
            if (J.mode == 0) { J.rew[e] = r; J.done[e] = d; J.succ[e] = su; }
        }
        J.sh->bar.wait();
    }
    return 0;
}
template <typename T> void ho_run(int mode, const xh::EnvCfg &cfg, int64_t E, double *state, const double *act, const uint8_t *mask,
                                  double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    PairShared sh;
    HoJob<T> j[2];
    pthread_t th[2];
    for (int a = 0; a < 2; a++) { j[a] = HoJob<T>{mode, cfg, E, state, act, mask, obs, ag, dg, rew, done, succ, &sh, a}; pthread_create(&th[a], 0, ho_thread<T>, &j[a]); }
    for (int a = 0; a < 2; a++) pthread_join(th[a], 0);
}
}


// ---- Handover, pad-free fast lane-pair step (xh::lane_step_fast) and the cooperative rows (xarm_handover_coop_core.h):
// the two rows of an environment run as two host threads (each carries its 16 lane values in LV<T>), the row exchange
// (v_permlane32_swap on the device) is a 16-wide slot + barrier
namespace {
struct RowShared { SpinBarrier bar; double slot[2][xc::GL]; };
struct RowXchg {
    RowShared *sh; int arm;
    template <typename T> void pair(T v, T &v0, T &v1) const {
        sh->slot[arm][0] = (double)v;
        sh->bar.wait();
        v0 = (T)sh->slot[0][0]; v1 = (T)sh->slot[1][0];
        sh->bar.wait();
    }
    template <typename T> T from0(T v) const { T a, b; pair(v, a, b); return a; }
    template <typename T> T from1(T v) const { T a, b; pair(v, a, b); return b; }
    template <typename T> T partner(T v) const { T a, b; pair(v, a, b); return arm == 0 ? b : a; }
    template <typename T> void both(xc::LV<T> v, xc::LV<T> &v0, xc::LV<T> &v1) const {
        for (int i = 0; i < xc::GL; i++) sh->slot[arm][i] = (double)v.v[i];
        sh->bar.wait();
        for (int i = 0; i < xc::GL; i++) { v0.v[i] = (T)sh->slot[0][i]; v1.v[i] = (T)sh->slot[1][i]; }
        sh->bar.wait();
    }
};
template <typename T> struct HocJob {
    int mode;   // 0 step, 1 reset, 2 step forced through the coupled sweep, 3 reset forced coupled, 4 the fast lane-pair step, 4 + n the staged pipeline with n stages
    xh::EnvCfg cfg; int64_t E; double *state; const double *act; const uint8_t *mask;
    double *obs, *ag, *dg, *rew; uint8_t *done, *succ, *ok; PairShared *psh; RowShared *rsh; int arm;
};
template <typename T, typename Scene> void hoc_env(HocJob<T> &J, int64_t e) {
    RowXchg x{J.rsh, J.arm};
    PairXchg px{J.psh, J.arm};
    const xc::Grp G{};
    xh::Lane<T> L; T lds[xk::LDS_FLOATS]; HostLds<T> hl{lds};
    hload(J.state + e * xh::STATE_DIM, J.arm, L);
    T r = 0; bool d = false, su = false, ok = true;
    int stage = 0;   // staged modes: 1 + the stage that handed the env off
    T a[4] = {0, 0, 0, 0};
    if (J.act) for (int k = 0; k < 4; k++) a[k] = (T)J.act[e * 8 + 4 * J.arm + k];
    if (J.mode == 0) xhc::env_step<T, HostLds<T>, RowXchg, Scene, false>(G, x, L, a, r, d, su, hl, J.cfg.reward_type);
    else if (J.mode == 2) xhc::env_step<T, HostLds<T>, RowXchg, Scene, true>(G, x, L, a, r, d, su, hl, J.cfg.reward_type);
    else if (J.mode == 1) xhc::env_reset<T, HostLds<T>, RowXchg, Scene, false>(G, x, J.cfg, e, L, hl);
    else if (J.mode == 3) xhc::env_reset<T, HostLds<T>, RowXchg, Scene, true>(G, x, J.cfg, e, L, hl);
    else if (J.mode >= 5) {
        // the staged pipeline of xarm_step (xarm_hip.hip) with J.mode - 4 stages: fast ticks stage by stage from the last accepted
        // state; the first stage that reports a pad row is dropped and the cooperative rows run the ticks from its first one on
        const int nst = J.mode - 4, N = xm::HO_N_TICKS;
        T qt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < nst; c++) {
            xh::Lane<T> L2 = L;
            T qt2[9];
            for (int k = 0; k < 9; k++) qt2[k] = qt[k];
            if (xh::lane_step_fast_range<T, HostLds<T>, PairXchg, Scene>(L2, J.arm, a, qt2, c * N / nst, (c + 1) * N / nst, r, d, su, hl, px, J.cfg.reward_type)) {
                L = L2;
                for (int k = 0; k < 9; k++) qt[k] = qt2[k];
            } else {
                xhc::env_step_from<T, HostLds<T>, RowXchg, Scene, false>(G, x, L, a, qt, c * N / nst, r, d, su, hl, J.cfg.reward_type);
                stage = 1 + c;
                break;
            }
        }
    }
    else ok = xh::lane_step_fast<T, HostLds<T>, PairXchg, Scene>(L, J.arm, a, r, d, su, hl, px, J.cfg.reward_type);
    T o8[8];
    xh::arm_obs(L, J.arm, o8);
    J.rsh->bar.wait();   // both rows finished computing before anyone overwrites the state row
    if (ok) {
        hstore(L, J.arm, J.state + e * xh::STATE_DIM);
        double *o = J.obs + e * xh::OBS_DIM;
        for (int k = 0; k < 8; k++) o[13 + 8 * J.arm + k] = o8[k];
        if (J.arm == 0) {
            for (int k = 0; k < 3; k++) { o[k] = L.st.bp[k]; o[7 + k] = L.st.bv[k]; o[10 + k] = L.st.bw[k]; J.ag[e * 3 + k] = L.st.bp[k]; J.dg[e * 3 + k] = L.st.goal[k]; }
            for (int k = 0; k < 4; k++) o[3 + k] = L.st.bq[k];
            if (J.rew) { J.rew[e] = r; J.done[e] = d; J.succ[e] = su; }
        }
    }
    if (J.arm == 0 && J.ok) J.ok[e] = J.mode >= 5 ? (uint8_t)stage : (uint8_t)ok;
    J.rsh->bar.wait();
}
template <typename T> void *hoc_thread(void *p) {
    HocJob<T> &J = *(HocJob<T> *)p;
    for (int64_t e = 0; e < J.E; e++) {
        if ((J.mode == 1 || J.mode == 3) && J.mask && !J.mask[e]) continue;
        if (J.cfg.use_stand) hoc_env<T, xh::HandoverStandScene>(J, e); else hoc_env<T, xh::HandoverScene>(J, e);
    }
    return 0;
}
template <typename T> void hoc_run(int mode, const xh::EnvCfg &cfg, int64_t E, double *state, const double *act, const uint8_t *mask,
                                   double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, uint8_t *ok) {
    PairShared psh; RowShared rsh;
    HocJob<T> j[2];
    pthread_t th[2];
    for (int a = 0; a < 2; a++) { j[a] = HocJob<T>{mode, cfg, E, state, act, mask, obs, ag, dg, rew, done, succ, ok, &psh, &rsh, a}; pthread_create(&th[a], 0, hoc_thread<T>, &j[a]); }
    for (int a = 0; a < 2; a++) pthread_join(th[a], 0);
}
}

extern "C" {
static int g_ho_reward_type = 0, g_ho_use_stand = 0;
void xh_ho_set_reward_type(int rt) { g_ho_reward_type = rt; }   // 0 sparse, 1 the staged dense reward
void xh_ho_set_use_stand(int us) { g_ho_use_stand = us; }
static xh::EnvCfg hcfg(uint64_t seed, int64_t off, double ssr, int gs) { xh::EnvCfg c; c.seed = seed; c.env_id_offset = off; c.same_side_rate = (float)ssr; c.goal_shape = gs; c.reward_type = g_ho_reward_type; c.use_stand = g_ho_use_stand; return c; }
void xh_ho_init(int f32, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state) {
    auto c = hcfg(seed, off, ssr, gs);
    for (int64_t e = 0; e < E; e++) for (int a = 1; a >= 0; a--) {
        if (f32) { xh::Lane<float> L; xh::lane_init<float>(c, e, L); hstore(L, a, state + e * xh::STATE_DIM); }
        else { xh::Lane<double> L; xh::lane_init<double>(c, e, L); hstore(L, a, state + e * xh::STATE_DIM); }
    }
}
void xh_ho_step(int f32, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    auto c = hcfg(seed, off, ssr, gs);
    if (f32) ho_run<float>(0, c, E, state, act, 0, obs, ag, dg, rew, done, succ); else ho_run<double>(0, c, E, state, act, 0, obs, ag, dg, rew, done, succ);
}
void xh_ho_reset(int f32, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = hcfg(seed, off, ssr, gs);
    if (f32) ho_run<float>(1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0); else ho_run<double>(1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0);
}
// mode: 0 cooperative rows, 2 cooperative rows forced through the coupled sweep, 4 the pad-free fast lane-pair step (ok[e] = 0: a pad row was active, row e untouched),
// 4 + n the staged pipeline of xarm_step with n stages (ok[e] = 0: finished on the fast path, 1 + c: handed off in stage c)
void xh_hoc_step(int f32, int mode, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, uint8_t *ok) {
    auto c = hcfg(seed, off, ssr, gs);
    if (f32) hoc_run<float>(mode, c, E, state, act, 0, obs, ag, dg, rew, done, succ, ok); else hoc_run<double>(mode, c, E, state, act, 0, obs, ag, dg, rew, done, succ, ok);
}
void xh_hoc_reset(int f32, int forced, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = hcfg(seed, off, ssr, gs);
    if (f32) hoc_run<float>(forced ? 3 : 1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0, 0); else hoc_run<double>(forced ? 3 : 1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0, 0);
}
static xr::EnvCfg rcfg(uint64_t seed, int64_t off, int rt) { xr::EnvCfg c; c.seed = seed; c.env_id_offset = off; c.reward_type = rt; return c; }
void xh_reach_init(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state) { auto c = rcfg(seed, off, rt); if (f32) reach_init<float>(c, E, state); else reach_init<double>(c, E, state); }
void xh_reach_step(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, int32_t *fut) {
    auto c = rcfg(seed, off, rt);
    if (f32) reach_step<float>(c, E, state, act, obs, ag, dg, rew, done, succ, fut); else reach_step<double>(c, E, state, act, obs, ag, dg, rew, done, succ, fut);
}
void xh_reach_coop_step(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, int32_t *fut) {
    auto c = rcfg(seed, off, rt);
    if (f32) reach_coop_step<float>(c, E, state, act, obs, ag, dg, rew, done, succ, fut); else reach_coop_step<double>(c, E, state, act, obs, ag, dg, rew, done, succ, fut);
}
void xh_reach_coop_reset(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = rcfg(seed, off, rt);
    if (f32) reach_coop_reset<float>(c, E, state, mask, obs, ag, dg); else reach_coop_reset<double>(c, E, state, mask, obs, ag, dg);
}
void xh_reach_reset(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = rcfg(seed, off, rt);
    if (f32) reach_reset<float>(c, E, state, mask, obs, ag, dg); else reach_reset<double>(c, E, state, mask, obs, ag, dg);
}
#define CFGARGS uint64_t seed, int64_t off, double igr, double ggr, int gs, int rt
void xh_init(int f32, CFGARGS, int64_t E, double *state) { auto c = mkcfg(seed, off, igr, ggr, gs, rt); if (f32) do_init<float>(c, E, state); else do_init<double>(c, E, state); }
// lazy auto-reset step: `done` returns the phase (0 ordinary, 1 episode ended in this call, 2 reset tick)
void xh_step_lazy(int f32, CFGARGS, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) do_step_lazy<float>(c, E, state, act, obs, ag, dg, rew, done, succ); else do_step_lazy<double>(c, E, state, act, obs, ag, dg, rew, done, succ);
}
void xh_step(int f32, CFGARGS, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) do_step<float>(c, E, state, act, obs, ag, dg, rew, done, succ); else do_step<double>(c, E, state, act, obs, ag, dg, rew, done, succ);
}
void xh_reset(int f32, CFGARGS, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) do_reset<float>(c, E, state, mask, obs, ag, dg); else do_reset<double>(c, E, state, mask, obs, ag, dg);
}
void xh_step_fast(int f32, CFGARGS, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, uint8_t *ok) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) do_step_fast<float>(c, E, state, act, obs, ag, dg, rew, done, succ, ok); else do_step_fast<double>(c, E, state, act, obs, ag, dg, rew, done, succ, ok);
}
void xh_coop_step(int f32, CFGARGS, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) coop_step<float>(c, E, state, act, obs, ag, dg, rew, done, succ); else coop_step<double>(c, E, state, act, obs, ag, dg, rew, done, succ);
}
void xh_coop_reset(int f32, CFGARGS, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) coop_reset<float>(c, E, state, mask, obs, ag, dg); else coop_reset<double>(c, E, state, mask, obs, ag, dg);
}
void xh_coop_substep(int f32, int64_t E, double *state, const double *qt, int n) { if (f32) coop_substep<float>(E, state, qt, n); else coop_substep<double>(E, state, qt, n); }
void xh_substep(int f32, int64_t E, double *state, const double *qt, int n) { if (f32) do_substep<float>(E, state, qt, n); else do_substep<double>(E, state, qt, n); }
void xh_ik(int f32, const double *q, const double *target, double *out) { if (f32) do_ik<float>(q, target, out); else do_ik<double>(q, target, out); }
}


// ---- StackTower: same two-thread lane-pair emulation as Handover
namespace {
template <typename T> void sload(const double *r, int arm, xs::Lane<T> &L) {
    for (int i = 0; i < 9; i++) { L.q[i] = (T)r[xs::K_Q + 9 * arm + i]; L.qd[i] = (T)r[xs::K_QD + 9 * arm + i]; L.qt[i] = (T)r[xs::K_QT + 9 * arm + i]; }
    for (int o = 0; o < 3; o++) {
        for (int k = 0; k < 3; k++) { L.bp[o][k] = (T)r[xs::K_BP + 3 * o + k]; L.bv[o][k] = (T)r[xs::K_BV + 3 * o + k]; L.bw[o][k] = (T)r[xs::K_BW + 3 * o + k]; L.goal[o][k] = (T)r[xs::K_GOAL + 3 * o + k]; }
        for (int k = 0; k < 4; k++) L.bq[o][k] = (T)r[xs::K_BQ + 4 * o + k];
        for (int k = 0; k < 8; k++) L.lam_t[o][k] = (T)r[xs::K_LT + 8 * o + k];
    }
    for (int k = 0; k < 4; k++) L.lam_p[k] = (T)r[xs::K_LP + 4 * arm + k];
    L.steps = (T)r[xs::K_STEPS]; L.episode = (T)r[xs::K_EPISODE];
}
template <typename T> void sstore(const xs::Lane<T> &L, int arm, double *r) {
    for (int i = 0; i < 9; i++) { r[xs::K_Q + 9 * arm + i] = L.q[i]; r[xs::K_QD + 9 * arm + i] = L.qd[i]; r[xs::K_QT + 9 * arm + i] = L.qt[i]; }
    for (int k = 0; k < 4; k++) r[xs::K_LP + 4 * arm + k] = L.lam_p[k];
    if (arm == 0) {
        for (int o = 0; o < 3; o++) {
            for (int k = 0; k < 3; k++) { r[xs::K_BP + 3 * o + k] = L.bp[o][k]; r[xs::K_BV + 3 * o + k] = L.bv[o][k]; r[xs::K_BW + 3 * o + k] = L.bw[o][k]; r[xs::K_GOAL + 3 * o + k] = L.goal[o][k]; }
            for (int k = 0; k < 4; k++) r[xs::K_BQ + 4 * o + k] = L.bq[o][k];
            for (int k = 0; k < 8; k++) r[xs::K_LT + 8 * o + k] = L.lam_t[o][k];
        }
        r[xs::K_STEPS] = L.steps; r[xs::K_EPISODE] = L.episode;
    }
}
template <typename T> struct StJob {
    int mode; xk::EnvCfg cfg; int64_t E; double *state; const double *act; const uint8_t *mask;
    double *obs, *ag, *dg, *rew; uint8_t *done, *succ; PairShared *sh; int arm;
};
template <typename T> void *st_thread(void *p) {
    StJob<T> &J = *(StJob<T> *)p;
    PairXchg x{J.sh, J.arm};
    for (int64_t e = 0; e < J.E; e++) {
        if (J.mode == 1 && J.mask && !J.mask[e]) continue;
        xs::Lane<T> L; T lds[xs::LDS_FLOATS]; HostLds<T> hl{lds};
        sload(J.state + e * xs::STATE_DIM, J.arm, L);
        T r = 0; bool d = false, su = false;
        if (J.mode == 0) {
            T a[4]; for (int k = 0; k < 4; k++) a[k] = (T)J.act[e * 8 + 4 * J.arm + k];
            xs::lane_step<T>(J.cfg, L, J.arm, a, r, d, su, hl, x);
        } else xs::lane_reset<T>(J.cfg, e, L, J.arm, hl, x);
        T o8[8];
        xs::arm_obs(L, J.arm, o8);
        J.sh->bar.wait();
        sstore(L, J.arm, J.state + e * xs::STATE_DIM);
        double *o = J.obs + e * xs::OBS_DIM;
        for (int k = 0; k < 8; k++) o[39 + 8 * J.arm + k] = o8[k];
        if (J.arm == 0) {
            for (int ob = 0; ob < 3; ob++) {
                for (int k = 0; k < 3; k++) {
                    o[3 * ob + k] = L.bp[ob][k]; o[21 + 3 * ob + k] = L.bv[ob][k]; o[30 + 3 * ob + k] = L.bw[ob][k];
                    J.ag[e * 9 + 3 * ob + k] = L.bp[ob][k]; J.dg[e * 9 + 3 * ob + k] = L.goal[ob][k];
                }
                for (int k = 0; k < 4; k++) o[9 + 4 * ob + k] = L.bq[ob][k];
            }
            if (J.mode == 0) { J.rew[e] = r; J.done[e] = d; J.succ[e] = su; }
        }
        J.sh->bar.wait();
    }
    return 0;
}
template <typename T> void st_run(int mode, const xk::EnvCfg &cfg, int64_t E, double *state, const double *act, const uint8_t *mask,
                                  double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    PairShared sh;
    StJob<T> j[2];
    pthread_t th[2];
    for (int a = 0; a < 2; a++) { j[a] = StJob<T>{mode, cfg, E, state, act, mask, obs, ag, dg, rew, done, succ, &sh, a}; pthread_create(&th[a], 0, st_thread<T>, &j[a]); }
    for (int a = 0; a < 2; a++) pthread_join(th[a], 0);
}
static xk::EnvCfg scfg(uint64_t seed, int64_t off, int rt) { xk::EnvCfg c; memset(&c, 0, sizeof c); c.seed = seed; c.env_id_offset = off; c.reward_type = rt; return c; }
}
extern "C" {
void xh_st_init(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state) {
    auto c = scfg(seed, off, rt);
    for (int64_t e = 0; e < E; e++) for (int a = 1; a >= 0; a--) {
        if (f32) { xs::Lane<float> L; xs::lane_init<float>(c, e, L); sstore(L, a, state + e * xs::STATE_DIM); }
        else { xs::Lane<double> L; xs::lane_init<double>(c, e, L); sstore(L, a, state + e * xs::STATE_DIM); }
    }
}
void xh_st_step(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    auto c = scfg(seed, off, rt);
    if (f32) st_run<float>(0, c, E, state, act, 0, obs, ag, dg, rew, done, succ); else st_run<double>(0, c, E, state, act, 0, obs, ag, dg, rew, done, succ);
}
void xh_st_reset(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = scfg(seed, off, rt);
    if (f32) st_run<float>(1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0); else st_run<double>(1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0);
}
// class-homogeneous visiting order of the StackTower step kernel (xs::class_layout / class_slot, as k_class_hist + k_class_place
// run them, with the envs arriving in index order): order[slot] = env; returns 1 when the classes are wavefront-aligned
int xh_class_order(const uint8_t *key, int64_t n, int group, int32_t *order) {
    int hist[xs::NCLS] = {0}, cursor[xs::NCLS] = {0};
    for (int64_t e = 0; e < n; e++) hist[key[e] & (xs::NCLS - 1)]++;
    xs::ClassLayout Y;
    xs::class_layout(hist, group, Y);
    for (int64_t e = 0; e < n; e++) order[e] = -1;
    for (int64_t e = 0; e < n; e++) {
        const int c = key[e] & (xs::NCLS - 1);
        const int slot = xs::class_slot(Y, c, cursor[c]++);
        if (slot < 0 || slot >= n || order[slot] != -1) return -1;
        order[slot] = (int32_t)e;
    }
    return Y.aligned ? 1 : 0;
}
// cube/cube manifold alone (R row-major, column k = axis k, as the oracle's xo_box_box)
int xh_cube_cube(int f32, const double *pA, const double *RA, const double *pB, const double *RB, double h, double margin, double *pts, double *nrm, double *dist) {
    int n;
    if (f32) {
        xk::V3<float> A[3], B[3], P[4], N; float D[4]; float lds[xs::LDS_FLOATS]; HostLds<float> hl{lds};
        for (int k = 0; k < 3; k++) { A[k] = xk::mk<float>((float)RA[k], (float)RA[3 + k], (float)RA[6 + k]); B[k] = xk::mk<float>((float)RB[k], (float)RB[3 + k], (float)RB[6 + k]); }
        n = xs::cube_cube<float, HostLds<float>>(xk::mk<float>((float)pA[0], (float)pA[1], (float)pA[2]), A, xk::mk<float>((float)pB[0], (float)pB[1], (float)pB[2]), B, (float)h, (float)margin, P, N, D, hl);
        for (int q = 0; q < n; q++) { pts[3 * q] = P[q].x; pts[3 * q + 1] = P[q].y; pts[3 * q + 2] = P[q].z; dist[q] = D[q]; }
        if (n) { nrm[0] = N.x; nrm[1] = N.y; nrm[2] = N.z; }
    } else {
        xk::V3<double> A[3], B[3], P[4], N; double D[4]; double lds[xs::LDS_FLOATS]; HostLds<double> hl{lds};
        for (int k = 0; k < 3; k++) { A[k] = xk::mk<double>(RA[k], RA[3 + k], RA[6 + k]); B[k] = xk::mk<double>(RB[k], RB[3 + k], RB[6 + k]); }
        n = xs::cube_cube<double, HostLds<double>>(xk::mk<double>(pA[0], pA[1], pA[2]), A, xk::mk<double>(pB[0], pB[1], pB[2]), B, h, margin, P, N, D, hl);
        for (int q = 0; q < n; q++) { pts[3 * q] = P[q].x; pts[3 * q + 1] = P[q].y; pts[3 * q + 2] = P[q].z; dist[q] = D[q]; }
        if (n) { nrm[0] = N.x; nrm[1] = N.y; nrm[2] = N.z; }
    }
    return n;
}
}


// ---- Handover with num_obj = 2 (xarm_handover2_core.h): same two-thread lane-pair emulation
namespace {
template <typename T> void h2load(const double *r, int arm, xh2::Lane<T> &L) {
    for (int i = 0; i < 9; i++) { L.q[i] = (T)r[xh2::G_Q + 9 * arm + i]; L.qd[i] = (T)r[xh2::G_QD + 9 * arm + i]; }
    L.ft = (T)r[xh2::G_FT + arm];
    for (int o = 0; o < 2; o++) {
        for (int k = 0; k < 3; k++) { L.bp[o][k] = (T)r[xh2::G_BP + 3 * o + k]; L.bv[o][k] = (T)r[xh2::G_BV + 3 * o + k]; L.bw[o][k] = (T)r[xh2::G_BW + 3 * o + k]; L.goal[o][k] = (T)r[xh2::G_GOAL + 3 * o + k]; }
        for (int k = 0; k < 4; k++) L.bq[o][k] = (T)r[xh2::G_BQ + 4 * o + k];
        for (int k = 0; k < 8; k++) L.lam_t[o][k] = (T)r[xh2::G_LT + 8 * o + k];
    }
    for (int k = 0; k < 4; k++) L.lam_p[k] = (T)r[xh2::G_LP + 4 * arm + k];
    L.touch = (T)r[xh2::G_TOUCH + arm]; L.mug = (T)r[xh2::G_MUG + arm];
    L.steps = (T)r[xh2::G_STEPS]; L.episode = (T)r[xh2::G_EPISODE];
}
template <typename T> void h2store(const xh2::Lane<T> &L, int arm, double *r) {
    for (int i = 0; i < 9; i++) { r[xh2::G_Q + 9 * arm + i] = L.q[i]; r[xh2::G_QD + 9 * arm + i] = L.qd[i]; }
    r[xh2::G_FT + arm] = L.ft;
    for (int k = 0; k < 4; k++) r[xh2::G_LP + 4 * arm + k] = L.lam_p[k];
    r[xh2::G_TOUCH + arm] = L.touch; r[xh2::G_MUG + arm] = L.mug;
    if (arm == 0) {
        for (int o = 0; o < 2; o++) {
            for (int k = 0; k < 3; k++) { r[xh2::G_BP + 3 * o + k] = L.bp[o][k]; r[xh2::G_BV + 3 * o + k] = L.bv[o][k]; r[xh2::G_BW + 3 * o + k] = L.bw[o][k]; r[xh2::G_GOAL + 3 * o + k] = L.goal[o][k]; }
            for (int k = 0; k < 4; k++) r[xh2::G_BQ + 4 * o + k] = L.bq[o][k];
            for (int k = 0; k < 8; k++) r[xh2::G_LT + 8 * o + k] = L.lam_t[o][k];
        }
        r[xh2::G_STEPS] = L.steps; r[xh2::G_EPISODE] = L.episode;
    }
}
template <typename T> struct H2Job {
    int mode; xh::EnvCfg cfg; int64_t E; double *state; const double *act; const uint8_t *mask;
    double *obs, *ag, *dg, *rew; uint8_t *done, *succ; PairShared *sh; int arm;
};
template <typename T> void *h2_thread(void *p) {
    H2Job<T> &J = *(H2Job<T> *)p;
    PairXchg x{J.sh, J.arm};
    for (int64_t e = 0; e < J.E; e++) {
        if (J.mode == 1 && J.mask && !J.mask[e]) continue;
        xh2::Lane<T> L; T lds[xh2::LDS_FLOATS]; HostLds<T> hl{lds};
        h2load(J.state + e * xh2::STATE_DIM, J.arm, L);
        T r = 0; bool d = false, su = false;
        if (J.mode == 0) {
            T a[4]; for (int k = 0; k < 4; k++) a[k] = (T)J.act[e * 8 + 4 * J.arm + k];
            if (J.cfg.use_stand) xh2::lane_step<T, HostLds<T>, PairXchg, xh::HandoverStandScene>(L, J.arm, a, r, d, su, hl, x);
            else xh2::lane_step<T>(L, J.arm, a, r, d, su, hl, x);
        } else if (J.cfg.use_stand) xh2::lane_reset<T, HostLds<T>, PairXchg, xh::HandoverStandScene>(J.cfg, e, L, J.arm, hl, x);
        else xh2::lane_reset<T>(J.cfg, e, L, J.arm, hl, x);
        T o8[8];
        xh2::arm_obs(L, J.arm, o8);
        J.sh->bar.wait();
        h2store(L, J.arm, J.state + e * xh2::STATE_DIM);
        double *o = J.obs + e * xh2::OBS_DIM;
        for (int k = 0; k < 8; k++) o[26 + 8 * J.arm + k] = o8[k];
        if (J.arm == 0) {
            for (int ob = 0; ob < 2; ob++) {
                for (int k = 0; k < 3; k++) {
                    o[3 * ob + k] = L.bp[ob][k]; o[14 + 3 * ob + k] = L.bv[ob][k]; o[20 + 3 * ob + k] = L.bw[ob][k];
                    J.ag[e * 6 + 3 * ob + k] = L.bp[ob][k]; J.dg[e * 6 + 3 * ob + k] = L.goal[ob][k];
                }
                for (int k = 0; k < 4; k++) o[6 + 4 * ob + k] = L.bq[ob][k];
            }
            if (J.mode == 0) { J.rew[e] = r; J.done[e] = d; J.succ[e] = su; }
        }
        J.sh->bar.wait();
    }
    return 0;
}
template <typename T> void h2_run(int mode, const xh::EnvCfg &cfg, int64_t E, double *state, const double *act, const uint8_t *mask,
                                  double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    PairShared sh;
    H2Job<T> j[2];
    pthread_t th[2];
    for (int a = 0; a < 2; a++) { j[a] = H2Job<T>{mode, cfg, E, state, act, mask, obs, ag, dg, rew, done, succ, &sh, a}; pthread_create(&th[a], 0, h2_thread<T>, &j[a]); }
    for (int a = 0; a < 2; a++) pthread_join(th[a], 0);
}
}
extern "C" {
void xh_ho2_init(int f32, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state) {
    auto c = hcfg(seed, off, ssr, gs);
    for (int64_t e = 0; e < E; e++) for (int a = 1; a >= 0; a--) {
        if (f32) { xh2::Lane<float> L; xh2::lane_init<float>(c, e, L); h2store(L, a, state + e * xh2::STATE_DIM); }
        else { xh2::Lane<double> L; xh2::lane_init<double>(c, e, L); h2store(L, a, state + e * xh2::STATE_DIM); }
    }
}
void xh_ho2_step(int f32, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    auto c = hcfg(seed, off, ssr, gs);
    if (f32) h2_run<float>(0, c, E, state, act, 0, obs, ag, dg, rew, done, succ); else h2_run<double>(0, c, E, state, act, 0, obs, ag, dg, rew, done, succ);
}
void xh_ho2_reset(int f32, uint64_t seed, int64_t off, double ssr, int gs, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = hcfg(seed, off, ssr, gs);
    if (f32) h2_run<float>(1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0); else h2_run<double>(1, c, E, state, 0, mask, obs, ag, dg, 0, 0, 0);
}
// box/box manifold alone with per-box half extents (R row-major, column k = axis k, as the oracle's xo_box_box)
int xh_box_box(int f32, const double *pA, const double *RA, const double *hA, const double *pB, const double *RB, const double *hB, double margin, double *pts, double *nrm, double *dist) {
    int n;
    if (f32) {
        xk::V3<float> A[3], B[3], P[4], N; float D[4]; float lds[xh2::LDS_FLOATS]; HostLds<float> hl{lds};
        const float ha[3] = {(float)hA[0], (float)hA[1], (float)hA[2]}, hb[3] = {(float)hB[0], (float)hB[1], (float)hB[2]};
        for (int k = 0; k < 3; k++) { A[k] = xk::mk<float>((float)RA[k], (float)RA[3 + k], (float)RA[6 + k]); B[k] = xk::mk<float>((float)RB[k], (float)RB[3 + k], (float)RB[6 + k]); }
        n = xs::box_box<float, HostLds<float>, xh2::LDS_CLIP>(xk::mk<float>((float)pA[0], (float)pA[1], (float)pA[2]), A, ha, xk::mk<float>((float)pB[0], (float)pB[1], (float)pB[2]), B, hb, (float)margin, P, N, D, hl);
        for (int q = 0; q < n; q++) { pts[3 * q] = P[q].x; pts[3 * q + 1] = P[q].y; pts[3 * q + 2] = P[q].z; dist[q] = D[q]; }
        if (n) { nrm[0] = N.x; nrm[1] = N.y; nrm[2] = N.z; }
    } else {
        xk::V3<double> A[3], B[3], P[4], N; double D[4]; double lds[xh2::LDS_FLOATS]; HostLds<double> hl{lds};
        const double ha[3] = {hA[0], hA[1], hA[2]}, hb[3] = {hB[0], hB[1], hB[2]};
        for (int k = 0; k < 3; k++) { A[k] = xk::mk<double>(RA[k], RA[3 + k], RA[6 + k]); B[k] = xk::mk<double>(RB[k], RB[3 + k], RB[6 + k]); }
        n = xs::box_box<double, HostLds<double>, xh2::LDS_CLIP>(xk::mk<double>(pA[0], pA[1], pA[2]), A, ha, xk::mk<double>(pB[0], pB[1], pB[2]), B, hb, margin, P, N, D, hl);
        for (int q = 0; q < n; q++) { pts[3 * q] = P[q].x; pts[3 * q + 1] = P[q].y; pts[3 * q + 2] = P[q].z; dist[q] = D[q]; }
        if (n) { nrm[0] = N.x; nrm[1] = N.y; nrm[2] = N.z; }
    }
    return n;
}
}
