// Host (g++) instantiation of gym_xarm_amd/csrc/xarm_core.h for CPU-side unit tests and
// sanitizer runs ONLY.  It lives under tests/ and is never loaded by the product package:
// gym_xarm_amd binds libxarm_hip.so (HIP, gfx950) or raises.  The point of this build is to
// check the kernel's restructured algorithm (CRBA + Cholesky + operational-space block PGS,
// lane-masked execution) against the oracle in both float and double before any GPU time
// is spent, and to run it under -fsanitize=address,undefined.
#define XARM_HOST_BUILD 1
#include "../../gym_xarm_amd/csrc/xarm_core.h"
#include "../../gym_xarm_amd/csrc/xarm_reach_core.h"
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

extern "C" {
static xr::EnvCfg rcfg(uint64_t seed, int64_t off, int rt) { xr::EnvCfg c; c.seed = seed; c.env_id_offset = off; c.reward_type = rt; return c; }
void xh_reach_init(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state) { auto c = rcfg(seed, off, rt); if (f32) reach_init<float>(c, E, state); else reach_init<double>(c, E, state); }
void xh_reach_step(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ, int32_t *fut) {
    auto c = rcfg(seed, off, rt);
    if (f32) reach_step<float>(c, E, state, act, obs, ag, dg, rew, done, succ, fut); else reach_step<double>(c, E, state, act, obs, ag, dg, rew, done, succ, fut);
}
void xh_reach_reset(int f32, uint64_t seed, int64_t off, int rt, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = rcfg(seed, off, rt);
    if (f32) reach_reset<float>(c, E, state, mask, obs, ag, dg); else reach_reset<double>(c, E, state, mask, obs, ag, dg);
}
#define CFGARGS uint64_t seed, int64_t off, double igr, double ggr, int gs, int rt
void xh_init(int f32, CFGARGS, int64_t E, double *state) { auto c = mkcfg(seed, off, igr, ggr, gs, rt); if (f32) do_init<float>(c, E, state); else do_init<double>(c, E, state); }
void xh_step(int f32, CFGARGS, int64_t E, double *state, const double *act, double *obs, double *ag, double *dg, double *rew, uint8_t *done, uint8_t *succ) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) do_step<float>(c, E, state, act, obs, ag, dg, rew, done, succ); else do_step<double>(c, E, state, act, obs, ag, dg, rew, done, succ);
}
void xh_reset(int f32, CFGARGS, int64_t E, double *state, const uint8_t *mask, double *obs, double *ag, double *dg) {
    auto c = mkcfg(seed, off, igr, ggr, gs, rt);
    if (f32) do_reset<float>(c, E, state, mask, obs, ag, dg); else do_reset<double>(c, E, state, mask, obs, ag, dg);
}
void xh_substep(int f32, int64_t E, double *state, const double *qt, int n) { if (f32) do_substep<float>(E, state, qt, n); else do_substep<double>(E, state, qt, n); }
void xh_ik(int f32, const double *q, const double *target, double *out) { if (f32) do_ik<float>(q, target, out); else do_ik<double>(q, target, out); }
}
