// xarm_reach_core.h - per-environment core of XarmReach-v0 (contact-free reach with the xArm gripper).
//
// Reference: /root/reference/gym_xarm/envs/xarm_reach.py
//   env_step  = XarmReachEnv.step          :81-94   (+ _set_action :131-142, _get_obs :144-161)
//   env_reset = XarmReachEnv.reset         :96-102  (+ _reset_sim :163-168, _sample_goal :170-173)
//   reward    = XarmReachEnv.compute_reward :107-116 (sparse / dense / dense_diff)
//   sim_tick  = p.stepSimulation(), numSubSteps = 20, timeStep 1/240 (:16-17,51,86)
// Model: urdf/xarm7.urdf = the same 7 arm joints as xarm7_pd plus the 6-joint xArm gripper tree
// (mimic tags are ignored by Bullet, so the six joints are independent, all about +-x of the hand).
//
// 13-dof fixed-base tree, no contacts: world-frame RNEA/CRBA -> 13x13 Cholesky -> M^-1, solver rows
// = 13 POSITION_CONTROL motors + the joint limits inside the limit window, 50 PGS sweeps, 20 substeps.
// Same thread-per-env / fully unrolled style as xarm_core.h; no LDS is needed (no contact blocks).
#pragma once
#include "xarm_core.h"
#include "xarm7_reach_model.h"

namespace xr {
using xk::V3;
using xk::mk;
using xk::dot;
using xk::cross;
using xk::SV;
using xk::RBI;
using xk::rbi_mul;
using xk::sdot;
using xk::tri;
using xk::symi;
using xk::Frame;
using xk::clampT;

constexpr int ND = 13;
constexpr int STATE_DIM = 45; // q[13] qd[13] motor_target[13] goal[3] d_old num_steps episode
constexpr int OBS_DIM = 8;
enum { R_Q = 0, R_QD = 13, R_QT = 26, R_GOAL = 39, R_DOLD = 42, R_STEPS = 43, R_EPISODE = 44 };

template <typename T> struct EnvState {
    T q[ND], qd[ND], qt[ND];
    T goal[3];
    T d_old, steps, episode;
};
struct EnvCfg {
    uint64_t seed;
    int64_t env_id_offset;
    int reward_type; // 0 sparse, 1 dense, 2 dense_diff
};

template <typename T> XARM_HD void set_body(RBI<T> &I, T m, V3<T> c0, V3<T> c1, V3<T> c2, V3<T> c, const double (&in)[6]) {
    const T ixx = (T)in[0], ixy = (T)in[1], ixz = (T)in[2], iyy = (T)in[3], iyz = (T)in[4], izz = (T)in[5];
    V3<T> m0 = c0 * ixx + c1 * ixy + c2 * ixz;
    V3<T> m1 = c0 * ixy + c1 * iyy + c2 * iyz;
    V3<T> m2 = c0 * ixz + c1 * iyz + c2 * izz;
    const T cc = dot(c, c);
    I.m = m;
    I.h = c * m;
    I.I[0] = m0.x * c0.x + m1.x * c1.x + m2.x * c2.x + m * (cc - c.x * c.x);
    I.I[1] = m0.x * c0.y + m1.x * c1.y + m2.x * c2.y - m * c.x * c.y;
    I.I[2] = m0.x * c0.z + m1.x * c1.z + m2.x * c2.z - m * c.x * c.z;
    I.I[3] = m0.y * c0.y + m1.y * c1.y + m2.y * c2.y + m * (cc - c.y * c.y);
    I.I[4] = m0.y * c0.z + m1.y * c1.z + m2.y * c2.z - m * c.y * c.z;
    I.I[5] = m0.z * c0.z + m1.z * c1.z + m2.z * c2.z + m * (cc - c.z * c.z);
}
template <typename T> XARM_HD SV<T> bias_force(const RBI<T> &I, SV<T> v, SV<T> a) {
    SV<T> Iv = rbi_mul(I, v), Ia = rbi_mul(I, a), f;
    f.w = Ia.w + cross(v.w, Iv.w) + cross(v.v, Iv.v);
    f.v = Ia.v + cross(v.w, Iv.v);
    return f;
}
template <typename T> XARM_HD void rbi_add(RBI<T> &a, const RBI<T> &b) {
    a.m += b.m;
    a.h = a.h + b.h;
#pragma unroll
    for (int e = 0; e < 6; e++) a.I[e] += b.I[e];
}

// one internal substep, dt = timeStep / numSubSteps = 1/4800 s
template <typename T> XARM_HD void substep(EnvState<T> &st, const T dt) {
    const T idt = (T)1 / dt;
    SV<T> S[ND];
    RBI<T> Ib[ND];
    SV<T> fb[ND];
    Frame<T> f = xk::frame_identity<T>();
    SV<T> vel, acc;
    vel.w = mk<T>(0, 0, 0); vel.v = mk<T>(0, 0, 0);
    acc.w = mk<T>(0, 0, 0); acc.v = mk<T>(0, 0, (T)xm::GRAVITY);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        xk::fk_advance(f, i, st.q[i]);
        S[i].w = f.c2;
        S[i].v = cross(f.o, f.c2);
        const T qd = st.qd[i];
        acc.w = acc.w + cross(vel.w, S[i].w) * qd;
        acc.v = acc.v + (cross(vel.w, S[i].v) + cross(vel.v, S[i].w)) * qd;
        vel.w = vel.w + S[i].w * qd;
        vel.v = vel.v + S[i].v * qd;
        if (i < 6) {
            V3<T> c = f.o + f.c0 * (T)xm::COM[i][0] + f.c1 * (T)xm::COM[i][1] + f.c2 * (T)xm::COM[i][2];
            set_body(Ib[i], (T)xm::MASS[i], f.c0, f.c1, f.c2, c, xm::INERTIA[i]);
        } else { // wrist composite: link7 + link_eef + gripper base + link_tcp
            V3<T> c = f.o + f.c0 * (T)xmr::WRIST_COM[0] + f.c1 * (T)xmr::WRIST_COM[1] + f.c2 * (T)xmr::WRIST_COM[2];
            set_body(Ib[i], (T)xmr::WRIST_MASS, f.c0, f.c1, f.c2, c, xmr::WRIST_INERTIA);
        }
        fb[i] = bias_force(Ib[i], vel, acc);
    }
    // gripper tree: every joint turns about +-x of the hand frame
    {
        V3<T> go[6];
        T gphi[6];
        SV<T> gv[6], ga[6];
#pragma unroll
        for (int g = 0; g < 6; g++) {
            const int p = xmr::G_PARENT[g];
            const T sg = (T)xmr::G_SIGN[g];
            // parent frame (hand, or the parent knuckle turned by its cumulative angle)
            T pc = (T)1, ps = (T)0;
            if (p >= 0) xk::xsincos(gphi[p >= 0 ? p : 0], ps, pc);
            const V3<T> pc1 = p < 0 ? f.c1 : f.c1 * pc + f.c2 * ps;
            const V3<T> pc2 = p < 0 ? f.c2 : f.c2 * pc - f.c1 * ps;
            const V3<T> po = p < 0 ? f.o : go[p >= 0 ? p : 0];
            go[g] = po + f.c0 * (T)xmr::G_ORG[g][0] + pc1 * (T)xmr::G_ORG[g][1] + pc2 * (T)xmr::G_ORG[g][2];
            gphi[g] = (p < 0 ? (T)0 : gphi[p >= 0 ? p : 0]) + sg * st.q[7 + g];
            T c, s;
            xk::xsincos(gphi[g], s, c);
            const V3<T> c1 = f.c1 * c + f.c2 * s, c2 = f.c2 * c - f.c1 * s;
            const V3<T> ax = f.c0 * sg;
            S[7 + g].w = ax;
            S[7 + g].v = cross(go[g], ax);
            const SV<T> vp = p < 0 ? vel : gv[p >= 0 ? p : 0];
            const SV<T> ap = p < 0 ? acc : ga[p >= 0 ? p : 0];
            const T qd = st.qd[7 + g];
            ga[g].w = ap.w + cross(vp.w, S[7 + g].w) * qd;
            ga[g].v = ap.v + (cross(vp.w, S[7 + g].v) + cross(vp.v, S[7 + g].w)) * qd;
            gv[g].w = vp.w + S[7 + g].w * qd;
            gv[g].v = vp.v + S[7 + g].v * qd;
            V3<T> cm = go[g] + f.c0 * (T)xmr::G_COM[g][0] + c1 * (T)xmr::G_COM[g][1] + c2 * (T)xmr::G_COM[g][2];
            set_body(Ib[7 + g], (T)xmr::G_MASS[g], f.c0, c1, c2, cm, xmr::G_INERTIA[g]);
            fb[7 + g] = bias_force(Ib[7 + g], gv[g], ga[g]);
        }
    }
    // composite inertias / forces bottom-up: children have higher gripper indices than their parents
    T M[91], tau[ND];
#pragma unroll
    for (int k = 0; k < 91; k++) M[k] = (T)0;
#pragma unroll
    for (int g = 5; g >= 0; g--) {
        const int p = xmr::G_PARENT[g];
        SV<T> F = rbi_mul(Ib[7 + g], S[7 + g]); // Ib is already the subtree composite of g
        M[tri(7 + g, 7 + g)] = sdot(S[7 + g], F);
        if (p >= 0) M[tri(7 + g, 7 + (p >= 0 ? p : 0))] = sdot(S[7 + (p >= 0 ? p : 0)], F);
#pragma unroll
        for (int i = 0; i < 7; i++) M[tri(7 + g, i)] = sdot(S[i], F);
        tau[7 + g] = -sdot(S[7 + g], fb[7 + g]);
        const int tgt = p < 0 ? 6 : 7 + (p >= 0 ? p : 0);
        rbi_add(Ib[tgt], Ib[7 + g]);
        fb[tgt].w = fb[tgt].w + fb[7 + g].w;
        fb[tgt].v = fb[tgt].v + fb[7 + g].v;
    }
    {
        RBI<T> Ic = Ib[6];
        SV<T> fc = fb[6];
#pragma unroll
        for (int j = 6; j >= 0; j--) {
            if (j < 6) {
                rbi_add(Ic, Ib[j]);
                fc.w = fc.w + fb[j].w; fc.v = fc.v + fb[j].v;
            }
            SV<T> F = rbi_mul(Ic, S[j]);
#pragma unroll
            for (int i = 0; i <= j; i++) M[tri(j, i)] = sdot(S[i], F);
            tau[j] = -sdot(S[j], fc) - (T)xm::DAMPING[j] * st.qd[j];
        }
    }
    // Cholesky, Linv, Minv (13 x 13 packed lower)
    T Minv[91];
    {
        T rd[ND];
#pragma unroll
        for (int c = 0; c < ND; c++) {
#pragma unroll
            for (int r = c; r < ND; r++) {
                T s = M[tri(r, c)];
#pragma unroll
                for (int k = 0; k < c; k++) s -= M[tri(r, k)] * M[tri(c, k)];
                if (r == c) { M[tri(c, c)] = xk::xsqrt(s); rd[c] = (T)1 / M[tri(c, c)]; }
                else M[tri(r, c)] = s * rd[c];
            }
        }
        T Li[91];
#pragma unroll
        for (int c = 0; c < ND; c++) {
            Li[tri(c, c)] = rd[c];
#pragma unroll
            for (int r = c + 1; r < ND; r++) {
                T s = (T)0;
#pragma unroll
                for (int k = c; k < r; k++) s -= M[tri(r, k)] * Li[tri(k, c)];
                Li[tri(r, c)] = s * rd[r];
            }
        }
#pragma unroll
        for (int r = 0; r < ND; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) {
                T s = (T)0;
#pragma unroll
                for (int k = r; k < ND; k++) s += Li[tri(k, r)] * Li[tri(k, c)];
                Minv[tri(r, c)] = s;
            }
    }
    T dq[ND];
#pragma unroll
    for (int r = 0; r < ND; r++) {
        T s = (T)0;
#pragma unroll
        for (int c = 0; c < ND; c++) s += Minv[symi(r, c)] * tau[c];
        dq[r] = st.qd[r] + dt * s;
    }
    // rows: 13 motors, then per dof the limit side that is inside the window (range > 2 * window)
    T m_vt[ND], m_invd[ND], m_lam[ND], l_vt[ND], l_sg[ND], l_lam[ND];
    const T m_hi = (T)(xmr::MOTOR_FORCE * xmr::TIME_STEP);
#pragma unroll
    for (int i = 0; i < ND; i++) {
        m_vt[i] = (T)xm::MOTOR_KP * (st.qt[i] - st.q[i]) * idt + (T)(1.0 - xm::MOTOR_KD) * dq[i];
        m_invd[i] = (T)1 / Minv[tri(i, i)];
        m_lam[i] = (T)0;
        const T lo_lim = i < 7 ? (T)xm::LOWER[i] : (T)xmr::G_LOWER[i < 7 ? 0 : i - 7];
        const T hi_lim = i < 7 ? (T)xm::UPPER[i] : (T)xmr::G_UPPER[i < 7 ? 0 : i - 7];
        const T g0 = st.q[i] - lo_lim, g1 = hi_lim - st.q[i];
        const bool lo = g0 < (T)xm::LIMIT_WINDOW, hi = g1 < (T)xm::LIMIT_WINDOW;
        const T g = lo ? g0 : g1;
        l_sg[i] = lo ? (T)1 : (hi ? (T)-1 : (T)0);
        l_vt[i] = g < (T)0 ? -(T)xm::GLOBAL_ERP * g * idt : -g * idt;
        l_lam[i] = (T)0;
    }
    // packed working set (xk::Pk, as the PickAndPlace sweep): 13 joint velocities = 6 pairs + 1, Minv columns as pairs
    static_assert(ND == 13, "pair layout below is for 13 dofs");
    xk::Pk<T> dqp[6], MC[ND][6];
    T dql = dq[12], ML[ND];
#pragma unroll
    for (int k = 0; k < 6; k++) dqp[k] = xk::mkpk<T>(dq[2 * k], dq[2 * k + 1]);
#pragma unroll
    for (int i = 0; i < ND; i++) {
#pragma unroll
        for (int k = 0; k < 6; k++) MC[i][k] = xk::mkpk<T>(Minv[symi(2 * k, i)], Minv[symi(2 * k + 1, i)]);
        ML[i] = Minv[symi(12, i)];
    }
#define XR_DQ(i) ((i) == 12 ? dql : (((i) & 1) ? xk::pkhi(dqp[(i) >> 1]) : xk::pklo(dqp[(i) >> 1])))
#define XR_DQ_AXPY(col, dl_) do { _Pragma("unroll") for (int k_ = 0; k_ < 6; k_++) dqp[k_] = xk::pkfma(MC[col][k_], (dl_), dqp[k_]); dql += ML[col] * (dl_); } while (0)
#pragma unroll 1
    for (int it = 0; it < xm::NUM_ITERATIONS; it++) {
#pragma unroll
        for (int i = 0; i < ND; i++) {
            T dl = (m_vt[i] - XR_DQ(i)) * m_invd[i];
            const T nl = xk::sclamp(m_lam[i] + dl, -m_hi, m_hi);
            dl = nl - m_lam[i];
            m_lam[i] = nl;
            XR_DQ_AXPY(i, dl);
        }
#pragma unroll
        for (int i = 0; i < ND; i++) {
            if (!XARM_ANY(l_sg[i] != (T)0)) continue;
            const T sg = l_sg[i];
            T dl = (l_vt[i] - sg * XR_DQ(i)) * (sg != (T)0 ? m_invd[i] : (T)0);
            T nl = l_lam[i] + dl;
            nl = xk::smax0(nl);
            dl = (nl - l_lam[i]) * sg;
            l_lam[i] = nl;
            XR_DQ_AXPY(i, dl);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) { dq[2 * k] = xk::pklo(dqp[k]); dq[2 * k + 1] = xk::pkhi(dqp[k]); }
    dq[12] = dql;
#undef XR_DQ
#undef XR_DQ_AXPY
#pragma unroll
    for (int i = 0; i < ND; i++) { st.qd[i] = dq[i]; st.q[i] += dt * dq[i]; }
}

template <typename T> XARM_HD void sim_tick(EnvState<T> &st) {
    const T dt = (T)(xmr::TIME_STEP / xmr::N_SUBSTEPS);
#pragma unroll 1
    for (int k = 0; k < xmr::N_SUBSTEPS; k++) substep<T>(st, dt);
}

// _get_obs :144-161: COM of link 9 (xarm_gripper_base_link) position / linear velocity, driver joint q, qd
template <typename T> XARM_HD void get_obs(const EnvState<T> &st, T (&obs)[OBS_DIM]) {
    Frame<T> f = xk::frame_identity<T>();
    V3<T> w = mk<T>(0, 0, 0), v = mk<T>(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        xk::fk_advance(f, i, st.q[i]);
        w = w + f.c2 * st.qd[i];
        v = v + cross(f.o, f.c2) * st.qd[i];
    }
    const V3<T> hp = f.o + f.c0 * (T)xmr::HAND_COM[0] + f.c1 * (T)xmr::HAND_COM[1] + f.c2 * (T)xmr::HAND_COM[2];
    const V3<T> hv = v + cross(w, hp);
    obs[0] = hp.x; obs[1] = hp.y; obs[2] = hp.z;
    obs[3] = hv.x; obs[4] = hv.y; obs[5] = hv.z;
    obs[6] = st.q[xmr::DRIVER_DOF];
    obs[7] = st.qd[xmr::DRIVER_DOF];
}

template <typename T> XARM_HD void sample_goal(const EnvCfg &cfg, int64_t env, int64_t episode, EnvState<T> &st) {
    uint32_t o[4];
    const uint64_t gid = (uint64_t)(cfg.env_id_offset + env);
    xk::philox(cfg.seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, 0u, o);
#pragma unroll
    for (int k = 0; k < 3; k++) st.goal[k] = (T)xmr::GOAL_LOW[k] + xk::u01<T>(o[k]) * (T)(xmr::GOAL_HIGH[k] - xmr::GOAL_LOW[k]);
}
template <typename T> XARM_HD void env_init(const EnvCfg &cfg, int64_t env, EnvState<T> &st) {
#pragma unroll
    for (int i = 0; i < ND; i++) { st.q[i] = st.qt[i] = (T)xmr::JOINT_INIT_POS[i]; st.qd[i] = (T)0; }
    st.d_old = st.steps = st.episode = (T)0;
    sample_goal(cfg, env, 0, st);
}
template <typename T> XARM_HD T goal_dist(const EnvState<T> &st, const T (&obs)[OBS_DIM]) {
    const T dx = obs[0] - st.goal[0], dy = obs[1] - st.goal[1], dz = obs[2] - st.goal[2];
    return xk::xsqrt(dx * dx + dy * dy + dz * dz);
}
// reset :96-102: resetJointState(joint_init_pos), one stepSimulation (motors keep their last targets),
// new goal, d_old
template <typename T> XARM_HD void env_reset(const EnvCfg &cfg, int64_t env, EnvState<T> &st, T (&obs)[OBS_DIM]) {
    const int64_t episode = (int64_t)st.episode + 1;
#pragma unroll
    for (int i = 0; i < ND; i++) { st.q[i] = (T)xmr::JOINT_INIT_POS[i]; st.qd[i] = (T)0; }
    sim_tick<T>(st);
    sample_goal(cfg, env, episode, st);
    get_obs(st, obs);
    st.d_old = goal_dist(st, obs);
    st.steps = (T)0;
    st.episode = (T)episode;
}
template <typename T>
XARM_HD void env_step(const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&obs)[OBS_DIM], T &reward, bool &done,
                      bool &success, int &future_length) {
    st.steps += (T)1;
    T a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    Frame<T> f = xk::frame_identity<T>();
#pragma unroll
    for (int i = 0; i < 7; i++) xk::fk_advance(f, i, st.q[i]);
    const T sc = (T)(xmr::MAX_VEL * xmr::ACTION_DT);
    const V3<T> target = mk<T>(clampT(f.o.x + a[0] * sc, (T)xmr::POS_LOW[0], (T)xmr::POS_HIGH[0]),
                               clampT(f.o.y + a[1] * sc, (T)xmr::POS_LOW[1], (T)xmr::POS_HIGH[1]),
                               clampT(f.o.z + a[2] * sc, (T)xmr::POS_LOW[2], (T)xmr::POS_HIGH[2]));
    const T g = st.q[xmr::DRIVER_DOF] + a[3] * (T)(xmr::ACTION_DT * xmr::MAX_GRIPPER_VEL); // no clip (:137)
    T qa[7], qo[7];
#pragma unroll
    for (int i = 0; i < 7; i++) qa[i] = st.q[i];
    xk::ik_arm<T, xmr::N_SUBSTEPS>(qa, target, qo);
#pragma unroll
    for (int i = 0; i < ND; i++) st.qt[i] = i < 7 ? qo[i < 7 ? i : 0] : g;
    sim_tick<T>(st);
    get_obs(st, obs);
    const T dist = goal_dist(st, obs);
    success = dist < (T)xmr::DISTANCE_THRESHOLD;
    if (cfg.reward_type == 0) reward = success ? (T)1 : (T)0;
    else if (cfg.reward_type == 1) reward = -dist;
    else { reward = st.d_old - dist; st.d_old = dist; }
    done = (int)st.steps == xmr::MAX_EPISODE_STEPS;
    future_length = xmr::MAX_EPISODE_STEPS - (int)st.steps;
}

} // namespace xr
