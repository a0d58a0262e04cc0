// xarm_handover_core.h - per-lane core of XarmHandover-v0 / XarmPDHandover-v0 (dual arm, one stick).
//
// Reference: /root/reference/gym_xarm/envs/xarm_handover.py
//   env_step  = XarmHandover.step   :128-139 (+ _set_action :244-297, _get_obs :299-336)
//   env_reset = XarmHandover.reset  :141-145 (+ _reset_sim :338-368, _sample_goal :370-393)
//   reward    = compute_reward (sparse, hard-wired :40) :164-183, _is_success :395-402
//   tick      = one p.stepSimulation() at timeStep 1/240 (no substeps), 15 per env step (:131-132)
// num_obj = 1 and use_stand = False (the configuration BASELINE.json names).
//
// Mapping: TWO lanes per environment, lane = arm.  Each lane carries one xarm7_pd arm (the same
// 9-dof chain, CRBA, Cholesky, motor / limit / gear rows and pad blocks as PickAndPlace: xk::substep with
// HandoverScene) plus a private copy of the object state.  Rows that involve only the object (stick /
// table / ground) are computed identically by both lanes; the finger/object blocks are swept arm 0 first,
// then arm 1, and the object velocity is handed from one lane to its partner in between (Xchg).  This keeps
// the per-lane register / LDS footprint of the single-arm kernel and still is the sequential Gauss-Seidel
// sweep T, (M L G)_0, (M L G)_1, F_0, F_1 of the oracle.
#pragma once
#include "xarm_core.h"

namespace xh {
using xk::V3;
using xk::mk;
using xk::Frame;
using xk::clampT;
using xk::EnvState;

constexpr int STATE_DIM = 76; // q[2][9] qd[2][9] finger_target[2] obj 13 goal 3 lam_table 8 lam_pad[2][4] touch[2] mug[2] steps episode
constexpr int OBS_DIM = 29;
constexpr int ACT_DIM = 8;
enum { H_Q = 0, H_QD = 18, H_FT = 36, H_BP = 38, H_BQ = 41, H_BV = 45, H_BW = 48, H_GOAL = 51, H_LT = 54, H_LP = 62,
       H_TOUCH = 70, H_MUG = 72, H_STEPS = 74, H_EPISODE = 75 };

struct EnvCfg {
    uint64_t seed;
    int64_t env_id_offset;
    float same_side_rate;
    int goal_shape; // 1 = 'ground'
    int use_stand;  // config['use_stand'] (:391-392): kernels instantiated with HandoverStandScene
    int reward_type; // 0 sparse (hard-wired in the reference, :40), 1 the staged dense reward (:184-199)
};

struct HandoverScene {
    static constexpr int NARMS = 2;
    static constexpr bool HAS_STAND = false;
    static constexpr double OBJ_HX = xm::HO_OBJ_HALF[0], OBJ_HY = xm::HO_OBJ_HALF[1], OBJ_HZ = xm::HO_OBJ_HALF[2];
    static constexpr double OBJ_MASS = xm::HO_OBJ_MASS;
    static constexpr double TIME_STEP = xm::HO_TIME_STEP;
    static constexpr double FINGER_MOTOR_FORCE = xm::HO_FINGER_MOTOR_FORCE;
    static constexpr double LIN_DAMP_FACTOR = xm::HO_LIN_DAMP_FACTOR, ANG_DAMP_FACTOR = xm::HO_ANG_DAMP_FACTOR;
    // arm bases at x = -+0.6, the second yawed by pi (:50-53)
    template <typename T> static XARM_HD Frame<T> base_frame(int arm) {
        const T c = arm == 0 ? (T)xm::HO_BASE_COS[0] : (T)xm::HO_BASE_COS[1];
        const T s = arm == 0 ? (T)xm::HO_BASE_SIN[0] : (T)xm::HO_BASE_SIN[1];
        Frame<T> f;
        f.c0 = mk<T>(c, s, (T)0); f.c1 = mk<T>(-s, c, (T)0); f.c2 = mk<T>((T)0, (T)0, (T)1);
        f.o = arm == 0 ? mk<T>((T)xm::HO_BASE_POS[0][0], (T)xm::HO_BASE_POS[0][1], (T)xm::HO_BASE_POS[0][2])
                       : mk<T>((T)xm::HO_BASE_POS[1][0], (T)xm::HO_BASE_POS[1][1], (T)xm::HO_BASE_POS[1][2]);
        return f;
    }
    // two table tops at z = 0 for table_x_min <= |x| <= table_x_max, |y| <= 0.5 (:81-83), the ground plane at
    // z = -0.625 (:79) everywhere else
    template <typename T> static XARM_HD bool support(V3<T> p, T &height) {
        const bool on_table = xk::xabs(p.x) >= (T)xm::HO_TABLE_X_MIN && xk::xabs(p.x) <= (T)xm::HO_TABLE_X_MAX &&
                              xk::xabs(p.y) <= (T)xm::HO_TABLE_HALF_Y;
        height = on_table ? (T)xm::TABLE_TOP_Z : (T)xm::HO_GROUND_Z;
        return true;
    }
};

// config['use_stand'] (:391-392): urdf/my_stand.urdf, a fixed 0.07 x 0.06 x 0.01 box whose top sits 25 mm under the goal.
// Contact model (model JSON handover._stand_cite, restated by the oracle): the stick's most downward face against the
// stand's top rectangle - the overlap of the face's axis-aligned footprint with the rectangle (the env zeroes the
// stick's roll and yaw every step, :282-297) gives up to four support points on the face's plane, normal +z.
struct HandoverStandScene : HandoverScene {
    static constexpr bool HAS_STAND = true;
    static constexpr double STAND_MIN_GAP = -(2.0 * xm::HO_STAND_HALF[2] + 0.01);
    // sp[v]: support point v on the face's plane, sd[v]: its gap to the stand's top (1e30 when there is no overlap)
    template <typename T> static XARM_HD void stand_points(const T (&goal)[3], V3<T> cb, V3<T> b0, V3<T> b1, V3<T> b2, V3<T> (&sp)[4], T (&sd)[4]) {
        const T top = goal[2] - (T)xm::HO_STAND_BELOW_GOAL + (T)xm::HO_STAND_HALF[2];
        // the face (axis kf, sign sg) whose outward normal points most downward; ties -> the first in (k, -/+) order
        const T z0 = b0.z, z1 = b1.z, z2 = b2.z;
        int kf = 0;
        T sg = (T)-1, best = -z0;
        if (z0 < best) { best = z0; kf = 0; sg = (T)1; }
        if (-z1 < best) { best = -z1; kf = 1; sg = (T)-1; }
        if (z1 < best) { best = z1; kf = 1; sg = (T)1; }
        if (-z2 < best) { best = -z2; kf = 2; sg = (T)-1; }
        if (z2 < best) { best = z2; kf = 2; sg = (T)1; }
        const T hx = (T)OBJ_HX, hy = (T)OBJ_HY, hz = (T)OBJ_HZ;
        const V3<T> af = xk::selv(kf == 0, b0, xk::selv(kf == 1, b1, b2));          // face axis
        const V3<T> a1 = xk::selv(kf == 0, b1, xk::selv(kf == 1, b2, b0));          // the two in-face axes (k+1, k+2)
        const V3<T> a2 = xk::selv(kf == 0, b2, xk::selv(kf == 1, b0, b1));
        const T hf = kf == 0 ? hx : (kf == 1 ? hy : hz), h1 = kf == 0 ? hy : (kf == 1 ? hz : hx), h2 = kf == 0 ? hz : (kf == 1 ? hx : hy);
        const V3<T> nf = af * sg, fc = cb + nf * hf;
        T xlo = (T)1e30, xhi = (T)-1e30, ylo = (T)1e30, yhi = (T)-1e30;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const T s1 = (v & 1) ? h1 : -h1, s2 = (v & 2) ? h2 : -h2;
            const T x = fc.x + s1 * a1.x + s2 * a2.x, y = fc.y + s1 * a1.y + s2 * a2.y;
            xlo = x < xlo ? x : xlo; xhi = x > xhi ? x : xhi; ylo = y < ylo ? y : ylo; yhi = y > yhi ? y : yhi;
        }
        const T sx0 = goal[0] - (T)xm::HO_STAND_HALF[0], sx1 = goal[0] + (T)xm::HO_STAND_HALF[0];
        const T sy0 = goal[1] - (T)xm::HO_STAND_HALF[1], sy1 = goal[1] + (T)xm::HO_STAND_HALF[1];
        const T ox0 = xlo > sx0 ? xlo : sx0, ox1 = xhi < sx1 ? xhi : sx1, oy0 = ylo > sy0 ? ylo : sy0, oy1 = yhi < sy1 ? yhi : sy1;
        const bool overlap = ox0 < ox1 && oy0 < oy1;
        const T inz = (T)1 / nf.z;                                                  // nf.z <= -1/sqrt(3)
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const T x = (v & 1) ? ox1 : ox0, y = (v & 2) ? oy1 : oy0;
            const T z = fc.z - (nf.x * (x - fc.x) + nf.y * (y - fc.y)) * inz;
            sp[v] = mk<T>(x, y, z);
            sd[v] = overlap ? z - top : (T)1e30;
        }
    }
};

// one lane = one arm: st.q/qd/lam_p[0..3]/touch/mug are the arm's, the rest is the lane's copy of the shared state
template <typename T> struct Lane {
    EnvState<T> st;
    T ft; // finger motor target of this arm (persists across reset ticks, :265-268 vs :347-353)
};

template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene> XARM_HD void tick(Lane<T> &L, const T (&qt)[9], Lds lds, int arm, Xchg x) {
    xk::substep<T, Lds, Scene, Xchg>(L.st, qt, (T)xm::HO_TIME_STEP, lds, arm, x);
}

template <typename T> XARM_HD V3<T> eef_pos(const Lane<T> &L, int arm) {
    Frame<T> f = HandoverScene::base_frame<T>(arm);
#pragma unroll
    for (int i = 0; i < 7; i++) xk::fk_advance(f, i, L.st.q[i]);
    return f.o;
}
template <typename T> XARM_HD void ik(const Lane<T> &L, int arm, V3<T> target, T (&qt)[9]) {
    T qa[7], qo[7];
#pragma unroll
    for (int i = 0; i < 7; i++) qa[i] = L.st.q[i];
    xk::ik_arm<T, xm::HO_N_TICKS>(qa, target, qo, HandoverScene::base_frame<T>(arm));
#pragma unroll
    for (int i = 0; i < 7; i++) qt[i] = qo[i];
    qt[7] = qt[8] = L.ft;
}

// the 8 per-arm observation entries (:304-313): grip_pos = hand COM - eef2grip, hand COM velocity, finger q, qd
template <typename T> XARM_HD void arm_obs(const Lane<T> &L, int arm, T (&o)[8]) {
    Frame<T> f = HandoverScene::base_frame<T>(arm);
    V3<T> w = mk<T>(0, 0, 0), v = mk<T>(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        xk::fk_advance(f, i, L.st.q[i]);
        w = w + f.c2 * L.st.qd[i];
        v = v + xk::cross(f.o, f.c2) * L.st.qd[i];
    }
    const V3<T> hp = f.o + f.c0 * (T)xm::HAND_COM[0] + f.c1 * (T)xm::HAND_COM[1] + f.c2 * (T)xm::HAND_COM[2];
    const V3<T> hv = v + xk::cross(w, hp);
    o[0] = hp.x - (T)xm::HO_EEF2GRIP[0]; o[1] = hp.y - (T)xm::HO_EEF2GRIP[1]; o[2] = hp.z - (T)xm::HO_EEF2GRIP[2];
    o[3] = hv.x; o[4] = hv.y; o[5] = hv.z;
    o[6] = L.st.q[7]; o[7] = L.st.qd[7];
}

// draws: 0-1 object xy, 2 mirror coin, 3-5 goal xyz, 6 same-side coin
template <typename T> XARM_HD void draws(const EnvCfg &cfg, int64_t env, int64_t episode, T (&u)[8]) {
    const uint64_t gid = (uint64_t)(cfg.env_id_offset + env);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        uint32_t o[4];
        xk::philox(cfg.seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, (uint32_t)b, o);
#pragma unroll
        for (int k = 0; k < 4; k++) u[b * 4 + k] = xk::u01<T>(o[k]);
    }
}
template <typename T> XARM_HD void sample_object(const T (&u)[8], Lane<T> &L) {
    const T x = (T)xm::HO_OBJ_LOW[0] + u[0] * (T)(xm::HO_OBJ_HIGH[0] - xm::HO_OBJ_LOW[0]);
    L.st.bp[0] = u[2] < (T)0.5 ? -x : x;
    L.st.bp[1] = (T)xm::HO_OBJ_LOW[1] + u[1] * (T)(xm::HO_OBJ_HIGH[1] - xm::HO_OBJ_LOW[1]);
    L.st.bp[2] = (T)xm::HO_HEIGHT_OFFSET;
    L.st.bq[0] = L.st.bq[1] = L.st.bq[2] = (T)0; L.st.bq[3] = (T)1;
#pragma unroll
    for (int k = 0; k < 3; k++) { L.st.bv[k] = (T)0; L.st.bw[k] = (T)0; }
#pragma unroll
    for (int k = 0; k < 8; k++) { L.st.lam_t[k] = (T)0; L.st.lam_p[k] = (T)0; }
}
template <typename T> XARM_HD void sample_goal(const EnvCfg &cfg, const T (&u)[8], Lane<T> &L) {
#pragma unroll
    for (int k = 0; k < 3; k++) L.st.goal[k] = (T)xm::HO_GOAL_LOW[k] + u[3 + k] * (T)(xm::HO_GOAL_HIGH[k] - xm::HO_GOAL_LOW[k]);
    const bool same = u[6] < (T)cfg.same_side_rate;
    if ((L.st.bp[0] > (T)0) != same) L.st.goal[0] = -L.st.goal[0];
    if (cfg.goal_shape == 1) L.st.goal[2] = (T)xm::HO_HEIGHT_OFFSET;
}
template <typename T> XARM_HD void lane_init(const EnvCfg &cfg, int64_t env, Lane<T> &L) {
#pragma unroll
    for (int i = 0; i < 9; i++) { L.st.q[i] = (T)xm::HO_JOINT_INIT_POS[i]; L.st.qd[i] = (T)0; }
    L.ft = (T)xm::HO_JOINT_INIT_POS[7];
    L.st.touch = L.st.mug = L.st.steps = L.st.episode = (T)0;
    T u[8];
    draws(cfg, env, 0, u);
    sample_object(u, L);
    sample_goal(cfg, u, L);
}

template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD void lane_reset(const EnvCfg &cfg, int64_t env, Lane<T> &L, int arm, Lds lds, Xchg x) {
    const int64_t episode = (int64_t)L.st.episode + 1;
    T qt[9], u[8];
    const V3<T> home = arm == 0 ? mk<T>((T)xm::HO_EFF_INIT_POS[0][0], (T)xm::HO_EFF_INIT_POS[0][1], (T)xm::HO_EFF_INIT_POS[0][2])
                                : mk<T>((T)xm::HO_EFF_INIT_POS[1][0], (T)xm::HO_EFF_INIT_POS[1][1], (T)xm::HO_EFF_INIT_POS[1][2]);
#pragma unroll 1
    for (int k = 0; k <= xm::HO_RESET_TICKS; k++) {
        if (k < xm::HO_RESET_TICKS) ik(L, arm, home, qt);
        else {
            draws(cfg, env, episode, u);
            sample_object(u, L);
        }
        tick<T, Lds, Xchg, Scene>(L, qt, lds, arm, x);
    }
    sample_goal(cfg, u, L);
    L.st.steps = (T)0;
    L.st.episode = (T)episode;
}

// staged dense reward (:184-199), evaluated identically by both lanes of the env: grip_k = hand COM of arm k -
// eef2grip_offset (this lane's from arm_obs, the other arm's from its lane), if_k = the grasp flags _set_action read
// before the step's simulation (:263-264 = st.mug).  The reference's last branch (only arm 2 holds the object) reads
// an undefined `d` (:199) and raises; d = |achieved_goal - goal| here, the distance its docstring's stage 7 means.
template <typename T, typename Xchg>
XARM_HD T dense_reward(const Lane<T> &L, int arm, T d_og, Xchg x) {
    T o8[8];
    arm_obs(L, arm, o8);
    const V3<T> g1 = mk<T>(x.from0(o8[0]), x.from0(o8[1]), x.from0(o8[2])), g2 = mk<T>(x.from1(o8[0]), x.from1(o8[1]), x.from1(o8[2]));
    const bool if1 = x.from0(L.st.mug) > (T)0.5, if2 = x.from1(L.st.mug) > (T)0.5;
    const V3<T> ag = mk<T>(L.st.bp[0], L.st.bp[1], L.st.bp[2]);
    const V3<T> e1 = g1 - ag + mk<T>((T)0.06, (T)0, (T)0), e2 = g2 - ag + mk<T>((T)-0.06, (T)0, (T)0);
    const T d1 = xk::xsqrt(xk::dot(e1, e1)), d2 = xk::xsqrt(xk::dot(e2, e2));
    const T k = (T)(1.0 / 2.25);
    if (!if1 && !if2) return (T)0.25 * ((T)1 - xk::xtanh(d1)) * k;
    if (if1 && !if2) return ag.z > (T)0.05 ? ((T)1 + (T)0.25 * ((T)1 - xk::xtanh(d2))) * k : (T)0.5 * k;
    if (if1 && if2) return (T)1.5 * k;
    return ((T)2 + (T)0.25 * ((T)1 - xk::xtanh(d_og))) * k;
}

// XarmHandover.step up to the simulation (:128-130, _set_action :244-297): clip, Cartesian target, IK, finger target, friction
// toggle, stick clamp.  Shared by every mapping of the step (lane pair, pad-free fast lane pair, cooperative rows).
template <typename T> XARM_HD void step_begin(Lane<T> &L, int arm, const T (&act)[4], T (&qt)[9]) {
    L.st.steps += (T)1;
    T a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    const V3<T> cur = eef_pos(L, arm);
    const T sc = (T)(xm::HO_MAX_VEL * xm::HO_ACTION_DT);
    const V3<T> lo = arm == 0 ? mk<T>((T)xm::HO_POS_LOW[0][0], (T)xm::HO_POS_LOW[0][1], (T)xm::HO_POS_LOW[0][2])
                              : mk<T>((T)xm::HO_POS_LOW[1][0], (T)xm::HO_POS_LOW[1][1], (T)xm::HO_POS_LOW[1][2]);
    const V3<T> hi = arm == 0 ? mk<T>((T)xm::HO_POS_HIGH[0][0], (T)xm::HO_POS_HIGH[0][1], (T)xm::HO_POS_HIGH[0][2])
                              : mk<T>((T)xm::HO_POS_HIGH[1][0], (T)xm::HO_POS_HIGH[1][1], (T)xm::HO_POS_HIGH[1][2]);
    const V3<T> target = mk<T>(clampT(cur.x + a[0] * sc, lo.x, hi.x), clampT(cur.y + a[1] * sc, lo.y, hi.y), clampT(cur.z + a[2] * sc, lo.z, hi.z));
    L.ft = clampT(L.st.q[7] + a[3] * (T)(xm::HO_ACTION_DT * xm::HO_MAX_GRIPPER_VEL), (T)xm::HO_GRIPPER_LOW, (T)xm::HO_GRIPPER_HIGH);
    ik(L, arm, target, qt);
    L.st.mug = L.st.touch; // friction toggle from the current contact points (:269-280)
    {
        // clamp the stick into the play field, keep only its pitch, zero its velocity (:282-297)
        const T qx = L.st.bq[0], qy = L.st.bq[1], qz = L.st.bq[2], qw = L.st.bq[3];
        const T sarg = (T)2 * (qw * qy - qx * qz);
        const T hp = (T)1.57079632679489661923;
        const T pitch = sarg <= (T)-0.99999 ? -hp : (sarg >= (T)0.99999 ? hp : xk::xasin(sarg));
        T sp, cp;
        xk::xsincos((T)0.5 * pitch, sp, cp);
        L.st.bq[0] = (T)0; L.st.bq[1] = sp; L.st.bq[2] = (T)0; L.st.bq[3] = cp;
        L.st.bp[0] = clampT(L.st.bp[0], -(T)xm::HO_OBJ_HIGH[0], (T)xm::HO_OBJ_HIGH[0]);
        L.st.bp[1] = clampT(L.st.bp[1], -(T)xm::HO_OBJ_HIGH[1], (T)xm::HO_OBJ_HIGH[1]);
#pragma unroll
        for (int k = 0; k < 3; k++) { L.st.bv[k] = (T)0; L.st.bw[k] = (T)0; }
    }
}
// ... and after it: reward (:164-199), success (:395-402), done (success or the 100-step limit)
template <typename T, typename Xchg>
XARM_HD void step_end(const Lane<T> &L, int arm, T &reward, bool &done, bool &success, Xchg x, int reward_type) {
    const T dx = L.st.bp[0] - L.st.goal[0], dy = L.st.bp[1] - L.st.goal[1], dz = L.st.bp[2] - L.st.goal[2];
    const T dist = xk::xsqrt(dx * dx + dy * dy + dz * dz);
    success = dist < (T)xm::HO_DISTANCE_THRESHOLD;
    reward = dist > (T)xm::HO_DISTANCE_THRESHOLD ? (T)-1 : (T)0;   // -sum(d > thr) for one object (:177-181)
    if (reward_type == 1) reward = dense_reward<T, Xchg>(L, arm, dist, x);
    done = success || ((int)L.st.steps == xm::HO_MAX_EPISODE_STEPS);
}

// Ticks [tick0, tick1) of XarmHandover.step: tick0 == 0 opens the step (action -> joint targets qt, friction toggle, stick
// clamp: step_begin), a later tick0 continues it with the qt the opening stage left; tick1 == HO_N_TICKS closes it (reward,
// done, success: step_end).  The staged hand-off of xarm_step (xarm_hip.hip) cuts a step into stages so that an env whose
// finger pads come alive in stage c re-runs only the ticks from that stage's first one on the cooperative rows.
// act = this arm's 4 action entries (:249-256)
template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD void lane_step_range(Lane<T> &L, int arm, const T (&act)[4], T (&qt)[9], int tick0, int tick1, T &reward, bool &done, bool &success,
                             Lds lds, Xchg x, int reward_type = 0) {
    if (tick0 == 0) step_begin(L, arm, act, qt);
#pragma unroll 1
    for (int k = tick0; k < tick1; k++) tick<T, Lds, Xchg, Scene>(L, qt, lds, arm, x);
    if (tick1 == xm::HO_N_TICKS) step_end<T, Xchg>(L, arm, reward, done, success, x, reward_type);
}
template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD void lane_step(Lane<T> &L, int arm, const T (&act)[4], T &reward, bool &done, bool &success, Lds lds, Xchg x, int reward_type = 0) {
    T qt[9];
    lane_step_range<T, Lds, Xchg, Scene>(L, arm, act, qt, 0, xm::HO_N_TICKS, reward, done, success, lds, x, reward_type);
}

// The same step on the pad-free fast substep (xk::substep<.., FAST>: no finger-pad rows, nothing of the arm in LDS, no
// exchange inside the sweep - without pad rows the two arms never interact and the object-only rows are computed
// identically by both lanes).  Returns false when a finger-pad row of EITHER arm was active in any of the ticks run: the
// outputs are then meaningless and the caller must not store them (the environment is stepped again, from the untouched
// state it had before these ticks, by the cooperative kernel: xarm_handover_coop_core.h).
template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD bool lane_step_fast_range(Lane<T> &L, int arm, const T (&act)[4], T (&qt)[9], int tick0, int tick1, T &reward, bool &done, bool &success,
                                  Lds lds, Xchg x, int reward_type = 0) {
    if (tick0 == 0) step_begin(L, arm, act, qt);
    bool pad = false;
#pragma unroll 1
    for (int k = tick0; k < tick1; k++)
        pad = xk::substep<T, Lds, Scene, Xchg, true>(L.st, qt, (T)xm::HO_TIME_STEP, lds, arm, x) || pad;
    if (tick1 == xm::HO_N_TICKS) step_end<T, Xchg>(L, arm, reward, done, success, x, reward_type);
    const bool other = x.partner(pad ? (T)1 : (T)0) != (T)0;
    return !(pad || other);
}
template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD bool lane_step_fast(Lane<T> &L, int arm, const T (&act)[4], T &reward, bool &done, bool &success, Lds lds, Xchg x, int reward_type = 0) {
    T qt[9];
    return lane_step_fast_range<T, Lds, Xchg, Scene>(L, arm, act, qt, 0, xm::HO_N_TICKS, reward, done, success, lds, x, reward_type);
}

} // namespace xh
