// xarm_reach_coop_core.h - cooperative (16 lanes per environment) form of the XarmReach-v0 substep.
//
// XarmReach-v0 at its BASELINE size (4 096 envs) fills 64 of 1 024 SIMDs under the one-env-per-lane mapping of
// xarm_reach_core.h and every step is 20 substeps of pure latency.  Here, as in xarm_coop_core.h, one environment owns
// a DPP row of 16 lanes: lane l = body l = dof l of the 13-dof tree (7 arm joints, 6 gripper joints).
//   * the kinematic chain (a serial recursion) is walked by every lane; lane l captures ITS body's frame, spatial
//     velocity / acceleration and joint axis on the way and forms that body's inertia and bias force;
//   * composites: the gripper tree is folded leaf to root by broadcasts (children have higher indices), the arm chain
//     by a DPP suffix scan; lane l computes row l of the joint-space inertia, rows are broadcast;
//   * the 13x13 Cholesky factor is formed redundantly, but M^-1 is never formed: lane l solves M x = e_l, i.e. it
//     owns ROW l of M^-1 - all the impulse-space sweep needs, because every solver row is a single joint;
//   * sweep: lane l owns motor row l and limit row l as one packed pair (g, lambda, 1/diag); row r is processed by
//     its owner, the impulse change broadcast by DPP row_newbcast and applied with one v_pk_fma_f32 per lane:
//     6-7 instructions per row instead of the 12 of the velocity-space sweep with its 13-vector updates.
// Same rows, same order (13 motors, then the limit rows inside their window) as xr::substep and the oracle.
// Reference: /root/reference/gym_xarm/envs/xarm_reach.py (see xarm_reach_core.h for the line map).
#pragma once
#include "xarm_reach_core.h"
#include "xarm_coop_core.h"

namespace xrc {
using xk::V3; using xk::mk; using xk::dot; using xk::cross; using xk::SV; using xk::RBI; using xk::Frame; using xk::tri; using xk::clampT;
using xc::Grp; using xc::LV; using xc::LV2; using xc::LVN; using xc::lane_of; using xc::lv_fill; using xc::lv_bcast; using xc::lv_get;
using xc::lv_commit; using xc::lv_suffix_sum; using xr::ND; using xr::EnvState; using xr::EnvCfg;

// ancestors-or-self of body l among the 13 dofs (bit i): arm joint l carries joints 0..l; gripper link g hangs off the
// hand (all 7 arm joints) through its own chain (G_PARENT)
XARM_HD constexpr unsigned anc_mask(int l) {
    if (l < 7) return (1u << (l + 1)) - 1u;
    if (l >= 13) return 0u;
    unsigned m = 0x7Fu | (1u << l);
    const int p = xmr::G_PARENT[l - 7];
    if (p >= 0) m |= 1u << (7 + p);
    return m;
}

// per-lane body constants
template <typename T> struct BodyLane { LV<T> mass, com[3], inertia[6], damping, lo, hi; };
template <typename T> XARM_HD BodyLane<T> body_lane_consts(const Grp &G) {
    BodyLane<T> C;
    XC_LANES {
        const int l = lane_of(G, i_);
        T m = (T)0, dmp = (T)0, lo = (T)0, hi = (T)0, c[3] = {(T)0, (T)0, (T)0}, in[6] = {(T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
#pragma unroll
        for (int b = 0; b < ND; b++) {
            const bool me = l == b;
            const double mass = b < 6 ? xm::MASS[b < 6 ? b : 0] : (b == 6 ? xmr::WRIST_MASS : xmr::G_MASS[b >= 7 ? b - 7 : 0]);
            m = me ? (T)mass : m;
            dmp = me ? (T)(b < 7 ? xm::DAMPING[b < 7 ? b : 0] : 0.0) : dmp;
            lo = me ? (T)(b < 7 ? xm::LOWER[b < 7 ? b : 0] : xmr::G_LOWER[b >= 7 ? b - 7 : 0]) : lo;
            hi = me ? (T)(b < 7 ? xm::UPPER[b < 7 ? b : 0] : xmr::G_UPPER[b >= 7 ? b - 7 : 0]) : hi;
#pragma unroll
            for (int k = 0; k < 3; k++)
                c[k] = me ? (T)(b < 6 ? xm::COM[b < 6 ? b : 0][k] : (b == 6 ? xmr::WRIST_COM[k] : xmr::G_COM[b >= 7 ? b - 7 : 0][k])) : c[k];
#pragma unroll
            for (int k = 0; k < 6; k++)
                in[k] = me ? (T)(b < 6 ? xm::INERTIA[b < 6 ? b : 0][k] : (b == 6 ? xmr::WRIST_INERTIA[k] : xmr::G_INERTIA[b >= 7 ? b - 7 : 0][k])) : in[k];
        }
        C.mass.v[i_] = m; C.damping.v[i_] = dmp; C.lo.v[i_] = lo; C.hi.v[i_] = hi;
#pragma unroll
        for (int k = 0; k < 3; k++) C.com[k].v[i_] = c[k];
#pragma unroll
        for (int k = 0; k < 6; k++) C.inertia[k].v[i_] = in[k];
    }
    return C;
}

// dst lane += value held by src lane (one step of the leaf-to-root fold of the gripper tree)
template <int SRC, int DST, typename T> XARM_HD void fold(const Grp &G, LV<T> (&x)[16]) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const LV<T> v = lv_bcast<SRC>(x[k]);
        lv_commit<DST>(G, x[k], xc::lv_add(x[k], v));
    }
}

// one internal substep (dt = 1/4800 s) of one environment, executed by the 16 lanes of its row
template <typename T> XARM_HD void substep(const Grp &G, const BodyLane<T> &C, EnvState<T> &st, const T dt) {
    const T idt = (T)1 / dt;
    // ---- the chain, walked by every lane; lane l captures body l's frame, velocity, acceleration, joint axis, state
    SV<T> S[ND];
    Frame<T> f = xk::frame_identity<T>();
    SV<T> vel, acc;
    vel.w = mk<T>(0, 0, 0); vel.v = mk<T>(0, 0, 0);
    acc.w = mk<T>(0, 0, 0); acc.v = mk<T>(0, 0, (T)xm::GRAVITY);
    LV<T> cap[34];
#pragma unroll
    for (int k = 0; k < 34; k++) cap[k] = lv_fill((T)0);
#define XRC_CAPTURE(i, c0_, c1_, c2_, o_, vv, aa, Sv, qv, qdv, qtv)                                          \
    {                                                                                                        \
        const T vals[34] = {c0_.x, c0_.y, c0_.z, c1_.x, c1_.y, c1_.z, c2_.x, c2_.y, c2_.z, o_.x, o_.y, o_.z, \
                            vv.w.x, vv.w.y, vv.w.z, vv.v.x, vv.v.y, vv.v.z, aa.w.x, aa.w.y, aa.w.z, aa.v.x, aa.v.y, aa.v.z, \
                            Sv.w.x, Sv.w.y, Sv.w.z, Sv.v.x, Sv.v.y, Sv.v.z, qv, qdv, qtv, (T)0};             \
        _Pragma("unroll") for (int k = 0; k < 33; k++) lv_commit<i>(G, cap[k], lv_fill(vals[k]));           \
    }
#define XRC_ARM(i)                                                                                           \
    {                                                                                                        \
        xk::fk_advance(f, i, st.q[i]);                                                                       \
        S[i].w = f.c2;                                                                                       \
        S[i].v = cross(f.o, f.c2);                                                                           \
        const T qd = st.qd[i];                                                                               \
        acc.w = acc.w + cross(vel.w, S[i].w) * qd;                                                           \
        acc.v = acc.v + (cross(vel.w, S[i].v) + cross(vel.v, S[i].w)) * qd;                                  \
        vel.w = vel.w + S[i].w * qd;                                                                         \
        vel.v = vel.v + S[i].v * qd;                                                                         \
        XRC_CAPTURE(i, f.c0, f.c1, f.c2, f.o, vel, acc, S[i], st.q[i], qd, st.qt[i])                         \
    }
    XRC_ARM(0) XRC_ARM(1) XRC_ARM(2) XRC_ARM(3) XRC_ARM(4) XRC_ARM(5) XRC_ARM(6)
#undef XRC_ARM
    // gripper tree: every joint turns about +-x of the hand frame (xr::substep)
    {
        V3<T> go[6];
        T gphi[6];
        SV<T> gv[6], ga[6];
#define XRC_GRIP(g)                                                                                          \
        {                                                                                                    \
            constexpr int p = xmr::G_PARENT[g];                                                              \
            const T sg = (T)xmr::G_SIGN[g];                                                                  \
            T pc = (T)1, ps = (T)0;                                                                          \
            if (p >= 0) xk::xsincos(gphi[p >= 0 ? p : 0], ps, pc);                                           \
            const V3<T> pc1 = p < 0 ? f.c1 : f.c1 * pc + f.c2 * ps;                                          \
            const V3<T> pc2 = p < 0 ? f.c2 : f.c2 * pc - f.c1 * ps;                                          \
            const V3<T> po = p < 0 ? f.o : go[p >= 0 ? p : 0];                                               \
            go[g] = po + f.c0 * (T)xmr::G_ORG[g][0] + pc1 * (T)xmr::G_ORG[g][1] + pc2 * (T)xmr::G_ORG[g][2]; \
            gphi[g] = (p < 0 ? (T)0 : gphi[p >= 0 ? p : 0]) + sg * st.q[7 + g];                              \
            T c, s;                                                                                          \
            xk::xsincos(gphi[g], s, c);                                                                      \
            const V3<T> c1 = f.c1 * c + f.c2 * s, c2 = f.c2 * c - f.c1 * s;                                  \
            const V3<T> ax = f.c0 * sg;                                                                      \
            S[7 + g].w = ax;                                                                                 \
            S[7 + g].v = cross(go[g], ax);                                                                   \
            const SV<T> vp = p < 0 ? vel : gv[p >= 0 ? p : 0];                                               \
            const SV<T> ap = p < 0 ? acc : ga[p >= 0 ? p : 0];                                               \
            const T qd = st.qd[7 + g];                                                                       \
            ga[g].w = ap.w + cross(vp.w, S[7 + g].w) * qd;                                                   \
            ga[g].v = ap.v + (cross(vp.w, S[7 + g].v) + cross(vp.v, S[7 + g].w)) * qd;                       \
            gv[g].w = vp.w + S[7 + g].w * qd;                                                                \
            gv[g].v = vp.v + S[7 + g].v * qd;                                                                \
            XRC_CAPTURE(7 + g, f.c0, c1, c2, go[g], gv[g], ga[g], S[7 + g], st.q[7 + g], qd, st.qt[7 + g])    \
        }
        XRC_GRIP(0) XRC_GRIP(1) XRC_GRIP(2) XRC_GRIP(3) XRC_GRIP(4) XRC_GRIP(5)
#undef XRC_GRIP
    }
#undef XRC_CAPTURE
    // ---- per lane: inertia of the lane's body about the world origin and its bias force
    LV<T> comp[16];
    XC_LANES {
        const int l = lane_of(G, i_);
        const V3<T> c0 = mk<T>(cap[0].v[i_], cap[1].v[i_], cap[2].v[i_]), c1 = mk<T>(cap[3].v[i_], cap[4].v[i_], cap[5].v[i_]),
                    c2 = mk<T>(cap[6].v[i_], cap[7].v[i_], cap[8].v[i_]), o = mk<T>(cap[9].v[i_], cap[10].v[i_], cap[11].v[i_]);
        SV<T> v, a;
        v.w = mk<T>(cap[12].v[i_], cap[13].v[i_], cap[14].v[i_]); v.v = mk<T>(cap[15].v[i_], cap[16].v[i_], cap[17].v[i_]);
        a.w = mk<T>(cap[18].v[i_], cap[19].v[i_], cap[20].v[i_]); a.v = mk<T>(cap[21].v[i_], cap[22].v[i_], cap[23].v[i_]);
        const T m = C.mass.v[i_];
        const V3<T> c = o + c0 * C.com[0].v[i_] + c1 * C.com[1].v[i_] + c2 * C.com[2].v[i_];
        const T ixx = C.inertia[0].v[i_], ixy = C.inertia[1].v[i_], ixz = C.inertia[2].v[i_], iyy = C.inertia[3].v[i_],
                iyz = C.inertia[4].v[i_], izz = C.inertia[5].v[i_];
        const V3<T> m0 = c0 * ixx + c1 * ixy + c2 * ixz, m1 = c0 * ixy + c1 * iyy + c2 * iyz, m2 = c0 * ixz + c1 * iyz + c2 * izz;
        const T cc = dot(c, c);
        RBI<T> I;
        I.m = m;
        I.h = c * m;
        I.I[0] = m0.x * c0.x + m1.x * c1.x + m2.x * c2.x + m * (cc - c.x * c.x);
        I.I[1] = m0.x * c0.y + m1.x * c1.y + m2.x * c2.y - m * c.x * c.y;
        I.I[2] = m0.x * c0.z + m1.x * c1.z + m2.x * c2.z - m * c.x * c.z;
        I.I[3] = m0.y * c0.y + m1.y * c1.y + m2.y * c2.y + m * (cc - c.y * c.y);
        I.I[4] = m0.y * c0.z + m1.y * c1.z + m2.y * c2.z - m * c.y * c.z;
        I.I[5] = m0.z * c0.z + m1.z * c1.z + m2.z * c2.z + m * (cc - c.z * c.z);
        const SV<T> fbias = xr::bias_force(I, v, a);
        const T vals[16] = {I.m, I.h.x, I.h.y, I.h.z, I.I[0], I.I[1], I.I[2], I.I[3], I.I[4], I.I[5],
                            fbias.w.x, fbias.w.y, fbias.w.z, fbias.v.x, fbias.v.y, fbias.v.z};
#pragma unroll
        for (int k = 0; k < 16; k++) comp[k].v[i_] = l < ND ? vals[k] : (T)0;
    }
    // ---- composites of the subtrees: gripper links leaf to root (children have higher indices than their parents)...
    static_assert(xmr::G_PARENT[0] == -1 && xmr::G_PARENT[1] == 0 && xmr::G_PARENT[2] == -1 && xmr::G_PARENT[3] == -1 &&
                  xmr::G_PARENT[4] == 3 && xmr::G_PARENT[5] == -1, "fold order below is written for this gripper tree");
    fold<12, 6>(G, comp); fold<11, 10>(G, comp); fold<10, 6>(G, comp); fold<9, 6>(G, comp); fold<8, 7>(G, comp); fold<7, 6>(G, comp);
    // ... then the arm chain: suffix sums over lanes 0..6 (lane 6 already carries the whole gripper)
    {
        LV<T> arm[16], scan[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            XC_LANES arm[k].v[i_] = lane_of(G, i_) < 7 ? comp[k].v[i_] : (T)0;
            scan[k] = lv_suffix_sum(arm[k]);
            XC_LANES comp[k].v[i_] = lane_of(G, i_) < 7 ? scan[k].v[i_] : comp[k].v[i_];
        }
    }
    // ---- per lane: row l of the joint-space inertia (entries of ancestors-or-self only) and the bias torque
    LV<T> Mrow[ND], taul;
    XC_LANES {
        const int l = lane_of(G, i_);
        unsigned anc = 0u;
#pragma unroll
        for (int b = 0; b < ND; b++) anc = l == b ? anc_mask(b) : anc;
        RBI<T> Ic;
        SV<T> fc, Sl;
        Ic.m = comp[0].v[i_];
        Ic.h = mk<T>(comp[1].v[i_], comp[2].v[i_], comp[3].v[i_]);
#pragma unroll
        for (int k = 0; k < 6; k++) Ic.I[k] = comp[4 + k].v[i_];
        fc.w = mk<T>(comp[10].v[i_], comp[11].v[i_], comp[12].v[i_]);
        fc.v = mk<T>(comp[13].v[i_], comp[14].v[i_], comp[15].v[i_]);
        Sl.w = mk<T>(cap[24].v[i_], cap[25].v[i_], cap[26].v[i_]);
        Sl.v = mk<T>(cap[27].v[i_], cap[28].v[i_], cap[29].v[i_]);
        const SV<T> F = xk::rbi_mul(Ic, Sl);
#pragma unroll
        for (int i = 0; i < ND; i++) Mrow[i].v[i_] = ((anc >> i) & 1u) ? xk::sdot(S[i], F) : (T)0;
        taul.v[i_] = -xk::sdot(Sl, fc) - C.damping.v[i_] * cap[31].v[i_];
    }
    // ---- rows and torques to every lane; Cholesky factor (lower, in M) with reciprocal diagonal, redundantly
    T M[91], tau[ND], rd[ND];
#define XRC_ROW(r)                                                                                           \
    {                                                                                                        \
        _Pragma("unroll") for (int c = 0; c <= r; c++) M[tri(r, c)] = lv_get<r>(Mrow[c]);                    \
        tau[r] = lv_get<r>(taul);                                                                            \
    }
    XRC_ROW(0) XRC_ROW(1) XRC_ROW(2) XRC_ROW(3) XRC_ROW(4) XRC_ROW(5) XRC_ROW(6) XRC_ROW(7) XRC_ROW(8) XRC_ROW(9) XRC_ROW(10) XRC_ROW(11) XRC_ROW(12)
#undef XRC_ROW
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
    // ---- lane l solves M x = e_l: x = row l of M^-1 (forward and back substitution with the shared factor)
    LV<T> Mi[ND], dqf, diag;
    XC_LANES {
        const int l = lane_of(G, i_);
        T y[ND], x[ND];
#pragma unroll
        for (int k = 0; k < ND; k++) {
            T s = k == l ? (T)1 : (T)0;
#pragma unroll
            for (int j = 0; j < k; j++) s -= M[tri(k, j)] * y[j];
            y[k] = s * rd[k];
        }
#pragma unroll
        for (int k = ND - 1; k >= 0; k--) {
            T s = y[k];
#pragma unroll
            for (int j = k + 1; j < ND; j++) s -= M[tri(j, k)] * x[j];
            x[k] = s * rd[k];
        }
        T acc_ = (T)0, dg = (T)1;
#pragma unroll
        for (int k = 0; k < ND; k++) { Mi[k].v[i_] = x[k]; acc_ += x[k] * tau[k]; dg = k == l ? x[k] : dg; }
        dqf.v[i_] = cap[31].v[i_] + dt * acc_;
        diag.v[i_] = dg;
    }
    // ---- rows owned by lane l: motor l (x half) and limit l (y half); -A entries as pairs per column
    const T m_hi = (T)(xmr::MOTOR_FORCE * xmr::TIME_STEP);
    LV<T> gm, gl, invm, invl, sgl;
    XC_LANES {
        const int l = lane_of(G, i_);
        const bool own = l < ND;
        const T q = cap[30].v[i_], qt = cap[32].v[i_], dq0 = dqf.v[i_];
        const T m_vt = (T)xm::MOTOR_KP * (qt - q) * idt + (T)(1.0 - xm::MOTOR_KD) * dq0;
        const T g0 = q - C.lo.v[i_], g1 = C.hi.v[i_] - q;
        const bool lo = g0 < (T)xm::LIMIT_WINDOW, hi = g1 < (T)xm::LIMIT_WINDOW;
        const T gg = lo ? g0 : g1;
        const T sg = !own ? (T)0 : (lo ? (T)1 : (hi ? (T)-1 : (T)0));
        const T l_vt = gg < (T)0 ? -(T)xm::GLOBAL_ERP * gg * idt : -gg * idt;
        const T inv = own ? (T)1 / diag.v[i_] : (T)0;
        gm.v[i_] = own ? m_vt - dq0 : (T)0;
        gl.v[i_] = l_vt - sg * dq0;
        invm.v[i_] = inv;
        invl.v[i_] = sg != (T)0 ? inv : (T)0;
        sgl.v[i_] = sg;
    }
    // the sweep carries c = lam + g / d per row (xarm_coop_core.h sweep_all: four dependent instructions per row step
    // instead of five); the columns are scaled by the receiving row's 1 / d, a row's own entry is zero.  Motor r and
    // limit r push the same joint but are different rows of lane r: one column pair per kind.
    const LV2<T> invd01 = xc::lv2_make(invm, invl), zero2 = xc::lv2_make(lv_fill((T)0), lv_fill((T)0));
    LV2<T> lam01 = zero2, c01 = xc::lv2_mul(xc::lv2_make(gm, gl), invd01), AM[ND], AL[ND];
#define XRC_COL(r)                                                                                           \
    {                                                                                                        \
        const LV2<T> col = xc::lv2_mul(xc::lv2_make(xc::lv_neg(Mi[r]), xc::lv_neg(xc::lv_mul(sgl, Mi[r]))), invd01); \
        LV<T> cx = xc::lv2_x(col), cy = xc::lv2_y(col);                                                      \
        AM[r] = col; AL[r] = col;                                                                            \
        lv_commit<r>(G, cx, lv_fill((T)0));                                                                  \
        lv_commit<r>(G, cy, lv_fill((T)0));                                                                  \
        AM[r] = xc::lv2_make(cx, xc::lv2_y(col));                                                            \
        AL[r] = xc::lv2_make(xc::lv2_x(col), cy);                                                            \
    }
    XRC_COL(0) XRC_COL(1) XRC_COL(2) XRC_COL(3) XRC_COL(4) XRC_COL(5) XRC_COL(6) XRC_COL(7) XRC_COL(8) XRC_COL(9) XRC_COL(10) XRC_COL(11) XRC_COL(12)
#undef XRC_COL
    bool lim_w[ND];   // limit row r is inside its window for some environment of the wavefront
#define XRC_LIMW(r) lim_w[r] = XARM_ANY_X(lv_get<r>(sgl) != (T)0);
    XRC_LIMW(0) XRC_LIMW(1) XRC_LIMW(2) XRC_LIMW(3) XRC_LIMW(4) XRC_LIMW(5) XRC_LIMW(6) XRC_LIMW(7) XRC_LIMW(8) XRC_LIMW(9) XRC_LIMW(10) XRC_LIMW(11) XRC_LIMW(12)
#undef XRC_LIMW
    const LV<T> mhi = lv_fill(m_hi), mlo = lv_fill(-m_hi);
#pragma unroll 1
    for (int it = 0; it < XC_SWEEP_ITERS; it++) {
#define XRC_MOTOR(r)                                                                                         \
        {                                                                                                    \
            const LV<T> lam = xc::lv2_x(lam01);                                                              \
            const LV<T> nl = xc::lv_med3(xc::lv2_x(c01), mlo, mhi);                                          \
            const LV<T> dl = xc::lv_sub(nl, lam);                                                            \
            LV<T> nlam = lam;                                                                                \
            lv_commit<r>(G, nlam, nl);                                                                       \
            lam01 = xc::lv2_make(nlam, xc::lv2_y(lam01));                                                    \
            const LV<T> b = lv_bcast<r>(dl);                                                                 \
            c01 = xc::lv2_fma(AM[r], xc::lv2_make(b, b), c01);                                               \
        }
#define XRC_LIMIT(r)                                                                                         \
        if (lim_w[r]) {                                                                                      \
            const LV<T> lam = xc::lv2_y(lam01);                                                              \
            const LV<T> nl = xc::lv_max0(xc::lv2_y(c01));                                                    \
            const LV<T> dl = xc::lv_mul(xc::lv_sub(nl, lam), sgl);   /* impulse on the joint: sg * d lambda */ \
            LV<T> nlam = lam;                                                                                \
            lv_commit<r>(G, nlam, nl);                                                                       \
            lam01 = xc::lv2_make(xc::lv2_x(lam01), nlam);                                                    \
            const LV<T> b = lv_bcast<r>(dl);                                                                 \
            c01 = xc::lv2_fma(AL[r], xc::lv2_make(b, b), c01);                                               \
        }
        XRC_MOTOR(0) XRC_MOTOR(1) XRC_MOTOR(2) XRC_MOTOR(3) XRC_MOTOR(4) XRC_MOTOR(5) XRC_MOTOR(6) XRC_MOTOR(7) XRC_MOTOR(8) XRC_MOTOR(9)
        XRC_MOTOR(10) XRC_MOTOR(11) XRC_MOTOR(12)
        XRC_LIMIT(0) XRC_LIMIT(1) XRC_LIMIT(2) XRC_LIMIT(3) XRC_LIMIT(4) XRC_LIMIT(5) XRC_LIMIT(6) XRC_LIMIT(7) XRC_LIMIT(8) XRC_LIMIT(9)
        XRC_LIMIT(10) XRC_LIMIT(11) XRC_LIMIT(12)
#undef XRC_MOTOR
#undef XRC_LIMIT
    }
    // ---- constrained joint rates: dq_l = dq_free_l + sum_r Minv[l][r] * (lam_motor_r + sg_r lam_limit_r); integrate; publish
    const LV<T> tsum = xc::lv_fma(sgl, xc::lv2_y(lam01), xc::lv2_x(lam01));
    LV<T> dq = dqf;
#define XRC_DQ(r) dq = xc::lv_fma(Mi[r], lv_bcast<r>(tsum), dq);
    XRC_DQ(0) XRC_DQ(1) XRC_DQ(2) XRC_DQ(3) XRC_DQ(4) XRC_DQ(5) XRC_DQ(6) XRC_DQ(7) XRC_DQ(8) XRC_DQ(9) XRC_DQ(10) XRC_DQ(11) XRC_DQ(12)
#undef XRC_DQ
    LV<T> qn;
    XC_LANES qn.v[i_] = cap[30].v[i_] + dt * dq.v[i_];
#define XRC_PUB(r) st.qd[r] = lv_get<r>(dq); st.q[r] = lv_get<r>(qn);
    XRC_PUB(0) XRC_PUB(1) XRC_PUB(2) XRC_PUB(3) XRC_PUB(4) XRC_PUB(5) XRC_PUB(6) XRC_PUB(7) XRC_PUB(8) XRC_PUB(9) XRC_PUB(10) XRC_PUB(11) XRC_PUB(12)
#undef XRC_PUB
}

template <typename T> XARM_HD void sim_tick(const Grp &G, const BodyLane<T> &C, EnvState<T> &st) {
    const T dt = (T)(xmr::TIME_STEP / xmr::N_SUBSTEPS);
#pragma unroll 1
    for (int k = 0; k < xmr::N_SUBSTEPS; k++) substep<T>(G, C, st, dt);
}

// hipcc 7.2's register allocator crashes (VirtRegAuxInfo::isRematerializable, "copy from non-existing value") on the
// reset kernel when the first tick starts from values that do not depend on the loaded state (constants, or constants
// behind an opaque asm move: both crash; -amdgpu-sched-strategy=iterative-ilp with -fno-slp-vectorize only).  The select
// below ties the teleported pose to the loaded one behind a condition that is never true at run time (steps >= 0 always;
// a NaN compares false), at the cost of 26 v_cndmask per reset.
// XarmReachEnv.reset (:96-102), same sequence as xr::env_reset
template <typename T> XARM_HD void env_reset(const Grp &G, const EnvCfg &cfg, int64_t env, EnvState<T> &st, T (&obs)[xr::OBS_DIM]) {
    const BodyLane<T> C = body_lane_consts<T>(G);
    const int64_t episode = (int64_t)st.episode + 1;
    const bool keep = st.steps < (T)-1;
#pragma unroll
    for (int i = 0; i < ND; i++) { st.q[i] = keep ? st.q[i] : (T)xmr::JOINT_INIT_POS[i]; st.qd[i] = keep ? st.qd[i] : (T)0; }
    sim_tick<T>(G, C, st);
    xr::sample_goal(cfg, env, episode, st);
    xr::get_obs(st, obs);
    st.d_old = xr::goal_dist(st, obs);
    st.steps = (T)0;
    st.episode = (T)episode;
}

// XarmReachEnv.step (:81-94), same sequence as xr::env_step
template <typename T>
XARM_HD void env_step(const Grp &G, const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&obs)[xr::OBS_DIM], T &reward, bool &done,
                      bool &success, int &future_length) {
    const BodyLane<T> C = body_lane_consts<T>(G);
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
    const T g = st.q[xmr::DRIVER_DOF] + a[3] * (T)(xmr::ACTION_DT * xmr::MAX_GRIPPER_VEL);
    T qa[7], qo[7];
#pragma unroll
    for (int i = 0; i < 7; i++) qa[i] = st.q[i];
    xk::ik_arm<T, xmr::N_SUBSTEPS>(qa, target, qo);
#pragma unroll
    for (int i = 0; i < ND; i++) st.qt[i] = i < 7 ? qo[i < 7 ? i : 0] : g;
    sim_tick<T>(G, C, st);
    xr::get_obs(st, obs);
    const T dist = xr::goal_dist(st, obs);
    success = dist < (T)xmr::DISTANCE_THRESHOLD;
    if (cfg.reward_type == 0) reward = success ? (T)1 : (T)0;
    else if (cfg.reward_type == 1) reward = -dist;
    else { reward = st.d_old - dist; st.d_old = dist; }
    done = (int)st.steps == xmr::MAX_EPISODE_STEPS;
    future_length = xmr::MAX_EPISODE_STEPS - (int)st.steps;
}

} // namespace xrc
