// xarm_handover_coop_core.h - cooperative form of the XarmHandover substep: TWO 16-lane rows per environment, row = arm.
//
// Reference: /root/reference/gym_xarm/envs/xarm_handover.py (step :128-139, _set_action :244-297, reset :141-145,338-393);
// same sequence as xarm_handover_core.h (xh::lane_step / lane_reset), which maps an environment to a lane PAIR.
//
// Why it exists: k_ho_step runs 16 384 environments as 512 wavefronts of a ~450-register kernel and a
// wavefront with ONE lane whose finger pads are near the stick sweeps the pad blocks for all 32 of its environments
// (2.3 ms per step against 0.96 ms for a contact-free batch).  The step is therefore split as for PickAndPlace: every
// environment runs the pad-free fast lane-pair step (xh::lane_step_fast) and the few with an active pad row are handed
// off, untouched, to this core, where an environment owns two DPP rows of 16 lanes - row 0 = arm 0, row 1 = arm 1 - and
// each row is the single-arm impulse-space core of xarm_coop_core.h (xc::) instantiated for the Handover scene:
//   * the arm dynamics, the 14 single-joint rows (M L G), the arm-limit rows and the 12 pad rows of arm k live in row k;
//   * the object and its 12 support rows (T) are replicated in both rows;
//   * with pad rows on ONE arm only (all but a handful of the handed-off environments) nothing couples the rows inside
//     the sweep: the row of the touching arm runs exactly xc::sweep_all (T, (M L G), F), the other row's arm rows are
//     independent of the object, and the object is taken from the touching row at the end of the substep;
//   * with pad rows on BOTH arms the sweep keeps the oracle's order T, (M L G)_0, (M L G)_1, F_0, F_1: the pair steps
//     T_i || A_i run in both rows on bit-identical copies of the T rows, every pad-row impulse of arm k is broadcast
//     inside row k (DPP row_newbcast) and handed to the other row by ONE v_permlane32_swap (gfx950), where it moves the
//     T copy and the pad rows of the other arm through the object-only cross columns of the Delassus matrix.
// Which of the two forms runs is a wave-uniform choice; the cross terms are exact no-ops for an environment with one
// touching arm, so an environment's result does not depend on the other environment of its wavefront
// (tests/test_handover_coop.py: forced-coupled == natural, bitwise).
//
// Exchange interface X (device: v_permlane32_swap between lanes l and l + 32; host: two threads and a slot):
//   int arm;  T from0(T) / from1(T) / partner(T);  void pair(T v, T &v0, T &v1);  void both(LV<T> v, LV<T> &v0, LV<T> &v1)
#pragma once
#include "xarm_coop_core.h"
#include "xarm_handover_core.h"

namespace xhc {
using xk::V3; using xk::mk; using xk::dot; using xk::cross; using xk::symmul; using xk::symi; using xk::EnvState;
using xk::NTS; using xk::NP; using xk::xsqrt;
using xc::Grp; using xc::LV; using xc::LV2; using xc::Setup; using xc::Sweep; using xc::ArmLane; using xc::lane_of;
using xc::lv_fill; using xc::lv_fill_vgpr; using xc::lv_fma; using xc::lv_mul; using xc::lv_sub; using xc::lv_neg; using xc::lv_max0; using xc::lv_med3;
using xc::lv_bcast; using xc::lv_get; using xc::lv_allsum; using xc::lv_commit;
using xc::lv2_make; using xc::lv2_x; using xc::lv2_y; using xc::lv2_fma; using xc::lv2_sub; using xc::lv2_mul; using xc::lv2_commit; using xc::lv2_bcast;
using xc::LVN; using xc::NT; using xc::NA1; using xc::NF; using xc::NLA; using xc::R_G; using xc::R_J0; using xc::R_J1; using xc::R_J2A; using xc::R_J2B; using xc::R_J3;
using xc::C0_T; using xc::C0_F; using xc::C1_A; using xc::C1_L; using xc::C1_F; using xc::C2_T; using xc::C2_A; using xc::C2_L; using xc::C2_F;

constexpr int ROW_ENVS = 2;     // environments per 64-lane wavefront: rows 0, 1 = arm 0 of env slots 0, 1; rows 2, 3 = arm 1

template <typename T> XARM_HD LV<T> lv_sel(bool c, LV<T> a, LV<T> b) { LV<T> r; XC_LANES r.v[i_] = c ? a.v[i_] : b.v[i_]; return r; }
template <typename T> XARM_HD LV2<T> lv2_sel(bool c, LV2<T> a, LV2<T> b) { return lv2_make(lv_sel(c, lv2_x(a), lv2_x(b)), lv_sel(c, lv2_y(a), lv2_y(b))); }
// dst = src on lane L of the rows for which `mine` holds
template <int L, typename T> XARM_HD void lv_commit_if(const Grp &G, bool mine, LV<T> &dst, LV<T> src) {
    XC_LANES dst.v[i_] = (mine && lane_of(G, i_) == L) ? src.v[i_] : dst.v[i_];
}

// generalized impulses of a solved substep: joints of this row's arm, object part of the support rows, object part of
// this row's pad rows (kept apart: the pad parts of the two arms are summed in arm order by both rows)
template <typename T, bool PAD, bool LA>
XARM_HD void reduce(const Sweep<T> &W, const LV<T> (&J)[R_G], T (&tauJ)[9], T (&pT)[6], T (&pF)[6]) {
#pragma unroll
    for (int d = 0; d < 9; d++) {
        LV<T> c = lv_mul(J[R_J1 + d], W.lam[1]);
        if (PAD) c = lv_fma(J[R_J2A + d], W.lam[2], c);
        if (LA && d < 7) c = lv_fma(J[R_J3 + d], W.lam[3], c);
        tauJ[d] = lv_allsum(c);
    }
#pragma unroll
    for (int d = 0; d < 6; d++) {
        pT[d] = lv_allsum(lv_mul(J[R_J0 + d], W.lam[0]));
        pF[d] = PAD ? lv_allsum(lv_mul(J[R_J2B + d], W.lam[2])) : (T)0;
    }
}

// the rows are decoupled inside the sweep (at most one arm of the environment has pad rows): xc::solve with the Handover
// scene and the object part of the impulse reduced per row set
template <typename T, typename Lds, bool PAD, bool LA, typename Scene>
XARM_HD void solve(const Grp &G, const Setup<T> &S, const bool (&padw)[NP], Lds lds, Sweep<T> &W, LV<T> (&J)[R_G], T (&tauJ)[9], T (&pT)[6], T (&pF)[6]) {
    LV<T> cfm = lv_fill((T)0);
    if (PAD || LA) xc::build_extra<T, Lds, PAD, LA, Scene>(G, S, lds, W, J, cfm);
    bool pw[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) pw[p] = PAD && padw[p];
    if (PAD) xc::pad_columns<T, LA, Scene>(G, S, J, cfm, W, pw);
    xc::apply_warm_start<T, PAD, LA>(W, pw);
    const T mu_t = (T)(xm::MU_OBJECT * xm::MU_TABLE);
    xc::sweep_all<T, PAD, LA>(G, W, mu_t, S.mu_p, pw);
    reduce<T, PAD, LA>(W, J, tauJ, pT, pF);
}

// ---------------------------------------------------------------------------------------------
// Both arms hold pad rows.  Own columns as in the decoupled form; cross columns (object part only) from the other
// row's M^-1 J^T; warm start and sweep in arm order.  Every operation a row performs on its own rows is the operation
// of xc::apply_warm_start / xc::sweep_all, and a cross term whose impulse is zero changes nothing: for an environment
// with one touching arm this is bit for bit the decoupled form.
template <typename T, bool LA, typename X>
XARM_HD void sweep_coupled(const Grp &G, const X &x, Sweep<T> &W, LV<T> (&XT)[NF], LV<T> (&XF)[NF], T mu_t, T mu_p, const bool (&padw)[NP],
                           const bool (&padk)[2][NP]) {
    const bool arm0 = x.arm == 0;
    LV2<T> lam01 = lv2_make(W.lam[0], W.lam[1]);
    const LV2<T> invd01 = lv2_make(W.invd[0], W.invd[1]), zero2 = lv2_make(lv_fill((T)0), lv_fill((T)0));
    LV2<T> c01 = lv2_fma(lv2_make(W.g[0], W.g[1]), invd01, lam01);
    LV2<T> A01[NA1];
#define XH_COL01(i)                                                                                          \
    A01[i] = lv2_mul(lv2_make((i) < NT ? W.nA0[C0_T + (i)] : lv_fill((T)0), W.nA1[C1_A + (i)]), invd01);    \
    lv2_commit<i>(G, A01[i], zero2);
    XH_COL01(0) XH_COL01(1) XH_COL01(2) XH_COL01(3) XH_COL01(4) XH_COL01(5) XH_COL01(6) XH_COL01(7) XH_COL01(8) XH_COL01(9)
    XH_COL01(10) XH_COL01(11) XH_COL01(12) XH_COL01(13)
#undef XH_COL01
    LV<T> AL1[7];
    if (LA) {
#pragma unroll
        for (int i = 0; i < 7; i++) AL1[i] = lv_mul(W.nA1[C1_L + i], W.invd[1]);
    }
    LV<T> c2 = lv_fma(W.g[2], W.invd[2], W.lam[2]), c3 = lv_fill((T)0);
#pragma unroll
    for (int i = 0; i < NT; i++) W.nA2[C2_T + i] = lv_mul(W.nA2[C2_T + i], W.invd[2]);
#pragma unroll
    for (int i = 0; i < NA1; i++) W.nA2[C2_A + i] = lv_mul(W.nA2[C2_A + i], W.invd[2]);
    if (LA) {
#pragma unroll
        for (int i = 0; i < 7; i++) W.nA2[C2_L + i] = lv_mul(W.nA2[C2_L + i], W.invd[2]);
    }
    // own pad columns as in xc::sweep_all (scaled by the receiving row's 1 / d, a row's own entry zeroed); the cross columns -
    // support rows and pad rows against the OTHER arm's pad rows, through the object - likewise.  No per-phase copies: in
    // phase k a row of arm k uses its own columns, the other row the cross columns, chosen per row step (selects on the
    // column instead of 72 more registers of per-phase copies, which pushed ~110 B/lane more of the substep's state into
    // scratch - also on the decoupled path every other environment takes)
    LV2<T> AF01[NF];
#define XH_COLF(r)                                                                                           \
    if (padw[(r) / 3]) {                                                                                     \
        AF01[r] = lv2_mul(lv2_make(W.nA0[C0_F + r], W.nA1[C1_F + r]), invd01);                               \
        W.nA2[C2_F + r] = lv_mul(W.nA2[C2_F + r], W.invd[2]);                                                \
        lv_commit<r>(G, W.nA2[C2_F + r], lv_fill((T)0));                                                     \
        if (LA) W.nA3[C1_F + r] = lv_mul(W.nA3[C1_F + r], W.invd[3]);                                        \
        XT[r] = lv_mul(XT[r], W.invd[0]);                                                                    \
        XF[r] = lv_mul(XF[r], W.invd[2]);                                                                    \
    }
    XH_COLF(0) XH_COLF(1) XH_COLF(2) XH_COLF(3) XH_COLF(4) XH_COLF(5) XH_COLF(6) XH_COLF(7) XH_COLF(8) XH_COLF(9) XH_COLF(10) XH_COLF(11)
#undef XH_COLF
    if (LA) {
        c3 = lv_fma(W.g[3], W.invd[3], W.lam[3]);
#pragma unroll
        for (int i = 0; i < NA1; i++) W.nA3[C1_A + i] = lv_mul(W.nA3[C1_A + i], W.invd[3]);
#define XH_COLL(i)                                                                                           \
        W.nA3[C1_L + i] = lv_mul(W.nA3[C1_L + i], W.invd[3]);                                                \
        lv_commit<i>(G, W.nA3[C1_L + i], lv_fill((T)0));
        XH_COLL(0) XH_COLL(1) XH_COLL(2) XH_COLL(3) XH_COLL(4) XH_COLL(5) XH_COLL(6)
#undef XH_COLL
    }
    const LV<T> mu_tv = lv_fill_vgpr(mu_t), mu_pv = lv_fill(mu_p);
    bool law[NLA];
#define XH_LAW(i) law[i] = LA && XARM_ANY_X(lv_get<i>(W.invd[3]) != (T)0);
    XH_LAW(0) XH_LAW(1) XH_LAW(2) XH_LAW(3) XH_LAW(4) XH_LAW(5) XH_LAW(6)
#undef XH_LAW
#pragma unroll 1
    for (int it = 0; it < XC_SWEEP_ITERS; it++) {
        LV<T> lim = lv_fill((T)0);
#define XH_PAIR(i)                                                                                           \
        {                                                                                                    \
            LV<T> nx = lv2_x(c01);                                                                           \
            if ((i) % 3 == 0 || (i) >= NT) nx = lv_max0(nx);                                                 \
            else nx = lv_med3(nx, lv_neg(lim), lim);                                                         \
            const LV2<T> nl = lv2_make(nx, lv_med3(lv2_y(c01), W.lo1, W.hi1));                               \
            const LV2<T> dl = lv2_sub(nl, lam01);                                                            \
            lv2_commit<i>(G, lam01, nl);                                                                     \
            if ((i) % 3 == 0 && (i) < NT) lim = lv_mul(lv_bcast<i>(nx), mu_tv);                              \
            const LV2<T> b = lv2_bcast<i>(dl);                                                               \
            c01 = lv2_fma(A01[i], b, c01);                                                                   \
            if ((i) < NT) c2 = lv_fma(W.nA2[C2_T + (i)], lv2_x(b), c2);                                      \
            c2 = lv_fma(W.nA2[C2_A + (i)], lv2_y(b), c2);                                                    \
            if (LA) c3 = lv_fma(W.nA3[C1_A + (i)], lv2_y(b), c3);                                            \
        }
#define XH_L_ROW(i)                                                                                          \
        if (law[i]) {                                                                                        \
            const LV<T> nl = lv_max0(c3);                                                                    \
            const LV<T> dl = lv_sub(nl, W.lam[3]);                                                           \
            lv_commit<i>(G, W.lam[3], nl);                                                                   \
            const LV<T> b = lv_bcast<i>(dl);                                                                 \
            c01 = lv2_make(lv2_x(c01), lv_fma(AL1[i], b, lv2_y(c01)));                                       \
            c2 = lv_fma(W.nA2[C2_L + i], b, c2);                                                             \
            c3 = lv_fma(W.nA3[C1_L + i], b, c3);                                                             \
        }
        // pad row (p, a) of arm k: solved by row k, its impulse change handed to both rows
#define XH_F_ROW(k, p, a)                                                                                    \
        {                                                                                                    \
            LV<T> nl;                                                                                        \
            if ((a) == 0) nl = lv_max0(c2);                                                                  \
            else {                                                                                           \
                const LV<T> flim = lv_mul(lv_bcast<3 * p>(W.lam[2]), mu_pv);                                 \
                nl = lv_med3(c2, lv_neg(flim), flim);                                                        \
            }                                                                                                \
            const LV<T> dl = lv_sub(nl, W.lam[2]);                                                           \
            const bool mine = arm0 == ((k) == 0);                                                            \
            lv_commit_if<3 * p + a>(G, mine, W.lam[2], nl);                                                  \
            LV<T> b0, b1;                                                                                    \
            x.both(lv_bcast<3 * p + a>(dl), b0, b1);                                                         \
            const LV<T> b = (k) == 0 ? b0 : b1, zero = lv_fill((T)0);                                        \
            const LV2<T> col01 = lv2_sel(mine, AF01[3 * p + a], lv2_make(XT[3 * p + a], zero));              \
            c01 = lv2_fma(col01, lv2_make(b, b), c01);                                                       \
            c2 = lv_fma(lv_sel(mine, W.nA2[C2_F + 3 * p + a], XF[3 * p + a]), b, c2);                        \
            if (LA) c3 = lv_fma(lv_sel(mine, W.nA3[C1_F + 3 * p + a], zero), b, c3);                         \
        }
#define XH_F_PAD(k, p) if (padk[k][p]) { XH_F_ROW(k, p, 0) XH_F_ROW(k, p, 1) XH_F_ROW(k, p, 2) }   /* pad p of arm k is live somewhere in the wavefront */
        XH_PAIR(0) XH_PAIR(1) XH_PAIR(2) XH_PAIR(3) XH_PAIR(4) XH_PAIR(5) XH_PAIR(6) XH_PAIR(7) XH_PAIR(8)
        if (LA) { XH_L_ROW(0) XH_L_ROW(1) XH_L_ROW(2) XH_L_ROW(3) XH_L_ROW(4) XH_L_ROW(5) XH_L_ROW(6) }
        XH_PAIR(9) XH_PAIR(10) XH_PAIR(11) XH_PAIR(12) XH_PAIR(13)
        XH_F_PAD(0, 0) XH_F_PAD(0, 1) XH_F_PAD(0, 2) XH_F_PAD(0, 3)
        XH_F_PAD(1, 0) XH_F_PAD(1, 1) XH_F_PAD(1, 2) XH_F_PAD(1, 3)
#undef XH_PAIR
#undef XH_L_ROW
#undef XH_F_ROW
#undef XH_F_PAD
    }
    W.lam[0] = lv2_x(lam01);
    W.lam[1] = lv2_y(lam01);
}

template <typename T, typename Lds, bool LA, typename X, typename Scene>
XARM_HD void solve_coupled(const Grp &G, const X &x, const Setup<T> &S, const bool (&padw)[NP], const bool (&padk)[2][NP], Lds lds, Sweep<T> &W, LV<T> (&J)[R_G],
                           T (&tauJ)[9], T (&pT)[6], T (&pF)[6]) {
    const bool arm0 = x.arm == 0;
    LV<T> cfm = lv_fill((T)0);
    xc::build_extra<T, Lds, true, LA, Scene>(G, S, lds, W, J, cfm);
    xc::pad_columns<T, LA, Scene>(G, S, J, cfm, W, padw);
    // cross columns: the other row's M^-1 J_r^T, object part (lane r of the other row holds its pad row r)
    LV<T> XT[NF], XF[NF];
    {
        LV<T> Ba[9], Bb[6], Bo[6];
        xc::pad_minv_jt<T, Scene>(S, J, Ba, Bb);
#pragma unroll
        for (int d = 0; d < 6; d++) {
            LV<T> b0, b1;
            x.both(Bb[d], b0, b1);
            Bo[d] = lv_sel(arm0, b1, b0);
        }
#define XH_CROSS_COL(r)                                                                                      \
        if (padw[(r) / 3]) {                                                                                 \
            LV<T> cb[6];                                                                                     \
            _Pragma("unroll") for (int d = 0; d < 6; d++) cb[d] = lv_bcast<r>(Bo[d]);                        \
            LV<T> a0 = lv_mul(J[R_J0], cb[0]), a2 = lv_mul(J[R_J2B], cb[0]);                                 \
            _Pragma("unroll") for (int d = 1; d < 6; d++) { a0 = lv_fma(J[R_J0 + d], cb[d], a0); a2 = lv_fma(J[R_J2B + d], cb[d], a2); } \
            XT[r] = lv_neg(a0);                                                                              \
            XF[r] = lv_neg(a2);                                                                              \
        }
        XH_CROSS_COL(0) XH_CROSS_COL(1) XH_CROSS_COL(2) XH_CROSS_COL(3) XH_CROSS_COL(4) XH_CROSS_COL(5)
        XH_CROSS_COL(6) XH_CROSS_COL(7) XH_CROSS_COL(8) XH_CROSS_COL(9) XH_CROSS_COL(10) XH_CROSS_COL(11)
#undef XH_CROSS_COL
    }
    // warm start, g -= A lam0, in the order T, F_0, F_1 in both rows
    {
        LV<T> l0, l1;
        x.both(W.lam[2], l0, l1);
        const LV<T> olam = lv_sel(arm0, l1, l0), zero = lv_fill((T)0);
#define XH_WS_T(s)                                                                                           \
        {                                                                                                    \
            const LV<T> b = lv_bcast<3 * s>(W.lam[0]);                                                       \
            W.g[0] = lv_fma(W.nA0[C0_T + 3 * s], b, W.g[0]);                                                \
            W.g[2] = lv_fma(W.nA2[C2_T + 3 * s], b, W.g[2]);                                                \
        }
        XH_WS_T(0) XH_WS_T(1) XH_WS_T(2) XH_WS_T(3)
#undef XH_WS_T
#define XH_WS_F(k, p)                                                                                        \
        if (padk[k][p]) {                                                                                    \
            const bool mine = arm0 == ((k) == 0);                                                            \
            const LV<T> b = lv_sel(mine, lv_bcast<3 * p>(W.lam[2]), lv_bcast<3 * p>(olam));                  \
            W.g[0] = lv_fma(lv_sel(mine, W.nA0[C0_F + 3 * p], XT[3 * p]), b, W.g[0]);                       \
            W.g[1] = lv_fma(lv_sel(mine, W.nA1[C1_F + 3 * p], zero), b, W.g[1]);                            \
            W.g[2] = lv_fma(lv_sel(mine, W.nA2[C2_F + 3 * p], XF[3 * p]), b, W.g[2]);                       \
            if (LA) W.g[3] = lv_fma(lv_sel(mine, W.nA3[C1_F + 3 * p], zero), b, W.g[3]);                    \
        }
        XH_WS_F(0, 0) XH_WS_F(0, 1) XH_WS_F(0, 2) XH_WS_F(0, 3)
        XH_WS_F(1, 0) XH_WS_F(1, 1) XH_WS_F(1, 2) XH_WS_F(1, 3)
#undef XH_WS_F
    }
    const T mu_t = (T)(xm::MU_OBJECT * xm::MU_TABLE);
    sweep_coupled<T, LA, X>(G, x, W, XT, XF, mu_t, S.mu_p, padw, padk);
    reduce<T, true, LA>(W, J, tauJ, pT, pF);
}

// ---------------------------------------------------------------------------------------------
// one p.stepSimulation() at timeStep 1/240 (no substeps) of one environment, executed by its two 16-lane rows
template <typename T, typename Lds, typename X, typename Scene, bool FORCE_COUPLED = false>
XARM_HD void substep(const Grp &G, const X &x, const ArmLane<T> &C, EnvState<T> &st, const T (&qt)[9], Lds lds) {
    const T dt = (T)xm::HO_TIME_STEP;
    const bool arm0 = x.arm == 0;
    Setup<T> S;
    xc::substep_setup<T, Lds, Scene>(G, C, st, qt, dt, lds, S, x.arm);
    XARM_LDS_FENCE();
    Sweep<T> W;
    LV<T> J[R_G];
    xc::build_base<T, Scene>(G, S, W, J);
    // the pad flags of both arms, as a bit mask: bit p = pad p has a live row, bit 4 = any
    T m0, m1;
    {
        int m = S.pad_any ? 16 : 0;
#pragma unroll
        for (int p = 0; p < NP; p++) m |= S.pact[p] ? (1 << p) : 0;
        x.pair((T)m, m0, m1);
    }
    const int mo = (int)(arm0 ? m1 : m0);
    const bool p0 = ((int)m0 & 16) != 0, p1 = ((int)m1 & 16) != 0;
#ifdef XHC_NO_COUPLED      // timing probe only (wrong for an env with pad rows on both arms): what the coupled path costs the others
    const bool both = false;
#else
    const bool both = FORCE_COUPLED || XARM_ANY_X(p0 && p1);
#endif
    const bool pad = XARM_ANY_X(S.pad_any), la = XARM_ANY_X(S.la_any);
    // padw[p]: pad p of EITHER arm has a live row somewhere in the wavefront (columns are built for it); padk[k][p]: pad p of
    // arm k has (the rows the coupled sweep and warm start visit in phase k - a skipped row carries exactly zero impulse)
    bool padw[NP], padk[2][NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        padw[p] = FORCE_COUPLED || XARM_ANY_X(S.pact[p] || (both && (mo >> p & 1)));
        const bool other = (mo >> p & 1) != 0;
        padk[0][p] = FORCE_COUPLED || XARM_ANY_X(arm0 ? S.pact[p] : other);
        padk[1][p] = FORCE_COUPLED || XARM_ANY_X(arm0 ? other : S.pact[p]);
    }
    T tauJ[9], pT[6], pF[6], oF[6];
    if (both) {
        if (la) solve_coupled<T, Lds, true, X, Scene>(G, x, S, padw, padk, lds, W, J, tauJ, pT, pF);
        else solve_coupled<T, Lds, false, X, Scene>(G, x, S, padw, padk, lds, W, J, tauJ, pT, pF);
#pragma unroll
        for (int d = 0; d < 6; d++) oF[d] = x.partner(pF[d]);
    } else {
        if (pad) {
            if (la) solve<T, Lds, true, true, Scene>(G, S, padw, lds, W, J, tauJ, pT, pF);
            else solve<T, Lds, true, false, Scene>(G, S, padw, lds, W, J, tauJ, pT, pF);
        } else {
            if (la) solve<T, Lds, false, true, Scene>(G, S, padw, lds, W, J, tauJ, pT, pF);
            else solve<T, Lds, false, false, Scene>(G, S, padw, lds, W, J, tauJ, pT, pF);
        }
#pragma unroll
        for (int d = 0; d < 6; d++) oF[d] = (T)0;
    }
    // constrained joint velocities of this row's arm, semi-implicit Euler
#pragma unroll
    for (int r = 0; r < 9; r++) {
        T s = S.dq[r];
#pragma unroll
        for (int c = 0; c < 9; c++) s += S.Minv[symi(r, c)] * tauJ[c];
        st.qd[r] = s;
        st.q[r] += dt * s;
    }
    // object: support rows + pad rows of arm 0 + pad rows of arm 1 (this order in both rows), then the copy of the row
    // whose arm touches (the support rows of the other row never saw the pad impulses unless the sweep was coupled)
    const T imb = (T)(1.0 / Scene::OBJ_MASS);
    T tl[6];
    {
        XC_NO_CONTRACT
#pragma unroll
        for (int d = 0; d < 6; d++) tl[d] = (pT[d] + (arm0 ? pF[d] : oF[d])) + (arm0 ? oF[d] : pF[d]);
    }
    V3<T> vb = S.vb + mk<T>(tl[0], tl[1], tl[2]) * imb;
    V3<T> wb = S.wb + symmul(S.Iinv, mk<T>(tl[3], tl[4], tl[5]));
    T lt[NTS] = {lv_get<0>(W.lam[0]), lv_get<3>(W.lam[0]), lv_get<6>(W.lam[0]), lv_get<9>(W.lam[0])};
    {
        const bool src1 = !p0 && p1;
        T a0, a1;
#define XH_TAKE(v) x.pair(v, a0, a1); v = src1 ? a1 : a0;
        XH_TAKE(vb.x) XH_TAKE(vb.y) XH_TAKE(vb.z) XH_TAKE(wb.x) XH_TAKE(wb.y) XH_TAKE(wb.z)
        XH_TAKE(lt[0]) XH_TAKE(lt[1]) XH_TAKE(lt[2]) XH_TAKE(lt[3])
#undef XH_TAKE
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        T l = (T)0;
#pragma unroll
        for (int s = 0; s < NTS; s++) l = S.tid[s] == i ? lt[s] : l;
        st.lam_t[i] = l;
        st.lam_p[i] = (T)0;
    }
    if (both || pad) {
        st.lam_p[0] = lv_get<0>(W.lam[2]); st.lam_p[1] = lv_get<3>(W.lam[2]);
        st.lam_p[2] = lv_get<6>(W.lam[2]); st.lam_p[3] = lv_get<9>(W.lam[2]);
    }
    st.bp[0] += dt * vb.x; st.bp[1] += dt * vb.y; st.bp[2] += dt * vb.z;
    {
        const T idt = (T)1 / dt;
        T ang = xsqrt(dot(wb, wb));
        if (ang * dt > (T)0.7853981633974483) ang = (T)0.7853981633974483 * idt;
        T sw, cw;
        xk::xsincos((T)0.5 * ang * dt, sw, cw);
        const T k = ang < (T)0.001 ? (T)0.5 * dt - dt * dt * dt * (T)0.020833333333 * ang * ang : sw / ang;
        const V3<T> ax = wb * k;
        const T qx = st.bq[0], qy = st.bq[1], qz = st.bq[2], w0 = st.bq[3];
        const T nx = cw * qx + ax.x * w0 + ax.y * qz - ax.z * qy;
        const T ny = cw * qy + ax.y * w0 + ax.z * qx - ax.x * qz;
        const T nz = cw * qz + ax.z * w0 + ax.x * qy - ax.y * qx;
        const T nw = cw * w0 - ax.x * qx - ax.y * qy - ax.z * qz;
        const T inv = (T)1 / xsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        st.bq[0] = nx * inv; st.bq[1] = ny * inv; st.bq[2] = nz * inv; st.bq[3] = nw * inv;
    }
    st.bv[0] = vb.x; st.bv[1] = vb.y; st.bv[2] = vb.z;
    st.bw[0] = wb.x; st.bw[1] = wb.y; st.bw[2] = wb.z;
}

// XarmHandover.step (:128-139) of one environment on its two rows; L is the row's arm + its copy of the shared state.
// env_step_from runs the ticks [tick0, HO_N_TICKS): tick0 == 0 is the whole step, a later one continues the step a fast
// stage opened (xh::lane_step_fast_range) with the joint targets qt that stage computed.
template <typename T, typename Lds, typename X, typename Scene = xh::HandoverScene, bool FORCE_COUPLED = false>
XARM_HD void env_step_from(const Grp &G, const X &x, xh::Lane<T> &L, const T (&act)[4], T (&qt)[9], int tick0, T &reward, bool &done, bool &success,
                           Lds lds, int reward_type) {
    const ArmLane<T> C = xc::arm_lane_consts<T>(G);
    if (tick0 == 0) xh::step_begin(L, x.arm, act, qt);
#pragma unroll 1
    for (int k = tick0; k < xm::HO_N_TICKS; k++) substep<T, Lds, X, Scene, FORCE_COUPLED>(G, x, C, L.st, qt, lds);
    xh::step_end<T, X>(L, x.arm, reward, done, success, x, reward_type);
}
template <typename T, typename Lds, typename X, typename Scene = xh::HandoverScene, bool FORCE_COUPLED = false>
XARM_HD void env_step(const Grp &G, const X &x, xh::Lane<T> &L, const T (&act)[4], T &reward, bool &done, bool &success, Lds lds, int reward_type) {
    T qt[9];
    env_step_from<T, Lds, X, Scene, FORCE_COUPLED>(G, x, L, act, qt, 0, reward, done, success, lds, reward_type);
}

// XarmHandover.reset (:141-145, _reset_sim :338-368, _sample_goal :370-393); same sequence as xh::lane_reset
template <typename T, typename Lds, typename X, typename Scene = xh::HandoverScene, bool FORCE_COUPLED = false>
XARM_HD void env_reset(const Grp &G, const X &x, const xh::EnvCfg &cfg, int64_t env, xh::Lane<T> &L, Lds lds) {
    const ArmLane<T> C = xc::arm_lane_consts<T>(G);
    const int arm = x.arm;
    const int64_t episode = (int64_t)L.st.episode + 1;
    T qt[9], u[8];
    const V3<T> home = arm == 0 ? mk<T>((T)xm::HO_EFF_INIT_POS[0][0], (T)xm::HO_EFF_INIT_POS[0][1], (T)xm::HO_EFF_INIT_POS[0][2])
                                : mk<T>((T)xm::HO_EFF_INIT_POS[1][0], (T)xm::HO_EFF_INIT_POS[1][1], (T)xm::HO_EFF_INIT_POS[1][2]);
#pragma unroll 1
    for (int k = 0; k <= xm::HO_RESET_TICKS; k++) {
        if (k < xm::HO_RESET_TICKS) xh::ik(L, arm, home, qt);
        else {
            xh::draws(cfg, env, episode, u);
            xh::sample_object(u, L);
        }
        substep<T, Lds, X, Scene, FORCE_COUPLED>(G, x, C, L.st, qt, lds);
    }
    xh::sample_goal(cfg, u, L);
    L.st.steps = (T)0;
    L.st.episode = (T)episode;
}

} // namespace xhc
