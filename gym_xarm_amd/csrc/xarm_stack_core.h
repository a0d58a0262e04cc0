// XarmPDStackTower-v0 on the device: two xarm7_pd arms (one lane each, as in the Handover scene) and three cubes.
//
// Reference: /root/reference/gym_xarm/envs/xarm_stack_tower.py (step :101-113, _set_action :142-162, _get_obs
// :164-199, _reset_sim :201-210, _sample_goal :212-219, compute_reward :124-129).  The contact model is the one the
// CPU oracle states (oracle/xarm_oracle_stack.inc.c, model JSON "stack_tower"._contact_model): cube corners against
// the table top, cube/cube by SAT + face clipping or one edge/edge point, each pad sphere against its nearest cube.
//
// Data placement: the arm part (arm_dynamics) is the PickAndPlace code and uses the same LDS columns; the object
// contact data (12 table slots, 3 x 4 cube/cube points) lives in further lane-private LDS columns - this scene's
// BASELINE size is 8192 envs per GPU = one wavefront per CU, so a wavefront may take ~151 KB of the CU's 160 KB.
// Both lanes of an env compute the object-only rows redundantly (bit-identical); the cube velocities are handed from
// lane to lane between the two finger phases of a sweep exactly as the Handover scene hands over its stick.
#pragma once
#include "xarm_core.h"

namespace xs {
using xk::V3; using xk::mk; using xk::dot; using xk::cross; using xk::clampT; using xk::Frame; using xk::PadPoint;
using xk::tri; using xk::symi; using xk::LDS_S; using xk::LDS_T; using xk::LDS_AHH; using xk::EnvCfg;

constexpr int NOBJ = 3, NPAIR = 3;
constexpr int STATE_DIM = 136, OBS_DIM = 55, ACT_DIM = 8, GOAL_DIM = 9;
enum { K_Q = 0, K_QD = 18, K_QT = 36, K_BP = 54, K_BQ = 63, K_BV = 75, K_BW = 84, K_GOAL = 93, K_LT = 102, K_LP = 126,
       K_STEPS = 134, K_EPISODE = 135 };
// extra LDS columns behind the arm's S | T | A_hh (the arm's table-slot columns are reused)
constexpr int TP_W = 11;                      // r3 lam3 vt invd3 id
constexpr int LDS_TP = xk::LDS_TBL;           // 12 table slots
constexpr int BB_W = 22, BB_PAIR = 6 + 4 * BB_W; // per pair: n3 t1_3, then 4 x (rA3 rB3 lam3 vt invd3 Kn3 Kt1_3 Kt2_3)
constexpr int LDS_BB = LDS_TP + NOBJ * 4 * TP_W;
constexpr int LDS_CLIP = LDS_BB + NPAIR * BB_PAIR;     // 3 x 8 x 3 floats: polygon ping-pong + kept points of cube_cube
constexpr int LDS_FLOATS = LDS_CLIP + 72;              // 603 floats = 2412 B per lane, 151 KB per wavefront

struct StackScene {
    static constexpr int NARMS = 2;
    static constexpr double TIME_STEP = xm::ST_TIME_STEP;
    static constexpr double FINGER_MOTOR_FORCE = xm::ST_FINGER_MOTOR_FORCE;
    template <typename T> static XARM_HD Frame<T> base_frame(int arm) {
        const T c = arm == 0 ? (T)xm::ST_BASE_COS[0] : (T)xm::ST_BASE_COS[1];
        const T s = arm == 0 ? (T)xm::ST_BASE_SIN[0] : (T)xm::ST_BASE_SIN[1];
        Frame<T> f;
        f.c0 = mk<T>(c, s, (T)0); f.c1 = mk<T>(-s, c, (T)0); f.c2 = mk<T>((T)0, (T)0, (T)1);
        f.o = arm == 0 ? mk<T>((T)xm::ST_BASE_POS[0][0], (T)xm::ST_BASE_POS[0][1], (T)xm::ST_BASE_POS[0][2])
                       : mk<T>((T)xm::ST_BASE_POS[1][0], (T)xm::ST_BASE_POS[1][1], (T)xm::ST_BASE_POS[1][2]);
        return f;
    }
};

// one lane = one arm: q/qd/qt/lam_p are the arm's, everything else is the lane's copy of the shared state
template <typename T> struct Lane {
    T q[9], qd[9], qt[9];
    T bp[NOBJ][3], bq[NOBJ][4], bv[NOBJ][3], bw[NOBJ][3];
    T goal[NOBJ][3];
    T lam_t[NOBJ][8];
    T lam_p[4];
    T steps, episode;
    int cls;   // row-set class of the last substep (class_key below); scheduling hint, not part of the state
};

// ---- class-homogeneous wavefronts.  A wavefront sweeps the UNION of the row sets its 32 environments need (each
// wave-uniform skip is a ballot over the lanes), and the launch lasts as long as its slowest wavefront: with the envs in
// arrival order nearly every wavefront holds some env with cube/cube contact in each of the three pairs and some env with
// a finger contact, so all pay for everything (k_st_step 5.6 ms), although 88 % of the envs have no cube/cube contact
// at all and ~0.05 % have two pairs (tools/st_pairs.py).  An env's result does not depend on its neighbours (bitwise,
// tests/test_edge_cases.py), so the step kernel may visit the envs in any order: they are grouped by the row sets their
// last substep used - class key: bits 0-2 cube pairs (0,1) (0,2) (1,2) in contact, bit 3 / 4 a finger pad of arm 0 / 1
// active - and every non-empty class other than 0 starts on a wavefront boundary, topped up with class-0 envs (which
// add nothing to a union).  A stale key (the contact set changed during the step) only makes that wavefront slower.
// (Also keying on "manifold wider than two points" and skipping point slots 2-3 per wavefront was measured: no gain, the
// contacts of these axis-aligned cubes are face against face.)
constexpr int NCLS = 32;
struct ClassLayout { int start[NCLS], hole_start[NCLS], hole_len[NCLS], tail_start; bool aligned; };
// slots [0, n) for the envs of each class, from the class histogram; `group` = envs per wavefront.  Class c > 0 occupies
// [start[c], start[c] + hist[c]); the hole up to the next multiple of `group` and the tail after the last class are
// filled by class 0 in order.  Without enough class-0 envs to fill the holes: plain contiguous order (aligned = false).
XARM_HD void class_layout(const int (&hist)[NCLS], int group, ClassLayout &Y) {
    int pos = 0, holes = 0;
    for (int c = 1; c < NCLS; c++) {
        Y.start[c] = pos;
        const int end = pos + hist[c];
        const int al = hist[c] > 0 ? (end + group - 1) / group * group : end;
        Y.hole_start[c] = end; Y.hole_len[c] = al - end;
        holes += al - end;
        pos = al;
    }
    Y.aligned = holes <= hist[0];
    if (!Y.aligned) {
        pos = 0;
        for (int c = 1; c < NCLS; c++) { Y.start[c] = pos; pos += hist[c]; Y.hole_start[c] = pos; Y.hole_len[c] = 0; }
    }
    Y.start[0] = 0; Y.hole_start[0] = 0; Y.hole_len[0] = 0;
    Y.tail_start = pos;
}
// slot of the k-th env (arrival order) of class c
XARM_HD int class_slot(const ClassLayout &Y, int c, int k) {
    if (c != 0) return Y.start[c] + k;
    for (int j = 1; j < NCLS; j++) {
        if (k < Y.hole_len[j]) return Y.hole_start[j] + k;
        k -= Y.hole_len[j];
    }
    return Y.tail_start + k;
}

template <typename T> XARM_HD T sel3(int i, T a, T b, T c) { return i == 0 ? a : (i == 1 ? b : c); }
template <typename T> XARM_HD V3<T> sel3v(int i, V3<T> a, V3<T> b, V3<T> c) { return mk<T>(sel3(i, a.x, b.x, c.x), sel3(i, a.y, b.y, c.y), sel3(i, a.z, b.z, c.z)); }
using xk::selv;
template <typename T> XARM_HD V3<T> ldv(const T (&a)[3]) { return mk<T>(a[0], a[1], a[2]); }

// ---- cube/cube manifold (same algorithm and tie-breaking as box_box in oracle/xarm_oracle_stack.inc.c).
// A[k], B[k]: box axes in the world; both boxes are cubes of half edge h.  Returns the number of points (<= 4),
// normal from B to A.  Everything with a compile-time index stays in registers (all fixed loops are unrolled, the
// run-time axis choices are selects); the clipped polygon, whose vertex count is data dependent, lives in the
// lane's LDS columns [LDS_CLIP, LDS_CLIP + 72) - private (scratch) arrays cost ~2 us per dependent access here.
template <typename T, typename Lds> XARM_HD int clip_axis(Lds lds, int in, int n, int out, int axis, T sgn, T h) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        const int ia = in + 3 * i, ib = in + 3 * (i + 1 == n ? 0 : i + 1);
        const T ax = lds[ia], ay = lds[ia + 1], az = lds[ia + 2], bx = lds[ib], by = lds[ib + 1], bz = lds[ib + 2];
        const T da = sgn * (axis == 0 ? ax : ay) - h, db = sgn * (axis == 0 ? bx : by) - h;
        if (da <= (T)0) { lds[out + 3 * m] = ax; lds[out + 3 * m + 1] = ay; lds[out + 3 * m + 2] = az; m++; }
        if ((da <= (T)0) != (db <= (T)0)) {
            const T t = da / (da - db);
            lds[out + 3 * m] = ax + t * (bx - ax); lds[out + 3 * m + 1] = ay + t * (by - ay); lds[out + 3 * m + 2] = az + t * (bz - az);
            m++;
        }
    }
    return m;
}
// box_box: general half extents hA / hB (the Handover sticks, xarm_handover2_core.h); CLIP = first of the 72 LDS columns
// that hold the clipped polygon.  cube_cube below is the same code with hA = hB = (h, h, h).
template <typename T, typename Lds, int CLIP>
XARM_HD int box_box(V3<T> pA, const V3<T> (&A)[3], const T (&hA)[3], V3<T> pB, const V3<T> (&B)[3], const T (&hB)[3], T margin, V3<T> (&pts)[4], V3<T> &nrm, T (&dist)[4], Lds lds) {
    const V3<T> t = pB - pA;
    T tA[3], tB[3], C[3][3], Q[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) { tA[i] = dot(A[i], t); tB[i] = dot(B[i], t); }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { C[i][j] = dot(A[i], B[j]); Q[i][j] = xk::xabs(C[i][j]); }
    T best = (T)-1e30;
    int code = -1;
    bool sep = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const T s = xk::xabs(tA[i]) - (hA[i] + hB[0] * Q[i][0] + hB[1] * Q[i][1] + hB[2] * Q[i][2]);
        sep = sep || s > margin;
        if (s > best) { best = s; code = i; }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const T s = xk::xabs(tB[j]) - (hB[j] + hA[0] * Q[0][j] + hA[1] * Q[1][j] + hA[2] * Q[2][j]);
        sep = sep || s > margin;
        if (s > best) { best = s; code = 3 + j; }
    }
    T ebest = (T)-1e30;
    V3<T> eaxis = mk<T>(0, 0, 0), eA = mk<T>(0, 0, 0), eB = mk<T>(0, 0, 0);
    int ei = -1, ej = -1;
    T euaub = (T)0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const T l2 = (T)1 - C[i][j] * C[i][j];
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const bool ok = !(l2 < (T)1e-6);
            const T l = xk::xsqrt(ok ? l2 : (T)1);
            const T expr = tA[i2] * C[i1][j] - tA[i1] * C[i2][j];
            const T ra = hA[i1] * Q[i2][j] + hA[i2] * Q[i1][j], rb = hB[j1] * Q[i][j2] + hB[j2] * Q[i][j1];
            const T s = (xk::xabs(expr) - (ra + rb)) / l;
            sep = sep || (ok && s > margin);
            if (ok && s > ebest) {
                ebest = s; ei = i; ej = j;
                const V3<T> L = cross(A[i], B[j]);
                const T sg = (expr < (T)0 ? (T)-1 : (T)1) / l;
                eaxis = L * sg; eA = A[i]; eB = B[j]; euaub = C[i][j];
            }
        }
    // a separating axis anywhere means no contact (the oracle returns at the first one; the verdict is the same)
    if (sep) return 0;
    if (ei >= 0 && ebest - (T)1e-5 - (T)0.05 * xk::xabs(ebest) > best) {
        V3<T> pa = pA, pb = pB;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const V3<T> da = A[k] * ((dot(eaxis, A[k]) >= (T)0 ? (T)1 : (T)-1) * hA[k]), db = B[k] * ((dot(eaxis, B[k]) >= (T)0 ? (T)-1 : (T)1) * hB[k]);
            pa = k != ei ? pa + da : pa;
            pb = k != ej ? pb + db : pb;
        }
        const V3<T> p = pb - pa;
        const T q1 = dot(eA, p), q2 = -dot(eB, p), dd = (T)1 - euaub * euaub;
        const T alpha = (q1 + euaub * q2) / dd, beta = (euaub * q1 + q2) / dd;
        pa = pa + eA * alpha;
        pb = pb + eB * beta;
        pts[0] = (pa + pb) * (T)0.5;
        nrm = eaxis * (T)-1;
        dist[0] = ebest;
        return 1;
    }
    const bool refA = code < 3;
    const int ri = refA ? code : code - 3;
    V3<T> Rx[3], Ix[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { Rx[k] = selv(refA, A[k], B[k]); Ix[k] = selv(refA, B[k], A[k]); }
    const V3<T> pR = selv(refA, pA, pB), pI = selv(refA, pB, pA);
    const T tAr = sel3(ri, tA[0], tA[1], tA[2]), tBr = sel3(ri, tB[0], tB[1], tB[2]);
    const T sgR = refA ? (tAr < (T)0 ? (T)-1 : (T)1) : (tBr > (T)0 ? (T)-1 : (T)1);
    const V3<T> Rri = sel3v(ri, Rx[0], Rx[1], Rx[2]);
    const V3<T> dR = Rri * sgR;
    int jj = 0;
    T bestdot = (T)-1;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const T d = xk::xabs(dot(Ix[j], dR));
        if (d > bestdot) { bestdot = d; jj = j; }
    }
    const V3<T> Ijj = sel3v(jj, Ix[0], Ix[1], Ix[2]), Ij1 = sel3v(jj, Ix[1], Ix[2], Ix[0]), Ij2 = sel3v(jj, Ix[2], Ix[0], Ix[1]);
    const V3<T> Rr1 = sel3v(ri, Rx[1], Rx[2], Rx[0]), Rr2 = sel3v(ri, Rx[2], Rx[0], Rx[1]);
    const T sj = dot(Ijj, dR) > (T)0 ? (T)-1 : (T)1;
    // half extents of the reference / incident box along the selected axes
    const T hRi = refA ? sel3(ri, hA[0], hA[1], hA[2]) : sel3(ri, hB[0], hB[1], hB[2]);
    const T hR1 = refA ? sel3(ri, hA[1], hA[2], hA[0]) : sel3(ri, hB[1], hB[2], hB[0]);
    const T hR2 = refA ? sel3(ri, hA[2], hA[0], hA[1]) : sel3(ri, hB[2], hB[0], hB[1]);
    const T hIj = refA ? sel3(jj, hB[0], hB[1], hB[2]) : sel3(jj, hA[0], hA[1], hA[2]);
    const T hI1 = refA ? sel3(jj, hB[1], hB[2], hB[0]) : sel3(jj, hA[1], hA[2], hA[0]);
    const T hI2 = refA ? sel3(jj, hB[2], hB[0], hB[1]) : sel3(jj, hA[2], hA[0], hA[1]);
    constexpr int P0 = CLIP, P1 = CLIP + 24, KP = CLIP + 48;
    int n = 4;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const T su = (v == 0 || v == 3) ? (T)1 : (T)-1, sv = v < 2 ? (T)1 : (T)-1;
        const V3<T> w = pI + Ijj * (sj * hIj) + Ij1 * (su * hI1) + Ij2 * (sv * hI2) - pR;
        lds[P0 + 3 * v] = dot(w, Rr1);
        lds[P0 + 3 * v + 1] = dot(w, Rr2);
        lds[P0 + 3 * v + 2] = dot(w, dR) - hRi;
    }
    n = clip_axis<T, Lds>(lds, P0, n, P1, 0, (T)1, hR1);
    n = clip_axis<T, Lds>(lds, P1, n, P0, 0, (T)-1, hR1);
    n = clip_axis<T, Lds>(lds, P0, n, P1, 1, (T)1, hR2);
    n = clip_axis<T, Lds>(lds, P1, n, P0, 1, (T)-1, hR2);
    int nk = 0;
    for (int i = 0; i < n; i++)
        if (lds[P0 + 3 * i + 2] < margin) {
            lds[KP + 3 * nk] = lds[P0 + 3 * i]; lds[KP + 3 * nk + 1] = lds[P0 + 3 * i + 1]; lds[KP + 3 * nk + 2] = lds[P0 + 3 * i + 2];
            nk++;
        }
    if (nk == 0) return 0;
    int s0 = 0, s1 = 1, s2 = 2, s3 = 3, ns = nk;
    if (nk > 4) {
        // the deepest point, the one farthest from it, and the farthest one on either side of that line
        int i0 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 1; i < nk; i++) if (lds[KP + 3 * i + 2] < lds[KP + 3 * i0 + 2]) i0 = i;
        const T x0 = lds[KP + 3 * i0], y0 = lds[KP + 3 * i0 + 1];
        T best1 = (T)-1;
        for (int i = 0; i < nk; i++) {
            if (i == i0) continue;
            const T dx = lds[KP + 3 * i] - x0, dy = lds[KP + 3 * i + 1] - y0, d2 = dx * dx + dy * dy;
            if (d2 > best1) { best1 = d2; i1 = i; }
        }
        const T ex = lds[KP + 3 * i1] - x0, ey = lds[KP + 3 * i1 + 1] - y0;
        T smax = (T)0, smin = (T)0;
        for (int i = 0; i < nk; i++) {
            if (i == i0 || i == i1) continue;
            const T sd = ex * (lds[KP + 3 * i + 1] - y0) - ey * (lds[KP + 3 * i] - x0);
            if (sd > smax) { smax = sd; i2 = i; }
            if (sd < smin) { smin = sd; i3 = i; }
        }
        s0 = i0; s1 = i1; ns = 2;
        if (i2 >= 0) { s2 = i2; ns = 3; }
        if (i3 >= 0) { if (ns == 2) s2 = i3; else s3 = i3; ns++; }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int si = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
        if (q < ns) {
            const T vx = lds[KP + 3 * si], vy = lds[KP + 3 * si + 1], vz = lds[KP + 3 * si + 2];
            pts[q] = pR + Rr1 * vx + Rr2 * vy + dR * (vz + hRi);
            dist[q] = vz;
        }
    }
    nrm = refA ? dR * (T)-1 : dR;
    return ns;
}
template <typename T, typename Lds>
XARM_HD int cube_cube(V3<T> pA, const V3<T> (&A)[3], V3<T> pB, const V3<T> (&B)[3], T h, T margin, V3<T> (&pts)[4], V3<T> &nrm, T (&dist)[4], Lds lds) {
    const T hh[3] = {h, h, h};
    return box_box<T, Lds, LDS_CLIP>(pA, A, hh, pB, B, hh, margin, pts, nrm, dist, lds);
}

// ---------------------------------------------------------------------------------------------
// one internal substep of the two-arm / three-cube scene (dt = timeStep / numSubSteps)
template <typename T, typename Lds, typename Xchg>
XARM_HD void substep(Lane<T> &L, const T dt, Lds lds, const int arm, const Xchg xchg) {
    const T idt = (T)1 / dt;
    xk::ArmDyn<T> AD;
    xk::arm_dynamics<T, Lds, StackScene>(L.q, L.qd, dt, lds, arm, AD);
    T (&Minv)[45] = AD.Minv;
    T (&dq)[9] = AD.dq;
    const V3<T> hc0 = AD.hc0, hc1 = AD.hc1, hc2 = AD.hc2;

    // ---------------- cubes: frames, unconstrained motion (isotropic inertia: no gyroscopic term)
    const T h = (T)xm::ST_CUBE_HALF;
    const T imb = (T)(1.0 / xm::ST_CUBE_MASS), ii = (T)(1.0 / (xm::ST_CUBE_MASS * 2.0 / 3.0 * xm::ST_CUBE_HALF * xm::ST_CUBE_HALF));
    V3<T> cb[NOBJ], Rb[NOBJ][3], vb[NOBJ], wb[NOBJ];
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        const T x = L.bq[o][0], y = L.bq[o][1], z = L.bq[o][2], w = L.bq[o][3];
        Rb[o][0] = mk<T>((T)1 - (T)2 * (y * y + z * z), (T)2 * (x * y + z * w), (T)2 * (x * z - y * w));
        Rb[o][1] = mk<T>((T)2 * (x * y - z * w), (T)1 - (T)2 * (x * x + z * z), (T)2 * (y * z + x * w));
        Rb[o][2] = mk<T>((T)2 * (x * z + y * w), (T)2 * (y * z - x * w), (T)1 - (T)2 * (x * x + y * y));
        cb[o] = ldv(L.bp[o]);
        vb[o] = ldv(L.bv[o]); wb[o] = ldv(L.bw[o]);
        vb[o].z -= dt * (T)xm::GRAVITY;
        vb[o] = vb[o] * (T)xm::LIN_DAMP_FACTOR;
        wb[o] = wb[o] * (T)xm::ANG_DAMP_FACTOR;
    }

    // ---------------- (T) cube corners against the table top: first <= 4 active corners per cube -> LDS slots
    const T mu_t = (T)(xm::MU_OBJECT * xm::MU_TABLE);
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int k = 0; k < TP_W; k++) lds[LDS_TP + (o * 4 + s) * TP_W + k] = k == TP_W - 1 ? (T)-1 : (T)0;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const V3<T> r = Rb[o][0] * ((i & 1) ? h : -h) + Rb[o][1] * ((i & 2) ? h : -h) + Rb[o][2] * ((i & 4) ? h : -h);
            const V3<T> p = cb[o] + r;
            const bool on_table = xk::xabs(p.x) <= (T)xm::TABLE_HALF_X && xk::xabs(p.y) <= (T)xm::TABLE_HALF_Y;
            const T dist = p.z - (T)xm::TABLE_TOP_Z;
            const bool act = on_table && dist < (T)xm::SOLVER_MARGIN && cnt < 4;
            if (act) {
                const int base = LDS_TP + (o * 4 + cnt) * TP_W;
                const T l0 = (T)xm::WARMSTART * L.lam_t[o][i];
                const T rr = dot(r, r);
                lds[base + 0] = r.x; lds[base + 1] = r.y; lds[base + 2] = r.z;
                lds[base + 3] = l0; lds[base + 4] = (T)0; lds[base + 5] = (T)0;
                lds[base + 6] = dist < (T)0 ? -(T)xm::CONTACT_ERP * dist * idt : -dist * idt;
                lds[base + 7] = (T)1 / (imb + ii * (rr - r.z * r.z));   // n = +z
                lds[base + 8] = (T)1 / (imb + ii * (rr - r.y * r.y));   // t1 = -y
                lds[base + 9] = (T)1 / (imb + ii * (rr - r.x * r.x));   // t2 = +x
                lds[base + 10] = (T)i;
                // warm start
                vb[o].z += imb * l0;
                wb[o] = wb[o] + cross(r, mk<T>((T)0, (T)0, l0)) * ii;
                cnt++;
            }
        }
    }

    // ---------------- (BB) cube / cube manifolds -> LDS
    const T mu_bb = (T)(xm::MU_OBJECT * xm::MU_OBJECT);
    bool bb_any = false, pair_act[NPAIR] = {false, false, false};
#pragma unroll
    for (int pr = 0; pr < NPAIR; pr++) {
        const int a = pr == 2 ? 1 : 0, b = pr == 0 ? 1 : 2;
        const int base = LDS_BB + pr * BB_PAIR;
#pragma unroll
        for (int k = 0; k < BB_PAIR; k++) lds[base + k] = (T)0;
        const V3<T> d = cb[a] - cb[b];
        // bounding spheres: 2 * sqrt(3) * h + margin
        const T reach = (T)(2.0 * 1.7320508075688772 * xm::ST_CUBE_HALF + xm::SOLVER_MARGIN);
        const bool near = dot(d, d) < reach * reach;
        if (XARM_ANY(near)) {
            V3<T> pts[4], nrm = mk<T>(0, 0, 1);
            T dist[4];
            const int np = near ? cube_cube<T, Lds>(cb[a], Rb[a], cb[b], Rb[b], h, (T)xm::SOLVER_MARGIN, pts, nrm, dist, lds) : 0;
            if (np > 0) {
                bb_any = true; pair_act[pr] = true;
                const V3<T> t1 = xk::plane_space(nrm), t2 = cross(nrm, t1);
                lds[base + 0] = nrm.x; lds[base + 1] = nrm.y; lds[base + 2] = nrm.z;
                lds[base + 3] = t1.x; lds[base + 4] = t1.y; lds[base + 5] = t1.z;
                for (int q = 0; q < np; q++) {
                    const int pb = base + 6 + q * BB_W;
                    const V3<T> rA = pts[q] - cb[a], rB = pts[q] - cb[b];
                    const T ra2 = dot(rA, rA), rb2 = dot(rB, rB);
                    lds[pb + 0] = rA.x; lds[pb + 1] = rA.y; lds[pb + 2] = rA.z;
                    lds[pb + 3] = rB.x; lds[pb + 4] = rB.y; lds[pb + 5] = rB.z;
                    lds[pb + 9] = dist[q] < (T)0 ? -(T)xm::CONTACT_ERP * dist[q] * idt : -dist[q] * idt;
                    // point Delassus block K = (2/m + (|rA|^2 + |rB|^2)/I) 1 - (rA rA^T + rB rB^T)/I; K d for the three rows
                    const T kd = (T)2 * imb + ii * (ra2 + rb2);
                    const V3<T> dirs[3] = {nrm, t1, t2};
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const V3<T> Kd = dirs[k] * kd - (rA * dot(rA, dirs[k]) + rB * dot(rB, dirs[k])) * ii;
                        lds[pb + 10 + k] = (T)1 / dot(dirs[k], Kd);
                        lds[pb + 13 + 3 * k] = Kd.x; lds[pb + 14 + 3 * k] = Kd.y; lds[pb + 15 + 3 * k] = Kd.z;
                    }
                }
            }
        }
    }

    // ---------------- (M) motors, (L) limits, (G) gear: row constants (as PickAndPlace)
    T m_vt[9], m_invd[9], m_lam[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        m_vt[i] = (T)xm::MOTOR_KP * (L.qt[i] - L.q[i]) * idt + (T)(1.0 - xm::MOTOR_KD) * dq[i];
        m_invd[i] = (T)1 / Minv[tri(i, i)];
        m_lam[i] = (T)0;
    }
    const T m_hi_arm = (T)(xm::ARM_MOTOR_FORCE * StackScene::TIME_STEP), m_hi_fin = (T)(StackScene::FINGER_MOTOR_FORCE * StackScene::TIME_STEP);
    T la_vt[7], la_sg[7], la_lam[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const T g0 = L.q[i] - (T)xm::LOWER[i], g1 = (T)xm::UPPER[i] - L.q[i];
        const bool lo = g0 < (T)xm::LIMIT_WINDOW, hi = g1 < (T)xm::LIMIT_WINDOW;
        const T g = lo ? g0 : g1;
        la_sg[i] = lo ? (T)1 : (hi ? (T)-1 : (T)0);
        la_vt[i] = g < (T)0 ? -(T)xm::GLOBAL_ERP * g * idt : -g * idt;
        la_lam[i] = (T)0;
    }
    T lf_vt[2][2], lf_lam[2][2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const T g0 = L.q[7 + k] - (T)xm::LOWER[7 + k], g1 = (T)xm::UPPER[7 + k] - L.q[7 + k];
        lf_vt[k][0] = g0 < (T)0 ? -(T)xm::GLOBAL_ERP * g0 * idt : -g0 * idt;
        lf_vt[k][1] = g1 < (T)0 ? -(T)xm::GLOBAL_ERP * g1 * idt : -g1 * idt;
        lf_lam[k][0] = lf_lam[k][1] = (T)0;
    }
    const T g_vt = -(T)(xm::GEAR_ERP * xm::GLOBAL_ERP) * (L.q[7] - L.q[8]) * idt;
    const T g_hi = (T)(xm::GEAR_MAX_FORCE * StackScene::TIME_STEP);
    const T g_invd = (T)1 / (Minv[tri(7, 7)] - (T)2 * Minv[tri(8, 7)] + Minv[tri(8, 8)]);
    T g_lam = (T)0;

    // ---------------- (F) finger pad spheres, each against its nearest cube
    constexpr int NP = xk::NP;
    static_assert(xm::NPAD == 2, "the finger block of the sweep fuses exactly two pad points per finger");
    PadPoint<T> pp[NP];
    int pc[NP];
    T K21[2][9];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int e = 0; e < 9; e++) K21[k][e] = (T)0;
    bool pad_any = false;
    const T pad_denom = dt * (T)xm::FINGER_CONTACT_STIFFNESS + (T)(xm::FINGER_CONTACT_DAMPING + xm::OBJECT_CONTACT_DAMPING);
    const T pad_cfm = ((T)1 / pad_denom) * idt, pad_erp = dt * (T)xm::FINGER_CONTACT_STIFFNESS / pad_denom;
    {
        T wtot[8];
#pragma unroll
        for (int k = 0; k < 8; k++) wtot[k] = (T)0;
        V3<T> vb_pre[NOBJ], wb_pre[NOBJ];
#pragma unroll
        for (int o = 0; o < NOBJ; o++) { vb_pre[o] = vb[o]; wb_pre[o] = wb[o]; }
#pragma unroll
        for (int idx = 0; idx < NP; idx++) {
            const int fk = idx / xm::NPAD, j = idx % xm::NPAD;
            const T sg = fk == 0 ? (T)1 : (T)-1;
            PadPoint<T> &P = pp[idx];
            const V3<T> c = AD.fo[fk] + hc0 * (T)xm::PAD_C[j][0] + hc1 * (sg * (T)xm::PAD_C[j][1]) + hc2 * (T)xm::PAD_C[j][2];
            T dist = (T)1e30;
            V3<T> nw = mk<T>(0, 0, 1), pw = mk<T>(0, 0, 0);
            int co = 0;
#pragma unroll
            for (int o = 0; o < NOBJ; o++) {
                // sphere against cube o, in the cube's axes
                const V3<T> d = c - cb[o];
                const V3<T> cl = mk<T>(dot(Rb[o][0], d), dot(Rb[o][1], d), dot(Rb[o][2], d));
                const V3<T> ql = mk<T>(clampT(cl.x, -h, h), clampT(cl.y, -h, h), clampT(cl.z, -h, h));
                const V3<T> dl = cl - ql;
                const T d2 = dot(dl, dl);
                V3<T> nl, pl;
                T di;
                if (d2 > (T)1e-12) {
                    const T len = xk::xsqrt(d2);
                    nl = dl * ((T)1 / len);
                    di = len - (T)xm::PAD_RADIUS;
                    pl = ql;
                } else {
                    const T px = h - xk::xabs(cl.x), py = h - xk::xabs(cl.y), pz = h - xk::xabs(cl.z);
                    int k = 0;
                    T bestp = px;
                    if (py < bestp) { bestp = py; k = 1; }
                    if (pz < bestp) { bestp = pz; k = 2; }
                    const T clk = k == 0 ? cl.x : (k == 1 ? cl.y : cl.z);
                    const T s1 = clk < (T)0 ? (T)-1 : (T)1;
                    nl = mk<T>(k == 0 ? s1 : (T)0, k == 1 ? s1 : (T)0, k == 2 ? s1 : (T)0);
                    di = -bestp - (T)xm::PAD_RADIUS;
                    pl = mk<T>(k == 0 ? s1 * h : cl.x, k == 1 ? s1 * h : cl.y, k == 2 ? s1 * h : cl.z);
                }
                if (di < dist) {
                    dist = di; co = o;
                    nw = Rb[o][0] * nl.x + Rb[o][1] * nl.y + Rb[o][2] * nl.z;
                    pw = cb[o] + Rb[o][0] * pl.x + Rb[o][1] * pl.y + Rb[o][2] * pl.z;
                }
            }
            const bool act = dist < (T)xm::SOLVER_MARGIN;
            pad_any = pad_any || act;
            pc[idx] = co;
            P.n = nw; P.p = pw;
            P.t1 = xk::plane_space(P.n);
            P.vt = dist < (T)0 ? -pad_erp * dist * idt : -dist * idt;
            P.lam[0] = act ? (T)xm::WARMSTART * L.lam_p[idx] : (T)0;
            P.lam[1] = P.lam[2] = (T)0;
            P.invd[0] = P.invd[1] = P.invd[2] = (T)0;
            P.Kn = P.Kt1 = P.Kt2 = mk<T>(0, 0, 0);
            if (XARM_ANY(act)) {
                const V3<T> af = hc1 * sg;
                const V3<T> cc = sel3v(co, cb[0], cb[1], cb[2]);
                const V3<T> r = P.p - cc;
                T K[3][3];
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const V3<T> ej = mk<T>(e == 0 ? (T)1 : (T)0, e == 1 ? (T)1 : (T)0, e == 2 ? (T)1 : (T)0);
                    const V3<T> mo = cross(P.p, ej);
                    const T W[6] = {mo.x, mo.y, mo.z, ej.x, ej.y, ej.z};
                    const T wf = xk::comp(af, e);
                    T Y[6];
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        T s = lds[LDS_T + (7 + fk) * 6 + a] * wf;
#pragma unroll
                        for (int b = 0; b < 6; b++) s += lds[LDS_AHH + symi(a, b)] * W[b];
                        Y[a] = s;
                    }
                    T yf = Minv[tri(7 + fk, 7 + fk)] * wf;
#pragma unroll
                    for (int b = 0; b < 6; b++) yf += lds[LDS_T + (7 + fk) * 6 + b] * W[b];
                    const V3<T> va = mk<T>(Y[3], Y[4], Y[5]) + cross(mk<T>(Y[0], Y[1], Y[2]), P.p) + af * yf;
                    const V3<T> vbj = ej * imb - cross(r, cross(r, ej)) * ii;
                    K[0][e] = va.x + vbj.x; K[1][e] = va.y + vbj.y; K[2][e] = va.z + vbj.z;
                }
                const V3<T> t2 = cross(P.n, P.t1);
                P.Kn = mk<T>(K[0][0] * P.n.x + K[0][1] * P.n.y + K[0][2] * P.n.z, K[1][0] * P.n.x + K[1][1] * P.n.y + K[1][2] * P.n.z,
                             K[2][0] * P.n.x + K[2][1] * P.n.y + K[2][2] * P.n.z);
                P.Kt1 = mk<T>(K[0][0] * P.t1.x + K[0][1] * P.t1.y + K[0][2] * P.t1.z, K[1][0] * P.t1.x + K[1][1] * P.t1.y + K[1][2] * P.t1.z,
                              K[2][0] * P.t1.x + K[2][1] * P.t1.y + K[2][2] * P.t1.z);
                P.Kt2 = mk<T>(K[0][0] * t2.x + K[0][1] * t2.y + K[0][2] * t2.z, K[1][0] * t2.x + K[1][1] * t2.y + K[1][2] * t2.z,
                              K[2][0] * t2.x + K[2][1] * t2.y + K[2][2] * t2.z);
                P.invd[0] = act ? (T)1 / (dot(P.n, P.Kn) + pad_cfm) : (T)0;
                P.invd[1] = act ? (T)1 / dot(P.t1, P.Kt1) : (T)0;
                P.invd[2] = act ? (T)1 / dot(t2, P.Kt2) : (T)0;
                // warm start: +lam0 n on the finger, -lam0 n on the cube
                const V3<T> fi = P.n * P.lam[0];
                const V3<T> mo = cross(P.p, fi);
                wtot[0] += mo.x; wtot[1] += mo.y; wtot[2] += mo.z;
                wtot[3] += fi.x; wtot[4] += fi.y; wtot[5] += fi.z;
                wtot[6 + fk] += dot(af, fi);
                const V3<T> dv = fi * imb, dw = cross(r, fi) * ii;
#pragma unroll
                for (int o = 0; o < NOBJ; o++) {
                    vb[o] = co == o ? vb[o] - dv : vb[o];
                    wb[o] = co == o ? wb[o] - dw : wb[o];
                }
            }
        }
        if (XARM_ANY(pad_any)) {
#pragma unroll
            for (int r = 0; r < 9; r++) {
                T s = Minv[symi(r, 7)] * wtot[6] + Minv[symi(r, 8)] * wtot[7];
#pragma unroll
                for (int k = 0; k < 6; k++) s += lds[LDS_T + r * 6 + k] * wtot[k];
                dq[r] += s;
            }
            // arm-side coupling of the two pad points of each finger: velocity of the finger at its second point per unit
            // impulse at its first one (3x3, row-major) - lets both points be swept before ONE operational-space update
            // (as xk::substep; the cube side is applied to the cube velocities point by point)
#pragma unroll
            for (int fk = 0; fk < 2; fk++) {
                const PadPoint<T> &P1 = pp[2 * fk], &P2 = pp[2 * fk + 1];
                if (!XARM_ANY(P1.invd[0] != (T)0 && P2.invd[0] != (T)0)) continue;
                const V3<T> af = hc1 * (fk == 0 ? (T)1 : (T)-1);
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const V3<T> ej = mk<T>(e == 0 ? (T)1 : (T)0, e == 1 ? (T)1 : (T)0, e == 2 ? (T)1 : (T)0);
                    const V3<T> mo = cross(P1.p, ej);
                    const T W[6] = {mo.x, mo.y, mo.z, ej.x, ej.y, ej.z};
                    const T wf = xk::comp(af, e);
                    T Y[6];
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        T s = lds[LDS_T + (7 + fk) * 6 + a] * wf;
#pragma unroll
                        for (int b = 0; b < 6; b++) s += lds[LDS_AHH + symi(a, b)] * W[b];
                        Y[a] = s;
                    }
                    T yf = Minv[tri(7 + fk, 7 + fk)] * wf;
#pragma unroll
                    for (int b = 0; b < 6; b++) yf += lds[LDS_T + (7 + fk) * 6 + b] * W[b];
                    const V3<T> va = mk<T>(Y[3], Y[4], Y[5]) + cross(mk<T>(Y[0], Y[1], Y[2]), P2.p) + af * yf;
                    K21[fk][0 * 3 + e] = va.x; K21[fk][1 * 3 + e] = va.y; K21[fk][2 * 3 + e] = va.z;
                }
            }
        }
        // the cubes also receive the warm-start impulses of the other arm's pads; afterwards both lanes must hold
        // bit-identical cube velocities: take arm 0's sums
#pragma unroll
        for (int o = 0; o < NOBJ; o++) {
            const V3<T> dv = vb[o] - vb_pre[o], dw = wb[o] - wb_pre[o];
            vb[o] = vb[o] + mk<T>(xchg.partner(dv.x), xchg.partner(dv.y), xchg.partner(dv.z));
            wb[o] = wb[o] + mk<T>(xchg.partner(dw.x), xchg.partner(dw.y), xchg.partner(dw.z));
            vb[o] = mk<T>(xchg.from0(vb[o].x), xchg.from0(vb[o].y), xchg.from0(vb[o].z));
            wb[o] = mk<T>(xchg.from0(wb[o].x), xchg.from0(wb[o].y), xchg.from0(wb[o].z));
        }
    }
    // cubes touched by this arm's active pads / by the partner arm's; `seq` is wave-uniform
    int mymask = 0;
#pragma unroll
    for (int idx = 0; idx < NP; idx++) mymask |= pp[idx].invd[0] != (T)0 ? (1 << pc[idx]) : 0;
    const int othermask = (int)xchg.partner((T)mymask);
    const bool seq = XARM_ANY_X((mymask & othermask) != 0);
    L.cls = (pair_act[0] ? 1 : 0) | (pair_act[1] ? 2 : 0) | (pair_act[2] ? 4 : 0) |
            ((arm == 0 ? mymask : othermask) != 0 ? 8 : 0) | ((arm == 1 ? mymask : othermask) != 0 ? 16 : 0);
    XARM_LDS_FENCE();

    bool la_lane = false;
#pragma unroll
    for (int i = 0; i < 7; i++) la_lane = la_lane || la_sg[i] != (T)0;
    const bool la_wave = XARM_ANY(la_lane);
    // ... and which of the seven: at 65 536 envs a handful always have ONE joint near a limit, and the launch lasts as
    // long as its slowest wavefront - that wavefront now sweeps the one row, not all seven
    bool la_row[7];
#pragma unroll
    for (int i = 0; i < 7; i++) la_row[i] = la_wave && XARM_ANY(la_sg[i] != (T)0);
    // packed working set of the sweep (as PickAndPlace): joint velocities as 4 pairs + dq[8], full Minv columns as pairs
    xk::Pk<T> dqp[4], MC[9][4];
    T dq8 = dq[8], ML[9];
#pragma unroll
    for (int k = 0; k < 4; k++) dqp[k] = xk::mkpk<T>(dq[2 * k], dq[2 * k + 1]);
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) MC[i][k] = xk::mkpk<T>(Minv[symi(2 * k, i)], Minv[symi(2 * k + 1, i)]);
        ML[i] = Minv[symi(8, i)];
    }
#define XARM_DQ(i) ((i) == 8 ? dq8 : (((i) & 1) ? xk::pkhi(dqp[(i) >> 1]) : xk::pklo(dqp[(i) >> 1])))
#define XARM_DQ_AXPY(col, dl_) do { _Pragma("unroll") for (int k_ = 0; k_ < 4; k_++) dqp[k_] = xk::pkfma(MC[col][k_], (dl_), dqp[k_]); dq8 += ML[col] * (dl_); } while (0)
    // ---------------- projected Gauss-Seidel: T, BB, (M L G) of this lane's arm, F arm 0, F arm 1
    const T mu_p = (T)(xm::MU_OBJECT * xm::MU_FINGER);
#pragma unroll 1
    for (int it = 0; it < XK_SWEEP_ITERS; it++) {
        XARM_LDS_FENCE();
        // (T) n = +z, t1 = -y, t2 = +x
#pragma unroll
        for (int o = 0; o < NOBJ; o++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                // no wave-level skip (1/diag = 0 makes an empty slot a no-op): the slots of different cubes and the arm
                // rows are independent chains, and in one basic block they fill each other's dependency stalls
                const int base = LDS_TP + (o * 4 + s) * TP_W;
                const T e0 = lds[base + 7];
                const V3<T> r = mk<T>(lds[base + 0], lds[base + 1], lds[base + 2]);
                const T e1 = lds[base + 8], e2 = lds[base + 9];
                T l0 = lds[base + 3], l1 = lds[base + 4], l2 = lds[base + 5];
                V3<T> v = vb[o], w = wb[o];
                T dl = (lds[base + 6] - (v.z + w.x * r.y - w.y * r.x)) * e0;
                T nl = l0 + dl;
                nl = xk::smax0(nl);
                dl = nl - l0; l0 = nl;
                v.z += imb * dl;
                w.x += ii * r.y * dl; w.y -= ii * r.x * dl;
                const T lim = mu_t * l0;
                dl = (v.y + w.z * r.x - w.x * r.z) * e1;        // jv = -u.y, target 0
                nl = xk::sclamp(l1 + dl, -lim, lim);
                dl = nl - l1; l1 = nl;
                v.y -= imb * dl;
                w.x += ii * r.z * dl; w.z -= ii * r.x * dl;
                dl = -(v.x + w.y * r.z - w.z * r.y) * e2;
                nl = xk::sclamp(l2 + dl, -lim, lim);
                dl = nl - l2; l2 = nl;
                v.x += imb * dl;
                w.y += ii * r.z * dl; w.z -= ii * r.y * dl;
                vb[o] = v; wb[o] = w;
                lds[base + 3] = l0; lds[base + 4] = l1; lds[base + 5] = l2;
            }
        // the arm rows (M L G) only touch this lane's joints and the T / BB rows only the cubes: issued next to the table
        // slots (one basic block) they fill each other's dependency stalls; the result is the oracle's T, BB, MLG order
        // (M) velocity-level PD motors
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const T hi = i < 7 ? m_hi_arm : m_hi_fin;
            T dl = (m_vt[i] - XARM_DQ(i)) * m_invd[i];
            const T nl = xk::sclamp(m_lam[i] + dl, -hi, hi);
            dl = nl - m_lam[i];
            m_lam[i] = nl;
            XARM_DQ_AXPY(i, dl);
        }
        // (L) joint limits
#pragma unroll
        for (int i = 0; i < 7; i++) {
            if (!la_row[i]) continue;   // wave-uniform, decided once per substep
            const T sg = la_sg[i];
            T dl = (la_vt[i] - sg * XARM_DQ(i)) * (sg != (T)0 ? m_invd[i] : (T)0);
            T nl = la_lam[i] + dl;
            nl = xk::smax0(nl);
            dl = (nl - la_lam[i]) * sg;
            la_lam[i] = nl;
            XARM_DQ_AXPY(i, dl);
        }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int side = 0; side < 2; side++) {
                const T sg = side == 0 ? (T)1 : (T)-1;
                T dl = (lf_vt[k][side] - sg * XARM_DQ(7 + k)) * m_invd[7 + k];
                T nl = lf_lam[k][side] + dl;
                nl = xk::smax0(nl);
                dl = (nl - lf_lam[k][side]) * sg;
                lf_lam[k][side] = nl;
                XARM_DQ_AXPY(7 + k, dl);
            }
        // (G) gear row
        {
            T dl = (g_vt - (XARM_DQ(7) - dq8)) * g_invd;
            const T nl = xk::sclamp(g_lam + dl, -g_hi, g_hi);
            dl = nl - g_lam;
            g_lam = nl;
            XARM_DQ_AXPY(7, dl);
            XARM_DQ_AXPY(8, -dl);
        }
        // (BB) cube / cube points
        if (XARM_ANY(bb_any)) {
#pragma unroll
            for (int pr = 0; pr < NPAIR; pr++) {
                const int a = pr == 2 ? 1 : 0, b = pr == 0 ? 1 : 2;
                const int base = LDS_BB + pr * BB_PAIR;
                if (!XARM_ANY(pair_act[pr])) continue;
                const V3<T> n = mk<T>(lds[base + 0], lds[base + 1], lds[base + 2]), t1 = mk<T>(lds[base + 3], lds[base + 4], lds[base + 5]);
                const V3<T> t2 = cross(n, t1);
                // all four slots of an active pair, unrolled and unconditional (an empty slot is a no-op): the LDS reads
                // of the next point are issued while the current one is solved
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int pb = base + 6 + q * BB_W;
                    const T e0 = lds[pb + 10];
                    const V3<T> rA = mk<T>(lds[pb + 0], lds[pb + 1], lds[pb + 2]), rB = mk<T>(lds[pb + 3], lds[pb + 4], lds[pb + 5]);
                    T lam[3] = {lds[pb + 6], lds[pb + 7], lds[pb + 8]};
                    const T ed[3] = {e0, lds[pb + 11], lds[pb + 12]};
                    const T vt = lds[pb + 9];
                    // relative velocity at the point once, then kept current through K d per row; one impulse at the end
                    V3<T> u = vb[a] + cross(wb[a], rA) - vb[b] - cross(wb[b], rB);
                    V3<T> f = mk<T>(0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const V3<T> d = k == 0 ? n : (k == 1 ? t1 : t2);
                        const V3<T> Kd = mk<T>(lds[pb + 13 + 3 * k], lds[pb + 14 + 3 * k], lds[pb + 15 + 3 * k]);
                        T dl = ((k == 0 ? vt : (T)0) - dot(d, u)) * ed[k];
                        const T lim = mu_bb * lam[0];
                        const T nl = k == 0 ? xk::smax0(lam[0] + dl) : xk::sclamp(lam[k] + dl, -lim, lim);
                        dl = nl - lam[k];
                        lam[k] = nl;
                        u = u + Kd * dl;
                        f = f + d * dl;
                    }
                    vb[a] = vb[a] + f * imb; wb[a] = wb[a] + cross(rA, f) * ii;
                    vb[b] = vb[b] - f * imb; wb[b] = wb[b] - cross(rB, f) * ii;
                    lds[pb + 6] = lam[0]; lds[pb + 7] = lam[1]; lds[pb + 8] = lam[2];
                }
            }
        }
        // (F) pad points.  Sequential form: arm 0's pads, hand the cube velocities over, arm 1's pads.  When no cube
        // of any environment in the wavefront is touched by both arms the two sweeps act on disjoint variables and
        // commute, so both lanes sweep at once (phase 0) and each cube is then taken from the lane that touched it.
        // One instruction stream serves both forms (a second copy of the sweep pushes the cube velocities to scratch).
#pragma unroll
        for (int ph = 0; ph < 2; ph++) {
            // sequential: phase 0 = arm 0, phase 1 = arm 1; concurrent: every lane sweeps in phase 0, phase 1 is empty
            const bool mine = seq ? arm == ph : ph == 0;
            if (XARM_ANY(pad_any && mine)) {
                T y[6], yf[2], wtot[8];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    T s = (T)0;
#pragma unroll
                    for (int i = 0; i < 7; i++) s += lds[LDS_S + i * 6 + k] * XARM_DQ(i);
                    y[k] = s;
                }
                yf[0] = XARM_DQ(7); yf[1] = dq8;
#pragma unroll
                for (int k = 0; k < 8; k++) wtot[k] = (T)0;
#pragma unroll
                for (int fk = 0; fk < 2; fk++) {
                    if (!XARM_ANY((pp[2 * fk].invd[0] != (T)0 || pp[2 * fk + 1].invd[0] != (T)0) && mine)) continue;
                    const V3<T> af = hc1 * (fk == 0 ? (T)1 : (T)-1);
                    const V3<T> yw = mk<T>(y[0], y[1], y[2]);
                    const V3<T> base = mk<T>(y[3], y[4], y[5]) + af * yf[fk];
                    V3<T> fsum = mk<T>(0, 0, 0), msum = mk<T>(0, 0, 0), f1 = mk<T>(0, 0, 0);   // sum f, sum p x f, the first point's impulse
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        PadPoint<T> &P = pp[2 * fk + j];
                        const int co = pc[2 * fk + j];
                        const T e0 = mine ? P.invd[0] : (T)0, e1 = mine ? P.invd[1] : (T)0, e2 = mine ? P.invd[2] : (T)0;
                        const V3<T> r = P.p - sel3v(co, cb[0], cb[1], cb[2]);
                        const V3<T> vc = sel3v(co, vb[0], vb[1], vb[2]), wc = sel3v(co, wb[0], wb[1], wb[2]);
                        const V3<T> t2 = cross(P.n, P.t1);
                        V3<T> u = base + cross(yw, P.p) - vc - cross(wc, r);
                        if (j == 1) // effect on the finger of the impulse just applied at its first point (the cube side went into vb / wb)
                            u = u + mk<T>(K21[fk][0] * f1.x + K21[fk][1] * f1.y + K21[fk][2] * f1.z,
                                          K21[fk][3] * f1.x + K21[fk][4] * f1.y + K21[fk][5] * f1.z,
                                          K21[fk][6] * f1.x + K21[fk][7] * f1.y + K21[fk][8] * f1.z);
                        T dl = (P.vt - pad_cfm * P.lam[0] - dot(P.n, u)) * e0;
                        T nl = P.lam[0] + dl;
                        nl = xk::smax0(nl);
                        dl = nl - P.lam[0];
                        P.lam[0] = nl;
                        V3<T> fi = P.n * dl;
                        u = u + P.Kn * dl;
                        const T lim = mu_p * P.lam[0];
                        dl = -dot(P.t1, u) * e1;
                        nl = xk::sclamp(P.lam[1] + dl, -lim, lim);
                        dl = nl - P.lam[1];
                        P.lam[1] = nl;
                        fi = fi + P.t1 * dl;
                        u = u + P.Kt1 * dl;
                        dl = -dot(t2, u) * e2;
                        nl = xk::sclamp(P.lam[2] + dl, -lim, lim);
                        dl = nl - P.lam[2];
                        P.lam[2] = nl;
                        fi = fi + t2 * dl;
                        if (j == 0) f1 = fi;
                        fsum = fsum + fi;
                        msum = msum + cross(P.p, fi);
                        // -fi on the cube, at once: the finger's second point may press on the same cube
                        const V3<T> dv = fi * imb, dw = cross(r, fi) * ii;
#pragma unroll
                        for (int o = 0; o < NOBJ; o++) {
                            vb[o] = co == o ? vb[o] - dv : vb[o];
                            wb[o] = co == o ? wb[o] - dw : wb[o];
                        }
                    }
                    // ONE operational-space update for the finger: +fsum on finger fk, moment msum about the world origin
                    const T W[6] = {msum.x, msum.y, msum.z, fsum.x, fsum.y, fsum.z};
                    const T wf = dot(af, fsum);
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        T s = lds[LDS_T + (7 + fk) * 6 + a] * wf;
#pragma unroll
                        for (int b = 0; b < 6; b++) s += lds[LDS_AHH + symi(a, b)] * W[b];
                        y[a] += s;
                    }
#pragma unroll
                    for (int k2 = 0; k2 < 2; k2++) {
                        T s = Minv[symi(7 + k2, 7 + fk)] * wf;
#pragma unroll
                        for (int b = 0; b < 6; b++) s += lds[LDS_T + (7 + k2) * 6 + b] * W[b];
                        yf[k2] += s;
                    }
#pragma unroll
                    for (int b = 0; b < 6; b++) wtot[b] += W[b];
                    wtot[6 + fk] += wf;
                }
                XARM_DQ_AXPY(7, wtot[6]);
                XARM_DQ_AXPY(8, wtot[7]);
#pragma unroll
                for (int k = 0; k < 6; k++) {
#pragma unroll
                    for (int r2 = 0; r2 < 4; r2++)
                        dqp[r2] = xk::pkfma(xk::mkpk<T>(lds[LDS_T + (2 * r2) * 6 + k], lds[LDS_T + (2 * r2 + 1) * 6 + k]), wtot[k], dqp[r2]);
                    dq8 += lds[LDS_T + 8 * 6 + k] * wtot[k];
                }
            }
            // hand the cube velocities over.  Three wave-uniform cases, the same values as one select cascade over all of them (which cost
            // 36 instructions per cube and sweep in every wavefront, pads or not): sequential - take arm 0's after phase 0, arm 1's after
            // phase 1; concurrent with a pad somewhere in the wavefront - take the partner's where only it touched; no pad at all - nothing
            if (ph == 0) {
                if (seq) {
#pragma unroll
                    for (int o = 0; o < NOBJ; o++) {
                        vb[o] = mk<T>(xchg.from0(vb[o].x), xchg.from0(vb[o].y), xchg.from0(vb[o].z));
                        wb[o] = mk<T>(xchg.from0(wb[o].x), xchg.from0(wb[o].y), xchg.from0(wb[o].z));
                    }
                } else if (XARM_ANY(mymask != 0 || othermask != 0)) {
#pragma unroll
                    for (int o = 0; o < NOBJ; o++) {
                        const bool take = ((othermask >> o) & 1) != 0;   // the partner lane touched cube o, this one did not
                        const V3<T> pv = mk<T>(xchg.partner(vb[o].x), xchg.partner(vb[o].y), xchg.partner(vb[o].z));
                        const V3<T> pw = mk<T>(xchg.partner(wb[o].x), xchg.partner(wb[o].y), xchg.partner(wb[o].z));
                        vb[o] = selv(take, pv, vb[o]);
                        wb[o] = selv(take, pw, wb[o]);
                    }
                }
            } else if (seq) {
#pragma unroll
                for (int o = 0; o < NOBJ; o++) {
                    vb[o] = mk<T>(xchg.from1(vb[o].x), xchg.from1(vb[o].y), xchg.from1(vb[o].z));
                    wb[o] = mk<T>(xchg.from1(wb[o].x), xchg.from1(wb[o].y), xchg.from1(wb[o].z));
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { dq[2 * k] = xk::pklo(dqp[k]); dq[2 * k + 1] = xk::pkhi(dqp[k]); }
    dq[8] = dq8;
#undef XARM_DQ
#undef XARM_DQ_AXPY
    XARM_LDS_FENCE();

    // ---------------- store warm-start impulses, integrate (semi-implicit Euler)
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
#pragma unroll
        for (int i = 0; i < 8; i++) L.lam_t[o][i] = (T)0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int base = LDS_TP + (o * 4 + s) * TP_W;
            const int id = (int)lds[base + 10];
            const T l0 = lds[base + 3];
#pragma unroll
            for (int i = 0; i < 8; i++) L.lam_t[o][i] = id == i ? l0 : L.lam_t[o][i];
        }
    }
#pragma unroll
    for (int i = 0; i < NP; i++) L.lam_p[i] = pp[i].invd[0] != (T)0 ? pp[i].lam[0] : (T)0;
#pragma unroll
    for (int i = 0; i < 9; i++) { L.qd[i] = dq[i]; L.q[i] += dt * dq[i]; }
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        L.bp[o][0] += dt * vb[o].x; L.bp[o][1] += dt * vb[o].y; L.bp[o][2] += dt * vb[o].z;
        T ang = xk::xsqrt(dot(wb[o], wb[o]));
        if (ang * dt > (T)0.7853981633974483) ang = (T)0.7853981633974483 * idt;
        T sw, cw;
        xk::xsincos((T)0.5 * ang * dt, sw, cw);
        const T k = ang < (T)0.001 ? (T)0.5 * dt - dt * dt * dt * (T)0.020833333333 * ang * ang : sw / ang;
        const V3<T> ax = wb[o] * k;
        const T x = L.bq[o][0], y = L.bq[o][1], z = L.bq[o][2], w0 = L.bq[o][3];
        const T nx = cw * x + ax.x * w0 + ax.y * z - ax.z * y;
        const T ny = cw * y + ax.y * w0 + ax.z * x - ax.x * z;
        const T nz = cw * z + ax.z * w0 + ax.x * y - ax.y * x;
        const T nw = cw * w0 - ax.x * x - ax.y * y - ax.z * z;
        const T inv = (T)1 / xk::xsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        L.bq[o][0] = nx * inv; L.bq[o][1] = ny * inv; L.bq[o][2] = nz * inv; L.bq[o][3] = nw * inv;
        L.bv[o][0] = vb[o].x; L.bv[o][1] = vb[o].y; L.bv[o][2] = vb[o].z;
        L.bw[o][0] = wb[o].x; L.bw[o][1] = wb[o].y; L.bw[o][2] = wb[o].z;
    }
}

// p.stepSimulation() with numSubSteps = 15
template <typename T, typename Lds, typename Xchg> XARM_HD void tick(Lane<T> &L, Lds lds, int arm, Xchg x) {
    const T dt = (T)(xm::ST_TIME_STEP / xm::ST_N_SUBSTEPS);
#pragma unroll 1
    for (int k = 0; k < xm::ST_N_SUBSTEPS; k++) substep<T, Lds, Xchg>(L, dt, lds, arm, x);
}

// the 8 per-arm observation entries (:170-181): hand COM position and velocity, finger q, qd
template <typename T> XARM_HD void arm_obs(const Lane<T> &L, int arm, T (&o)[8]) {
    Frame<T> f = StackScene::base_frame<T>(arm);
    V3<T> w = mk<T>(0, 0, 0), v = mk<T>(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        xk::fk_advance(f, i, L.q[i]);
        w = w + f.c2 * L.qd[i];
        v = v + cross(f.o, f.c2) * L.qd[i];
    }
    const V3<T> hp = f.o + f.c0 * (T)xm::HAND_COM[0] + f.c1 * (T)xm::HAND_COM[1] + f.c2 * (T)xm::HAND_COM[2];
    const V3<T> hv = v + cross(w, hp);
    o[0] = hp.x; o[1] = hp.y; o[2] = hp.z;
    o[3] = hv.x; o[4] = hv.y; o[5] = hv.z;
    o[6] = L.q[7]; o[7] = L.qd[7];
}

// draws 0-5: cube xy (cube i: 2i, 2i+1), 6-7: tower xy
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
template <typename T> XARM_HD void sample_objects(const T (&u)[8], Lane<T> &L) {
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        L.bp[o][0] = (T)xm::ST_OBJ_LOW[0] + u[2 * o] * (T)(xm::ST_OBJ_HIGH[0] - xm::ST_OBJ_LOW[0]);
        L.bp[o][1] = (T)xm::ST_OBJ_LOW[1] + u[2 * o + 1] * (T)(xm::ST_OBJ_HIGH[1] - xm::ST_OBJ_LOW[1]);
        L.bp[o][2] = (T)xm::ST_HEIGHT_OFFSET;
        L.bq[o][0] = L.bq[o][1] = L.bq[o][2] = (T)0; L.bq[o][3] = (T)1;
#pragma unroll
        for (int k = 0; k < 3; k++) { L.bv[o][k] = (T)0; L.bw[o][k] = (T)0; }
#pragma unroll
        for (int k = 0; k < 8; k++) L.lam_t[o][k] = (T)0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) L.lam_p[k] = (T)0;
}
template <typename T> XARM_HD void sample_goal(const T (&u)[8], Lane<T> &L) {
    const T x = (T)xm::ST_GOAL_LOW[0] + u[6] * (T)(xm::ST_GOAL_HIGH[0] - xm::ST_GOAL_LOW[0]);
    const T y = (T)xm::ST_GOAL_LOW[1] + u[7] * (T)(xm::ST_GOAL_HIGH[1] - xm::ST_GOAL_LOW[1]);
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {   // :212-219
        L.goal[o][0] = x; L.goal[o][1] = y;
        L.goal[o][2] = (T)(xm::ST_HEIGHT_OFFSET * (2 * o + 1));
    }
}
template <typename T> XARM_HD void teleport_arm(Lane<T> &L) {
#pragma unroll
    for (int i = 0; i < 9; i++) { L.q[i] = (T)xm::ST_JOINT_INIT_POS[i]; L.qd[i] = (T)0; }
}
template <typename T> XARM_HD void lane_init(const EnvCfg &cfg, int64_t env, Lane<T> &L) {
    teleport_arm(L);
#pragma unroll
    for (int i = 0; i < 9; i++) L.qt[i] = L.q[i];   // no motor command yet: hold the init pose
    L.steps = L.episode = (T)0;
    L.cls = 0;
    T u[8];
    draws(cfg, env, 0, u);
    sample_objects(u, L);
    sample_goal(u, L);
}
// _reset_sim + _sample_goal (:201-219)
template <typename T, typename Lds, typename Xchg>
XARM_HD void lane_reset(const EnvCfg &cfg, int64_t env, Lane<T> &L, int arm, Lds lds, Xchg x) {
    const int64_t episode = (int64_t)L.episode + 1;
    T u[8];
    teleport_arm(L);
    draws(cfg, env, episode, u);
    sample_objects(u, L);
    tick<T, Lds, Xchg>(L, lds, arm, x);   // with the motor targets of the last step still set (:210)
    sample_goal(u, L);
    L.steps = (T)0;
    L.episode = (T)episode;
}
template <typename T> XARM_HD T goal_distance(const Lane<T> &L) {
    T d2 = (T)0;
#pragma unroll
    for (int o = 0; o < NOBJ; o++)
#pragma unroll
        for (int k = 0; k < 3; k++) { const T d = L.bp[o][k] - L.goal[o][k]; d2 += d * d; }
    return xk::xsqrt(d2);
}
// act = this arm's 4 action entries (:142-162)
template <typename T, typename Lds, typename Xchg>
XARM_HD void lane_step(const EnvCfg &cfg, Lane<T> &L, int arm, const T (&act)[4], T &reward, bool &done, bool &success, Lds lds, Xchg x) {
    L.steps += (T)1;
    T a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    Frame<T> f = StackScene::base_frame<T>(arm);
#pragma unroll
    for (int i = 0; i < 7; i++) xk::fk_advance(f, i, L.q[i]);
    const V3<T> cur = f.o;
    const T sc = (T)(xm::ST_MAX_VEL * xm::ST_ACTION_DT);
    const V3<T> lo = arm == 0 ? mk<T>((T)xm::ST_POS_LOW[0][0], (T)xm::ST_POS_LOW[0][1], (T)xm::ST_POS_LOW[0][2])
                              : mk<T>((T)xm::ST_POS_LOW[1][0], (T)xm::ST_POS_LOW[1][1], (T)xm::ST_POS_LOW[1][2]);
    const V3<T> hi = arm == 0 ? mk<T>((T)xm::ST_POS_HIGH[0][0], (T)xm::ST_POS_HIGH[0][1], (T)xm::ST_POS_HIGH[0][2])
                              : mk<T>((T)xm::ST_POS_HIGH[1][0], (T)xm::ST_POS_HIGH[1][1], (T)xm::ST_POS_HIGH[1][2]);
    const V3<T> target = mk<T>(clampT(cur.x + a[0] * sc, lo.x, hi.x), clampT(cur.y + a[1] * sc, lo.y, hi.y), clampT(cur.z + a[2] * sc, lo.z, hi.z));
    const T g = clampT(L.q[7] + a[3] * (T)(xm::ST_ACTION_DT * xm::ST_MAX_GRIPPER_VEL), (T)xm::ST_GRIPPER_LOW, (T)xm::ST_GRIPPER_HIGH);
    T qa[7], qo[7];
#pragma unroll
    for (int i = 0; i < 7; i++) qa[i] = L.q[i];
    xk::ik_arm<T, xm::ST_N_SUBSTEPS>(qa, target, qo, StackScene::base_frame<T>(arm));   // maxNumIterations = n_substeps (:154-155)
#pragma unroll
    for (int i = 0; i < 7; i++) L.qt[i] = qo[i];
    L.qt[7] = L.qt[8] = g;
    tick<T, Lds, Xchg>(L, lds, arm, x);
    const T dist = goal_distance(L);
    success = dist < (T)xm::ST_DISTANCE_THRESHOLD;                                   // :221-223
    reward = cfg.reward_type == 0 ? (dist > (T)xm::ST_DISTANCE_THRESHOLD ? (T)-1 : (T)0) : -dist;   // :124-129
    done = (int)L.steps == xm::ST_MAX_EPISODE_STEPS;                                // step() itself never ends (:111)
}

} // namespace xs
