// xarm_handover2_core.h - XarmHandover-v0 with config['num_obj'] = 2: two xarm7_pd arms (one lane each) and two sticks.
//
// This is the configuration of the reference's only test (/root/reference/test.py:9-15: num_obj 2, goal_shape 'any',
// same_side_rate 0.5, use_stand False) and of the env file's own __main__ (xarm_handover.py:448-455); use_stand True: Scene below.
// Reference: /root/reference/gym_xarm/envs/xarm_handover.py
//   lane_step  = XarmHandover.step :128-139 (+ _set_action :244-297: grasp flags / friction toggle from contacts with
//                legos[0] only :263-280, the clamp loop over every stick :282-297; _get_obs :299-336, obs 13 N + 16 = 42)
//   lane_reset = reset :141-145 (+ _reset_sim :338-368 with the y-spacing rejection of the second stick :357-360,
//                _sample_goal :370-393 with the goal rejection :375-379)
//   reward     = compute_reward sparse -sum_i [d_i > 0.05] :177-183, _is_success AND_i :395-402
// The contact model is the one the CPU oracle states (oracle/xarm_oracle_handover2.inc.c, model JSON
// handover._num_obj_2_cite): per stick the N = 1 support rows, stick/stick by the StackTower box/box manifold with the
// sticks' half extents, every pad sphere against its nearest stick.  Row order T stick 0, T stick 1, BB, (M L G) of the
// lane's arm, F arm 0, F arm 1 - the oracle's sequential sweep.
//
// Mapping and data placement as StackTower (xarm_stack_core.h): two lanes per environment, lane = arm; the arm part is
// xk::arm_dynamics with its LDS columns; the object rows (8 support slots, 4 stick/stick points) live in further
// lane-private LDS columns (411 floats = 1 644 B per lane, 105 KB per wavefront: one wavefront per CU); both lanes
// compute the object-only rows redundantly and bit-identically, the stick velocities are handed from lane to lane
// between the two finger phases.  Unlike the cubes the sticks have an anisotropic inertia: world inverse inertia
// tensors per stick and Bullet's implicit gyroscopic term, as in the one-stick scene (xk::substep).
#pragma once
#include "xarm_core.h"
#include "xarm_handover_core.h"
#include "xarm_stack_core.h"

namespace xh2 {
using xk::V3; using xk::mk; using xk::dot; using xk::cross; using xk::clampT; using xk::Frame; using xk::PadPoint;
using xk::tri; using xk::symi; using xk::LDS_S; using xk::LDS_T; using xk::LDS_AHH; using xk::symmul; using xk::selv;
using xh::HandoverScene; using xh::EnvCfg;

constexpr int NOBJ = 2;
constexpr int STATE_DIM = 100, OBS_DIM = 42, ACT_DIM = 8, GOAL_DIM = 6;
enum { G_Q = 0, G_QD = 18, G_FT = 36, G_BP = 38, G_BQ = 44, G_BV = 52, G_BW = 58, G_GOAL = 64, G_LT = 70, G_LP = 86,
       G_TOUCH = 94, G_MUG = 96, G_STEPS = 98, G_EPISODE = 99 };
// LDS columns behind the arm's S | T | A_hh (the arm's table-slot columns are reused)
constexpr int TP_W = 16;                       // r3 lam3 vt Kxy Kxz Kyy Kyz Kzz e0 e1 e2 id
constexpr int LDS_TP = xk::LDS_TBL;            // 2 sticks x 4 support slots
constexpr int BB_W = 22, BB_PAIR = 6 + 4 * BB_W; // n3 t1_3, then 4 x (rA3 rB3 lam3 vt invd3 Kn3 Kt1_3 Kt2_3)
constexpr int LDS_BB = LDS_TP + NOBJ * 4 * TP_W;
constexpr int LDS_CLIP = LDS_BB + BB_PAIR;     // 3 x 8 x 3 floats: polygon ping-pong + kept points of box_box
constexpr int LDS_FLOATS = LDS_CLIP + 72;      // 411 floats

// one lane = one arm: q/qd/ft/lam_p/touch/mug are the arm's, everything else is the lane's copy of the shared state
template <typename T> struct Lane {
    T q[9], qd[9], ft;
    T bp[NOBJ][3], bq[NOBJ][4], bv[NOBJ][3], bw[NOBJ][3];
    T goal[NOBJ][3];
    T lam_t[NOBJ][8];
    T lam_p[4];
    T touch, mug, steps, episode;
};

template <typename T> XARM_HD V3<T> ldv(const T (&a)[3]) { return mk<T>(a[0], a[1], a[2]); }

// ---------------------------------------------------------------------------------------------
// one p.stepSimulation() at timeStep 1/240 (no internal substeps, :28-29,131-132)
// Scene = xh::HandoverStandScene (config['use_stand'], :391-392): one static stand per goal; every stick's most downward face
// against the top of EITHER stand gives up to four more candidates of that stick's <= 4 point support manifold, after its
// corners, stand 0 before stand 1 (ids 8..15, no warm start) - the one-stick stand model per (stick, stand) pair.
template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD void substep(Lane<T> &L, const T (&qt)[9], Lds lds, const int arm, const Xchg xchg) {
    const T dt = (T)xm::HO_TIME_STEP;
    const T idt = (T)1 / dt;
    xk::ArmDyn<T> AD;
    xk::arm_dynamics<T, Lds, HandoverScene>(L.q, L.qd, dt, lds, arm, AD);
    T (&Minv)[45] = AD.Minv;
    T (&dq)[9] = AD.dq;
    const V3<T> hc0 = AD.hc0, hc1 = AD.hc1, hc2 = AD.hc2;

    // ---------------- sticks: frames, world inverse inertia, unconstrained motion
    const T hx = (T)xm::HO_OBJ_HALF[0], hy = (T)xm::HO_OBJ_HALF[1], hz = (T)xm::HO_OBJ_HALF[2];
    const T hh[3] = {hx, hy, hz};
    const T imb = (T)(1.0 / xm::HO_OBJ_MASS);
    const T Ibx = (T)(xm::HO_OBJ_MASS / 3.0 * (xm::HO_OBJ_HALF[1] * xm::HO_OBJ_HALF[1] + xm::HO_OBJ_HALF[2] * xm::HO_OBJ_HALF[2]));
    const T Iby = (T)(xm::HO_OBJ_MASS / 3.0 * (xm::HO_OBJ_HALF[0] * xm::HO_OBJ_HALF[0] + xm::HO_OBJ_HALF[2] * xm::HO_OBJ_HALF[2]));
    const T Ibz = (T)(xm::HO_OBJ_MASS / 3.0 * (xm::HO_OBJ_HALF[0] * xm::HO_OBJ_HALF[0] + xm::HO_OBJ_HALF[1] * xm::HO_OBJ_HALF[1]));
    V3<T> cb[NOBJ], Rb[NOBJ][3], vb[NOBJ], wb[NOBJ];
    T Iinv[NOBJ][6];
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        const T x = L.bq[o][0], y = L.bq[o][1], z = L.bq[o][2], w = L.bq[o][3];
        const V3<T> b0 = mk<T>((T)1 - (T)2 * (y * y + z * z), (T)2 * (x * y + z * w), (T)2 * (x * z - y * w));
        const V3<T> b1 = mk<T>((T)2 * (x * y - z * w), (T)1 - (T)2 * (x * x + z * z), (T)2 * (y * z + x * w));
        const V3<T> b2 = mk<T>((T)2 * (x * z + y * w), (T)2 * (y * z - x * w), (T)1 - (T)2 * (x * x + y * y));
        Rb[o][0] = b0; Rb[o][1] = b1; Rb[o][2] = b2;
        cb[o] = ldv(L.bp[o]);
        {
            const T ix = (T)1 / Ibx, iy = (T)1 / Iby, iz = (T)1 / Ibz;
            Iinv[o][0] = b0.x * b0.x * ix + b1.x * b1.x * iy + b2.x * b2.x * iz;
            Iinv[o][1] = b0.x * b0.y * ix + b1.x * b1.y * iy + b2.x * b2.y * iz;
            Iinv[o][2] = b0.x * b0.z * ix + b1.x * b1.z * iy + b2.x * b2.z * iz;
            Iinv[o][3] = b0.y * b0.y * ix + b1.y * b1.y * iy + b2.y * b2.y * iz;
            Iinv[o][4] = b0.y * b0.z * ix + b1.y * b1.z * iy + b2.y * b2.z * iz;
            Iinv[o][5] = b0.z * b0.z * ix + b1.z * b1.z * iy + b2.z * b2.z * iz;
        }
        vb[o] = ldv(L.bv[o]); wb[o] = ldv(L.bw[o]);
        {
            // gyroscopic term as btRigidBody::computeGyroscopicImpulseImplicit_Body (see xk::substep)
            const V3<T> wl = mk<T>(dot(b0, wb[o]), dot(b1, wb[o]), dot(b2, wb[o]));
            const V3<T> iw = mk<T>(Ibx * wl.x, Iby * wl.y, Ibz * wl.z);
            const V3<T> f = cross(wl, iw) * dt;
            const T J00 = Ibx, J01 = dt * (-wl.z * Iby + iw.z), J02 = dt * (wl.y * Ibz - iw.y);
            const T J10 = dt * (wl.z * Ibx - iw.z), J11 = Iby, J12 = dt * (-wl.x * Ibz + iw.x);
            const T J20 = dt * (-wl.y * Ibx + iw.y), J21 = dt * (wl.x * Iby - iw.x), J22 = Ibz;
            const T c00 = J11 * J22 - J12 * J21, c01 = J12 * J20 - J10 * J22, c02 = J10 * J21 - J11 * J20;
            const T id = (T)1 / (J00 * c00 + J01 * c01 + J02 * c02);
            const V3<T> xx = mk<T>((f.x * c00 + f.y * (J02 * J21 - J01 * J22) + f.z * (J01 * J12 - J02 * J11)) * id,
                                   (f.x * c01 + f.y * (J00 * J22 - J02 * J20) + f.z * (J02 * J10 - J00 * J12)) * id,
                                   (f.x * c02 + f.y * (J01 * J20 - J00 * J21) + f.z * (J00 * J11 - J01 * J10)) * id);
            const V3<T> wn = wl - xx;
            wb[o] = b0 * wn.x + b1 * wn.y + b2 * wn.z;
        }
        vb[o].z -= dt * (T)xm::GRAVITY;
        vb[o] = vb[o] * (T)xm::HO_LIN_DAMP_FACTOR;
        wb[o] = wb[o] * (T)xm::HO_ANG_DAMP_FACTOR;
    }

    // ---------------- (T) stick corners against the table tops / the ground: first <= 4 active corners per stick -> LDS
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
            const V3<T> r = Rb[o][0] * ((i & 1) ? hx : -hx) + Rb[o][1] * ((i & 2) ? hy : -hy) + Rb[o][2] * ((i & 4) ? hz : -hz);
            const V3<T> p = cb[o] + r;
            T hsup;
            HandoverScene::template support<T>(p, hsup);
            const T dist = p.z - hsup;
            const bool act = dist < (T)xm::SOLVER_MARGIN && cnt < 4;
            if (act) {
                const int base = LDS_TP + (o * 4 + cnt) * TP_W;
                const T l0 = (T)xm::WARMSTART * L.lam_t[o][i];
                // K = 1/m + C^T Iinv C, C = [r]x, columns c_j = r x e_j
                const V3<T> cx = mk<T>((T)0, r.z, -r.y), cy = mk<T>(-r.z, (T)0, r.x), cz = mk<T>(r.y, -r.x, (T)0);
                const V3<T> wx = symmul(Iinv[o], cx), wy = symmul(Iinv[o], cy), wz = symmul(Iinv[o], cz);
                const T K0 = imb + dot(cx, wx), K3 = imb + dot(cy, wy), K5 = imb + dot(cz, wz);
                lds[base + 0] = r.x; lds[base + 1] = r.y; lds[base + 2] = r.z;
                lds[base + 3] = l0; lds[base + 4] = (T)0; lds[base + 5] = (T)0;
                lds[base + 6] = dist < (T)0 ? -(T)xm::CONTACT_ERP * dist * idt : -dist * idt;
                lds[base + 7] = dot(cx, wy); lds[base + 8] = dot(cx, wz); lds[base + 9] = K3;
                lds[base + 10] = dot(cy, wz); lds[base + 11] = K5;
                lds[base + 12] = (T)1 / K5;   // n = +z
                lds[base + 13] = (T)1 / K3;   // t1 = -y
                lds[base + 14] = (T)1 / K0;   // t2 = +x
                lds[base + 15] = (T)i;
                // warm start
                const V3<T> fi = mk<T>((T)0, (T)0, l0);
                vb[o] = vb[o] + fi * imb;
                wb[o] = wb[o] + symmul(Iinv[o], cross(r, fi));
                cnt++;
            }
        }
        if constexpr (Scene::HAS_STAND) {
#pragma unroll
            for (int j = 0; j < NOBJ; j++) {
                V3<T> sp[4];
                T sd[4];
                Scene::template stand_points<T>(L.goal[j], cb[o], Rb[o][0], Rb[o][1], Rb[o][2], sp, sd);
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const T dist = sd[v];
                    const bool act = dist < (T)xm::SOLVER_MARGIN && dist > (T)Scene::STAND_MIN_GAP && cnt < 4;
                    if (act) {
                        const int base = LDS_TP + (o * 4 + cnt) * TP_W;
                        const V3<T> r = sp[v] - cb[o];
                        const V3<T> cx = mk<T>((T)0, r.z, -r.y), cy = mk<T>(-r.z, (T)0, r.x), cz = mk<T>(r.y, -r.x, (T)0);
                        const V3<T> wx = symmul(Iinv[o], cx), wy = symmul(Iinv[o], cy), wz = symmul(Iinv[o], cz);
                        const T K0 = imb + dot(cx, wx), K3 = imb + dot(cy, wy), K5 = imb + dot(cz, wz);
                        lds[base + 0] = r.x; lds[base + 1] = r.y; lds[base + 2] = r.z;
                        lds[base + 3] = (T)0; lds[base + 4] = (T)0; lds[base + 5] = (T)0;
                        lds[base + 6] = dist < (T)0 ? -(T)xm::CONTACT_ERP * dist * idt : -dist * idt;
                        lds[base + 7] = dot(cx, wy); lds[base + 8] = dot(cx, wz); lds[base + 9] = K3;
                        lds[base + 10] = dot(cy, wz); lds[base + 11] = K5;
                        lds[base + 12] = (T)1 / K5;
                        lds[base + 13] = (T)1 / K3;
                        lds[base + 14] = (T)1 / K0;
                        lds[base + 15] = (T)(8 + 4 * j + v);
                        cnt++;
                    }
                }
            }
        }
    }

    // ---------------- (BB) stick 0 / stick 1 manifold -> LDS
    const T mu_bb = (T)(xm::MU_OBJECT * xm::MU_OBJECT);
    bool bb_act = false;
    {
        const int base = LDS_BB;
#pragma unroll
        for (int k = 0; k < BB_PAIR; k++) lds[base + k] = (T)0;
        const V3<T> d = cb[0] - cb[1];
        // bounding spheres: 2 * |half extents| + margin
        const T reach = (T)(2.0 * 0.08291561975888499 + xm::SOLVER_MARGIN);   // |(0.075, 0.025, 0.025)| = 0.0829156...
        static_assert(xm::HO_OBJ_HALF[0] == 0.075 && xm::HO_OBJ_HALF[1] == 0.025 && xm::HO_OBJ_HALF[2] == 0.025, "bounding radius above");
        const bool near = dot(d, d) < reach * reach;
        if (XARM_ANY(near)) {
            V3<T> pts[4], nrm = mk<T>(0, 0, 1);
            T dist[4];
            const int np = near ? xs::box_box<T, Lds, LDS_CLIP>(cb[0], Rb[0], hh, cb[1], Rb[1], hh, (T)xm::SOLVER_MARGIN, pts, nrm, dist, lds) : 0;
            if (np > 0) {
                bb_act = true;
                const V3<T> t1 = xk::plane_space(nrm), t2 = cross(nrm, t1);
                lds[base + 0] = nrm.x; lds[base + 1] = nrm.y; lds[base + 2] = nrm.z;
                lds[base + 3] = t1.x; lds[base + 4] = t1.y; lds[base + 5] = t1.z;
                for (int q = 0; q < np; q++) {
                    const int pb = base + 6 + q * BB_W;
                    const V3<T> rA = pts[q] - cb[0], rB = pts[q] - cb[1];
                    lds[pb + 0] = rA.x; lds[pb + 1] = rA.y; lds[pb + 2] = rA.z;
                    lds[pb + 3] = rB.x; lds[pb + 4] = rB.y; lds[pb + 5] = rB.z;
                    lds[pb + 9] = dist[q] < (T)0 ? -(T)xm::CONTACT_ERP * dist[q] * idt : -dist[q] * idt;
                    // point Delassus block K = 2/m 1 - [rA]x IinvA [rA]x - [rB]x IinvB [rB]x; K d for the three rows
                    const V3<T> dirs[3] = {nrm, t1, t2};
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const V3<T> Kd = dirs[k] * ((T)2 * imb) - cross(rA, symmul(Iinv[0], cross(rA, dirs[k]))) - cross(rB, symmul(Iinv[1], cross(rB, dirs[k])));
                        lds[pb + 10 + k] = (T)1 / dot(dirs[k], Kd);
                        lds[pb + 13 + 3 * k] = Kd.x; lds[pb + 14 + 3 * k] = Kd.y; lds[pb + 15 + 3 * k] = Kd.z;
                    }
                }
            }
        }
    }

    // ---------------- (M) motors, (L) limits, (G) gear: row constants (as the one-stick scene)
    T m_vt[9], m_invd[9], m_lam[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        m_vt[i] = (T)xm::MOTOR_KP * (qt[i] - L.q[i]) * idt + (T)(1.0 - xm::MOTOR_KD) * dq[i];
        m_invd[i] = (T)1 / Minv[tri(i, i)];
        m_lam[i] = (T)0;
    }
    const T m_hi_arm = (T)(xm::ARM_MOTOR_FORCE * HandoverScene::TIME_STEP), m_hi_fin = (T)(HandoverScene::FINGER_MOTOR_FORCE * HandoverScene::TIME_STEP);
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
    const T g_hi = (T)(xm::GEAR_MAX_FORCE * HandoverScene::TIME_STEP);
    const T g_invd = (T)1 / (Minv[tri(7, 7)] - (T)2 * Minv[tri(8, 7)] + Minv[tri(8, 8)]);
    T g_lam = (T)0;

    // ---------------- (F) finger pad spheres, each against its nearest stick; grasp flag = both fingers within the
    // contact margin of stick 0 (getContactPoints(xarm, self.legos[0], finger), :263-264)
    constexpr int NP = xk::NP;
    PadPoint<T> pp[NP];
    int pc[NP];
    bool pad_any = false;
    bool touch_f[2] = {false, false};
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
                // sphere against stick o, in the stick's axes
                const V3<T> d = c - cb[o];
                const V3<T> cl = mk<T>(dot(Rb[o][0], d), dot(Rb[o][1], d), dot(Rb[o][2], d));
                const V3<T> ql = mk<T>(clampT(cl.x, -hx, hx), clampT(cl.y, -hy, hy), clampT(cl.z, -hz, hz));
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
                    const T px = hx - xk::xabs(cl.x), py = hy - xk::xabs(cl.y), pz = hz - xk::xabs(cl.z);
                    int k = 0;
                    T bestp = px;
                    if (py < bestp) { bestp = py; k = 1; }
                    if (pz < bestp) { bestp = pz; k = 2; }
                    const T clk = k == 0 ? cl.x : (k == 1 ? cl.y : cl.z);
                    const T s1 = clk < (T)0 ? (T)-1 : (T)1;
                    nl = mk<T>(k == 0 ? s1 : (T)0, k == 1 ? s1 : (T)0, k == 2 ? s1 : (T)0);
                    di = -bestp - (T)xm::PAD_RADIUS;
                    pl = mk<T>(k == 0 ? s1 * hx : cl.x, k == 1 ? s1 * hy : cl.y, k == 2 ? s1 * hz : cl.z);
                }
                if (o == 0) touch_f[fk] = touch_f[fk] || (di < (T)xm::CONTACT_MARGIN);
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
                const V3<T> cc = selv(co == 0, cb[0], cb[1]);
                const V3<T> r = P.p - cc;
                T Ic[6];
#pragma unroll
                for (int k = 0; k < 6; k++) Ic[k] = co == 0 ? Iinv[0][k] : Iinv[1][k];
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
                    const V3<T> vbj = ej * imb - cross(r, symmul(Ic, cross(r, ej)));
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
                // warm start: +lam0 n on the finger, -lam0 n on the stick
                const V3<T> fi = P.n * P.lam[0];
                const V3<T> mo = cross(P.p, fi);
                wtot[0] += mo.x; wtot[1] += mo.y; wtot[2] += mo.z;
                wtot[3] += fi.x; wtot[4] += fi.y; wtot[5] += fi.z;
                wtot[6 + fk] += dot(af, fi);
                const V3<T> dv = fi * imb, dw = symmul(Ic, cross(r, fi));
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
        }
        // the sticks also receive the warm-start impulses of the other arm's pads; afterwards both lanes must hold
        // bit-identical stick velocities: take arm 0's sums
#pragma unroll
        for (int o = 0; o < NOBJ; o++) {
            const V3<T> dv = vb[o] - vb_pre[o], dw = wb[o] - wb_pre[o];
            vb[o] = vb[o] + mk<T>(xchg.partner(dv.x), xchg.partner(dv.y), xchg.partner(dv.z));
            wb[o] = wb[o] + mk<T>(xchg.partner(dw.x), xchg.partner(dw.y), xchg.partner(dw.z));
            vb[o] = mk<T>(xchg.from0(vb[o].x), xchg.from0(vb[o].y), xchg.from0(vb[o].z));
            wb[o] = mk<T>(xchg.from0(wb[o].x), xchg.from0(wb[o].y), xchg.from0(wb[o].z));
        }
    }
    L.touch = (touch_f[0] && touch_f[1]) ? (T)1 : (T)0;
    // sticks touched by this arm's active pads / by the partner arm's; `seq` is wave-uniform
    int mymask = 0;
#pragma unroll
    for (int idx = 0; idx < NP; idx++) mymask |= pp[idx].invd[0] != (T)0 ? (1 << pc[idx]) : 0;
    const int othermask = (int)xchg.partner((T)mymask);
    const bool seq = XARM_ANY_X((mymask & othermask) != 0);
    XARM_LDS_FENCE();

    bool la_lane = false;
#pragma unroll
    for (int i = 0; i < 7; i++) la_lane = la_lane || la_sg[i] != (T)0;
    const bool la_wave = XARM_ANY(la_lane);
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
    const T mu_p = (T)xm::MU_OBJECT * (L.mug > (T)0.5 ? (T)xm::MU_FINGER_GRASP : (T)xm::MU_FINGER);
#pragma unroll 1
    for (int it = 0; it < XK_SWEEP_ITERS; it++) {
        XARM_LDS_FENCE();
        // (T) n = +z, t1 = -y, t2 = +x (btPlaneSpace1 of (0,0,1)); an empty slot is a no-op (1/diag = 0)
#pragma unroll
        for (int o = 0; o < NOBJ; o++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int base = LDS_TP + (o * 4 + s) * TP_W;
                const V3<T> r = mk<T>(lds[base + 0], lds[base + 1], lds[base + 2]);
                T l0 = lds[base + 3], l1 = lds[base + 4], l2 = lds[base + 5];
                const T Kxy = lds[base + 7], Kxz = lds[base + 8], Kyy = lds[base + 9], Kyz = lds[base + 10], Kzz = lds[base + 11];
                V3<T> u = vb[o] + cross(wb[o], r);
                T dl = (lds[base + 6] - u.z) * lds[base + 12];
                T nl = l0 + dl;
                nl = xk::smax0(nl);
                dl = nl - l0; l0 = nl;
                V3<T> fi = mk<T>((T)0, (T)0, dl);
                u = u + mk<T>(Kxz, Kyz, Kzz) * dl;
                const T lim = mu_t * l0;
                dl = u.y * lds[base + 13];        // t1 = -y: jv = -u.y, target 0
                nl = xk::sclamp(l1 + dl, -lim, lim);
                dl = nl - l1; l1 = nl;
                fi.y = -dl;
                u = u - mk<T>(Kxy, Kyy, Kyz) * dl;
                dl = -u.x * lds[base + 14];       // t2 = +x
                nl = xk::sclamp(l2 + dl, -lim, lim);
                dl = nl - l2; l2 = nl;
                fi.x = dl;
                vb[o] = vb[o] + fi * imb;
                wb[o] = wb[o] + symmul(Iinv[o], cross(r, fi));
                lds[base + 3] = l0; lds[base + 4] = l1; lds[base + 5] = l2;
            }
        // the arm rows (M L G) only touch this lane's joints and the T / BB rows only the sticks: the result is the
        // oracle's T, BB, MLG order
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
            if (!la_row[i]) continue;   // wave-uniform, decided once per tick
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
        // (BB) stick / stick points: A = stick 0, B = stick 1, normal from B to A
        if (XARM_ANY(bb_act)) {
            const int base = LDS_BB;
            const V3<T> n = mk<T>(lds[base + 0], lds[base + 1], lds[base + 2]), t1 = mk<T>(lds[base + 3], lds[base + 4], lds[base + 5]);
            const V3<T> t2 = cross(n, t1);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int pb = base + 6 + q * BB_W;
                const T e0 = lds[pb + 10];
                const V3<T> rA = mk<T>(lds[pb + 0], lds[pb + 1], lds[pb + 2]), rB = mk<T>(lds[pb + 3], lds[pb + 4], lds[pb + 5]);
                T lam[3] = {lds[pb + 6], lds[pb + 7], lds[pb + 8]};
                const T ed[3] = {e0, lds[pb + 11], lds[pb + 12]};
                const T vt = lds[pb + 9];
                V3<T> u = vb[0] + cross(wb[0], rA) - vb[1] - cross(wb[1], rB);
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
                vb[0] = vb[0] + f * imb; wb[0] = wb[0] + symmul(Iinv[0], cross(rA, f));
                vb[1] = vb[1] - f * imb; wb[1] = wb[1] - symmul(Iinv[1], cross(rB, f));
                lds[pb + 6] = lam[0]; lds[pb + 7] = lam[1]; lds[pb + 8] = lam[2];
            }
        }
        // (F) pad points.  Sequential form: arm 0's pads, hand the stick velocities over, arm 1's pads.  When no stick
        // of any environment in the wavefront is touched by both arms the two sweeps act on disjoint variables and
        // commute, so both lanes sweep at once (phase 0) and each stick is then taken from the lane that touched it.
#pragma unroll
        for (int ph = 0; ph < 2; ph++) {
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
                for (int idx = 0; idx < NP; idx++) {
                    PadPoint<T> &P = pp[idx];
                    if (!XARM_ANY(P.invd[0] != (T)0 && mine)) continue;
                    const T e0 = mine ? P.invd[0] : (T)0, e1 = mine ? P.invd[1] : (T)0, e2 = mine ? P.invd[2] : (T)0;
                    const int fk = idx / xm::NPAD, co = pc[idx];
                    const V3<T> af = hc1 * (fk == 0 ? (T)1 : (T)-1);
                    const V3<T> r = P.p - selv(co == 0, cb[0], cb[1]);
                    const V3<T> vc = selv(co == 0, vb[0], vb[1]), wc = selv(co == 0, wb[0], wb[1]);
                    const V3<T> t2 = cross(P.n, P.t1);
                    V3<T> u = mk<T>(y[3], y[4], y[5]) + cross(mk<T>(y[0], y[1], y[2]), P.p) + af * yf[fk] - vc - cross(wc, r);
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
                    // apply the block impulse: +fi on finger fk at p, -fi on the stick
                    const V3<T> mo = cross(P.p, fi);
                    const T W[6] = {mo.x, mo.y, mo.z, fi.x, fi.y, fi.z};
                    const T wf = dot(af, fi);
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
                    T Ic[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) Ic[k] = co == 0 ? Iinv[0][k] : Iinv[1][k];
                    const V3<T> dv = fi * imb, dw = symmul(Ic, cross(r, fi));
#pragma unroll
                    for (int o = 0; o < NOBJ; o++) {
                        vb[o] = co == o ? vb[o] - dv : vb[o];
                        wb[o] = co == o ? wb[o] - dw : wb[o];
                    }
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
            // hand the stick velocities over.  Three wave-uniform cases, the same values as one select cascade over all of them (which cost
            // 36 instructions per stick and sweep in every wavefront, pads or not): sequential - take arm 0's after phase 0, arm 1's after
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
                        const bool take = ((othermask >> o) & 1) != 0;   // the partner lane touched stick o, this one did not
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
            const int id = (int)lds[base + 15];
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

template <typename T> XARM_HD V3<T> eef_pos(const Lane<T> &L, int arm) {
    Frame<T> f = HandoverScene::base_frame<T>(arm);
#pragma unroll
    for (int i = 0; i < 7; i++) xk::fk_advance(f, i, L.q[i]);
    return f.o;
}
template <typename T> XARM_HD void ik(const Lane<T> &L, int arm, V3<T> target, T (&qt)[9]) {
    T qa[7], qo[7];
#pragma unroll
    for (int i = 0; i < 7; i++) qa[i] = L.q[i];
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
        xk::fk_advance(f, i, L.q[i]);
        w = w + f.c2 * L.qd[i];
        v = v + cross(f.o, f.c2) * L.qd[i];
    }
    const V3<T> hp = f.o + f.c0 * (T)xm::HAND_COM[0] + f.c1 * (T)xm::HAND_COM[1] + f.c2 * (T)xm::HAND_COM[2];
    const V3<T> hv = v + cross(w, hp);
    o[0] = hp.x - (T)xm::HO_EEF2GRIP[0]; o[1] = hp.y - (T)xm::HO_EEF2GRIP[1]; o[2] = hp.z - (T)xm::HO_EEF2GRIP[2];
    o[3] = hv.x; o[4] = hv.y; o[5] = hv.z;
    o[6] = L.q[7]; o[7] = L.qd[7];
}

template <typename T> XARM_HD void block(const EnvCfg &cfg, int64_t env, int64_t episode, int b, T (&u)[4]) {
    const uint64_t gid = (uint64_t)(cfg.env_id_offset + env);
    uint32_t o[4];
    xk::philox(cfg.seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, (uint32_t)b, o);
#pragma unroll
    for (int k = 0; k < 4; k++) u[k] = xk::u01<T>(o[k]);
}
// _reset_sim's spawn (:354-363).  Stick 0: the N = 1 draws.  Stick 1: attempt k = Philox block 2 + k, the first whose y is
// at least spawn_min_dy away from stick 0's y (:357-360 compares y only); mirror coin = word 3 of block 1.  The y test is
// made on the uniforms (exact in float32 and float64 alike) against a float-rounded threshold, as the oracle does.
template <typename T> XARM_HD void sample_object(const EnvCfg &cfg, int64_t env, int64_t episode, Lane<T> &L) {
    T u0[4], u1[4], ua[4] = {(T)0, (T)0, (T)0, (T)0};
    block(cfg, env, episode, 0, u0);
    block(cfg, env, episode, 1, u1);
    const T wx = (T)(xm::HO_OBJ_HIGH[0] - xm::HO_OBJ_LOW[0]), wy = (T)(xm::HO_OBJ_HIGH[1] - xm::HO_OBJ_LOW[1]);
    const T thr = (T)(float)(xm::HO_SPAWN_MIN_DY / (xm::HO_OBJ_HIGH[1] - xm::HO_OBJ_LOW[1]));
    const T x0 = (T)xm::HO_OBJ_LOW[0] + u0[0] * wx;
    L.bp[0][0] = u0[2] < (T)0.5 ? -x0 : x0;
    L.bp[0][1] = (T)xm::HO_OBJ_LOW[1] + u0[1] * wy;
    bool ok = false;
#pragma unroll 1
    for (int k = 0; k < xm::HO_SAMPLE_MAX_TRIES && !ok; k++) {
        block(cfg, env, episode, 2 + k, ua);
        ok = !(xk::xabs(ua[1] - u0[1]) < thr);
    }
    const T x1 = (T)xm::HO_OBJ_LOW[0] + ua[0] * wx;
    T y1 = (T)xm::HO_OBJ_LOW[1] + ua[1] * wy;
    if (!ok) {
        const T y0 = L.bp[0][1];
        y1 = y0 + (T)xm::HO_SPAWN_MIN_DY <= (T)xm::HO_OBJ_HIGH[1] ? y0 + (T)xm::HO_SPAWN_MIN_DY : y0 - (T)xm::HO_SPAWN_MIN_DY;
    }
    L.bp[1][0] = u1[3] < (T)0.5 ? -x1 : x1;
    L.bp[1][1] = y1;
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        L.bp[o][2] = (T)xm::HO_HEIGHT_OFFSET;
        L.bq[o][0] = L.bq[o][1] = L.bq[o][2] = (T)0; L.bq[o][3] = (T)1;
#pragma unroll
        for (int k = 0; k < 3; k++) { L.bv[o][k] = (T)0; L.bw[o][k] = (T)0; }
#pragma unroll
        for (int k = 0; k < 8; k++) L.lam_t[o][k] = (T)0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) L.lam_p[k] = (T)0;
}
// _sample_goal (:370-393).  Goal 0: the N = 1 draws, no rejection.  Goal 1: attempt k = block 2 + max_tries + k (x y z +
// same-side coin), the first whose y is at least goal_min_dy from goal 0's y and whose xy is at least goal_min_obj_dist
// from the xy of EVERY stick, with the goal's x still positive (:375-379 test before the side flip of :380-382).
template <typename T> XARM_HD void sample_goal(const EnvCfg &cfg, int64_t env, int64_t episode, Lane<T> &L) {
    T u0[4], u1[4], ua[4] = {(T)0, (T)0, (T)0, (T)0}, g[3] = {(T)0, (T)0, (T)0};
    block(cfg, env, episode, 0, u0);
    block(cfg, env, episode, 1, u1);
    const T w[3] = {(T)(xm::HO_GOAL_HIGH[0] - xm::HO_GOAL_LOW[0]), (T)(xm::HO_GOAL_HIGH[1] - xm::HO_GOAL_LOW[1]), (T)(xm::HO_GOAL_HIGH[2] - xm::HO_GOAL_LOW[2])};
    const T thr = (T)(float)(xm::HO_GOAL_MIN_DY / (xm::HO_GOAL_HIGH[1] - xm::HO_GOAL_LOW[1]));
    const T ug0[3] = {u0[3], u1[0], u1[1]};
#pragma unroll
    for (int k = 0; k < 3; k++) L.goal[0][k] = (T)xm::HO_GOAL_LOW[k] + ug0[k] * w[k];
    if ((L.bp[0][0] > (T)0) != (u1[2] < (T)cfg.same_side_rate)) L.goal[0][0] = -L.goal[0][0];
    bool ok = false;
#pragma unroll 1
    for (int k = 0; k < xm::HO_SAMPLE_MAX_TRIES && !ok; k++) {
        block(cfg, env, episode, 2 + xm::HO_SAMPLE_MAX_TRIES + k, ua);
#pragma unroll
        for (int j = 0; j < 3; j++) g[j] = (T)xm::HO_GOAL_LOW[j] + ua[j] * w[j];
        ok = !(xk::xabs(ua[1] - ug0[1]) < thr);
#pragma unroll
        for (int o = 0; o < NOBJ; o++) {
            const T dx = g[0] - L.bp[o][0], dy = g[1] - L.bp[o][1];
            if (xk::xsqrt(dx * dx + dy * dy) < (T)xm::HO_GOAL_MIN_OBJ_DIST) ok = false;
        }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) L.goal[1][j] = g[j];
    if ((L.bp[1][0] > (T)0) != (ua[3] < (T)cfg.same_side_rate)) L.goal[1][0] = -L.goal[1][0];
    if (cfg.goal_shape == 1) L.goal[0][2] = L.goal[1][2] = (T)xm::HO_HEIGHT_OFFSET;
}
template <typename T> XARM_HD void lane_init(const EnvCfg &cfg, int64_t env, Lane<T> &L) {
#pragma unroll
    for (int i = 0; i < 9; i++) { L.q[i] = (T)xm::HO_JOINT_INIT_POS[i]; L.qd[i] = (T)0; }
    L.ft = (T)xm::HO_JOINT_INIT_POS[7];
    L.touch = L.mug = L.steps = L.episode = (T)0;
    sample_object(cfg, env, 0, L);
    sample_goal(cfg, env, 0, L);
}
template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD void lane_reset(const EnvCfg &cfg, int64_t env, Lane<T> &L, int arm, Lds lds, Xchg x) {
    const int64_t episode = (int64_t)L.episode + 1;
    T qt[9];
    const V3<T> home = arm == 0 ? mk<T>((T)xm::HO_EFF_INIT_POS[0][0], (T)xm::HO_EFF_INIT_POS[0][1], (T)xm::HO_EFF_INIT_POS[0][2])
                                : mk<T>((T)xm::HO_EFF_INIT_POS[1][0], (T)xm::HO_EFF_INIT_POS[1][1], (T)xm::HO_EFF_INIT_POS[1][2]);
#pragma unroll 1
    for (int k = 0; k <= xm::HO_RESET_TICKS; k++) {
        if (k < xm::HO_RESET_TICKS) ik(L, arm, home, qt);
        else sample_object(cfg, env, episode, L);
        substep<T, Lds, Xchg, Scene>(L, qt, lds, arm, x);
    }
    sample_goal(cfg, env, episode, L);
    L.steps = (T)0;
    L.episode = (T)episode;
}
// act = this arm's 4 action entries (:249-256)
template <typename T, typename Lds, typename Xchg, typename Scene = HandoverScene>
XARM_HD void lane_step(Lane<T> &L, int arm, const T (&act)[4], T &reward, bool &done, bool &success, Lds lds, Xchg x) {
    L.steps += (T)1;
    T a[4], qt[9];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    const V3<T> cur = eef_pos(L, arm);
    const T sc = (T)(xm::HO_MAX_VEL * xm::HO_ACTION_DT);
    const V3<T> lo = arm == 0 ? mk<T>((T)xm::HO_POS_LOW[0][0], (T)xm::HO_POS_LOW[0][1], (T)xm::HO_POS_LOW[0][2])
                              : mk<T>((T)xm::HO_POS_LOW[1][0], (T)xm::HO_POS_LOW[1][1], (T)xm::HO_POS_LOW[1][2]);
    const V3<T> hi = arm == 0 ? mk<T>((T)xm::HO_POS_HIGH[0][0], (T)xm::HO_POS_HIGH[0][1], (T)xm::HO_POS_HIGH[0][2])
                              : mk<T>((T)xm::HO_POS_HIGH[1][0], (T)xm::HO_POS_HIGH[1][1], (T)xm::HO_POS_HIGH[1][2]);
    const V3<T> target = mk<T>(clampT(cur.x + a[0] * sc, lo.x, hi.x), clampT(cur.y + a[1] * sc, lo.y, hi.y), clampT(cur.z + a[2] * sc, lo.z, hi.z));
    L.ft = clampT(L.q[7] + a[3] * (T)(xm::HO_ACTION_DT * xm::HO_MAX_GRIPPER_VEL), (T)xm::HO_GRIPPER_LOW, (T)xm::HO_GRIPPER_HIGH);
    ik(L, arm, target, qt);
    L.mug = L.touch; // friction toggle from the current contact points with stick 0 (:269-280)
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        // clamp every stick into the play field, keep only its pitch, zero its velocity (:282-297)
        const T qx = L.bq[o][0], qy = L.bq[o][1], qz = L.bq[o][2], qw = L.bq[o][3];
        const T sarg = (T)2 * (qw * qy - qx * qz);
        const T hp = (T)1.57079632679489661923;
        const T pitch = sarg <= (T)-0.99999 ? -hp : (sarg >= (T)0.99999 ? hp : xk::xasin(sarg));
        T sp, cp;
        xk::xsincos((T)0.5 * pitch, sp, cp);
        L.bq[o][0] = (T)0; L.bq[o][1] = sp; L.bq[o][2] = (T)0; L.bq[o][3] = cp;
        L.bp[o][0] = clampT(L.bp[o][0], -(T)xm::HO_OBJ_HIGH[0], (T)xm::HO_OBJ_HIGH[0]);
        L.bp[o][1] = clampT(L.bp[o][1], -(T)xm::HO_OBJ_HIGH[1], (T)xm::HO_OBJ_HIGH[1]);
#pragma unroll
        for (int k = 0; k < 3; k++) { L.bv[o][k] = (T)0; L.bw[o][k] = (T)0; }
    }
#pragma unroll 1
    for (int k = 0; k < xm::HO_N_TICKS; k++) substep<T, Lds, Xchg, Scene>(L, qt, lds, arm, x);
    bool all = true;
    T rew = (T)0;
#pragma unroll
    for (int o = 0; o < NOBJ; o++) {
        const T dx = L.bp[o][0] - L.goal[o][0], dy = L.bp[o][1] - L.goal[o][1], dz = L.bp[o][2] - L.goal[o][2];
        const T dist = xk::xsqrt(dx * dx + dy * dy + dz * dz);
        all = all && dist < (T)xm::HO_DISTANCE_THRESHOLD;                  // :395-402
        rew += dist > (T)xm::HO_DISTANCE_THRESHOLD ? (T)1 : (T)0;          // :177-181
    }
    success = all;
    reward = -rew;
    done = success || ((int)L.steps == xm::HO_MAX_EPISODE_STEPS);
}

} // namespace xh2
