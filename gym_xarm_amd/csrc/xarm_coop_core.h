// xarm_coop_core.h - cooperative (16 lanes per environment) form of the PickAndPlace substep.
//
// Why it exists: xarm_core.h maps one environment to one lane and is throughput-optimal when every SIMD holds a
// full wavefront of environments.  The auto-reset that follows a step (XarmPickAndPlace.reset,
// /root/reference/gym_xarm/envs/xarm_pick_and_place.py:121-127,250-267: five motor-driven ticks + one teleport
// tick) touches only the handful of environments whose episode just ended, and it is six sim ticks of pure
// latency: with one environment per lane, one wavefront crawls through ~4 M dependent instructions while 1023
// SIMDs idle (profiles/r01k: 82 % of GPU time).  Here one environment owns a DPP row of 16 lanes, so a
// wavefront holds 4 environments and the Gauss-Seidel sweep is spread over the row.
//
// Formulation (same sweep, same row order T, M, L, G, F as xk::substep and the oracle, exact in exact arithmetic):
// projected Gauss-Seidel in IMPULSE space.  Every solver row r has a Jacobian J_r over the 15 velocity
// dofs (9 joints + object twist) and the Delassus matrix A = J Minv J^T is assembled once per substep; lane i
// keeps g_i = target_i - J_i u for the rows it owns and row i of A.  Processing row r is then
//     nl = clamp(lam_r + g_r / A_rr), dl = nl - lam_r  (on the lane that owns r)
//     g_i -= A_ir * broadcast(dl)                        (one v_fmac per owned row, dl by DPP row_newbcast)
// i.e. ~6-8 instructions per row instead of the ~30-40 of the velocity-space sweep.  Rows are dealt to
// (slot, lane): slot 0 = the 12 object/table rows, slot 1 = 9 motors + 4 finger limits + gear, slot 2 = the 12
// finger-pad rows, slot 3 = the 7 arm joint limits; slots 2 and 3 are skipped when no lane of the wavefront
// needs them.  Everything that is not the sweep (kinematics, CRBA, Cholesky, collision) is computed
// redundantly by the 16 lanes of an environment with the code of xarm_core.h, so all 16 lanes carry
// bit-identical copies of the environment state.
//
// Host build: LV<T> holds the 16 lane values and the cross-lane operations are plain loops, so the float64
// instantiation can be checked against the oracle like the other cores (tests/hostbuild, tests/test_coop.py).
#pragma once
#include "xarm_core.h"

namespace xc {
using xk::V3; using xk::mk; using xk::dot; using xk::cross; using xk::symmul; using xk::tri; using xk::symi;
using xk::EnvState; using xk::EnvCfg; using xk::NTS; using xk::NP; using xk::clampT; using xk::xabs; using xk::xsqrt;
using xk::LDS_S; using xk::LDS_T;

constexpr int GL = 16;          // lanes per environment = one DPP row
constexpr int NT = 3 * NTS;     // 12 object/table rows (slot 0)
constexpr int NA1 = 14;         // slot 1: 9 motors, 4 finger-limit rows, gear
constexpr int NF = 3 * NP;      // 12 pad rows (slot 2)
constexpr int NLA = 7;          // arm joint limits (slot 3)
// column layout of the per-slot rows of A (columns a slot is never coupled to are not stored)
constexpr int C0_T = 0, C0_F = NT, C0_N = NT + NF;                                 // slot 0: T | F
constexpr int C1_A = 0, C1_L = NA1, C1_F = NA1 + NLA, C1_N = NA1 + NLA + NF;       // slot 1 / 3: arm | La | F
constexpr int C2_T = 0, C2_A = NT, C2_L = NT + NA1, C2_F = NT + NA1 + NLA, C2_N = NT + NA1 + NLA + NF;

#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
constexpr int LVN = 1;
struct Grp { int l; };   // lane index inside the 16-lane row
XARM_HD int lane_of(const Grp &G, int) { return G.l; }
template <int CTRL> XARM_HD float dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
#else
constexpr int LVN = GL;
struct Grp {};
XARM_HD int lane_of(const Grp &, int i) { return i; }
#endif

// one value per lane of the row
template <typename T> struct LV { T v[LVN]; };
#define XC_LANES for (int i_ = 0; i_ < LVN; i_++)
template <typename T> XARM_HD LV<T> lv_fill(T u) { LV<T> r; XC_LANES r.v[i_] = u; return r; }
// a compile-time constant held in a VGPR (an inline constant cannot be the DPP operand of v_mul_f32_dpp)
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
XARM_HD LV<float> lv_fill_vgpr(float u) { LV<float> r; r.v[0] = u; asm volatile("" : "+v"(r.v[0])); return r; }
#else
template <typename T> XARM_HD LV<T> lv_fill_vgpr(T u) { return lv_fill(u); }
#endif
// one fused multiply-add, never a separately rounded product: the sweep must round identically in every
// instantiation (the row set is chosen per wavefront, an environment's result must not depend on its neighbours)
XARM_HD float fm(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
XARM_HD double fm(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> XARM_HD LV<T> lv_fma(LV<T> a, LV<T> b, LV<T> c) { LV<T> r; XC_LANES r.v[i_] = fm(a.v[i_], b.v[i_], c.v[i_]); return r; }
// products and sums are rounded on their own (no contraction with a neighbouring operation: LLVM fuses a product
// into the first add of the lane reduction in one instantiation and not in the other, a 1e-6 difference in the
// near-cancelling torque sum of a resting object)
#define XC_NO_CONTRACT _Pragma("clang fp contract(off)")
template <typename T> XARM_HD LV<T> lv_sub(LV<T> a, LV<T> b) { XC_NO_CONTRACT LV<T> r; XC_LANES r.v[i_] = a.v[i_] - b.v[i_]; return r; }
template <typename T> XARM_HD LV<T> lv_mul(LV<T> a, LV<T> b) { XC_NO_CONTRACT LV<T> r; XC_LANES r.v[i_] = a.v[i_] * b.v[i_]; return r; }
template <typename T> XARM_HD LV<T> lv_add(LV<T> a, LV<T> b) { XC_NO_CONTRACT LV<T> r; XC_LANES r.v[i_] = a.v[i_] + b.v[i_]; return r; }
template <typename T> XARM_HD LV<T> lv_scale(LV<T> a, T u) { XC_NO_CONTRACT LV<T> r; XC_LANES r.v[i_] = a.v[i_] * u; return r; }
template <typename T> XARM_HD LV<T> lv_fmas(LV<T> a, T u, LV<T> c) { LV<T> r; XC_LANES r.v[i_] = fm(a.v[i_], u, c.v[i_]); return r; }
// act ? 1 / x : 0 (act is uniform over the row)
template <typename T> XARM_HD LV<T> lv_rcp_if(LV<T> x, bool act) { LV<T> r; XC_LANES r.v[i_] = act ? (T)1 / x.v[i_] : (T)0; return r; }
template <typename T> XARM_HD LV<T> lv_neg(LV<T> a) { LV<T> r; XC_LANES r.v[i_] = -a.v[i_]; return r; }
XARM_HD double med3(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
XARM_HD float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
#else
XARM_HD float med3(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
#endif
template <typename T> XARM_HD LV<T> lv_med3(LV<T> x, LV<T> lo, LV<T> hi) { LV<T> r; XC_LANES r.v[i_] = med3(x.v[i_], lo.v[i_], hi.v[i_]); return r; }
XARM_HD float max0(float x) { return __builtin_fmaxf(x, 0.0f); }   // one v_max_f32
XARM_HD double max0(double x) { return x < 0.0 ? 0.0 : x; }
template <typename T> XARM_HD LV<T> lv_max0(LV<T> x) { LV<T> r; XC_LANES r.v[i_] = max0(x.v[i_]); return r; }
// dst = src on lane L only
template <int L, typename T> XARM_HD void lv_commit(const Grp &G, LV<T> &dst, LV<T> src) {
    XC_LANES dst.v[i_] = lane_of(G, i_) == L ? src.v[i_] : dst.v[i_];
}
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
// every lane of the row receives lane L's value (DPP row_newbcast, gfx90a+)
template <int L> XARM_HD LV<float> lv_bcast(LV<float> x) { LV<float> r; r.v[0] = dpp<0x150 + L>(x.v[0]); return r; }
// sum over the 16 lanes of the row, identical in every lane: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror
XARM_HD float lv_allsum(LV<float> x) {
    XC_NO_CONTRACT
    float s = x.v[0];
    s += dpp<0xB1>(s);
    s += dpp<0x4E>(s);
    s += dpp<0x141>(s);
    s += dpp<0x140>(s);
    return s;
}
#else
template <int L, typename T> XARM_HD LV<T> lv_bcast(LV<T> x) { LV<T> r; XC_LANES r.v[i_] = x.v[L]; return r; }
template <typename T> XARM_HD T lv_allsum(LV<T> x) {
    // same butterfly order as the DPP form
    T a[GL], b[GL];
    for (int i = 0; i < GL; i++) a[i] = x.v[i];
    for (int i = 0; i < GL; i++) b[i] = a[i] + a[i ^ 1];
    for (int i = 0; i < GL; i++) a[i] = b[i] + b[i ^ 2];
    for (int i = 0; i < GL; i++) b[i] = a[i] + a[(i & 8) | (7 - (i & 7))];
    for (int i = 0; i < GL; i++) a[i] = b[i] + b[15 - i];
    return a[0];
}
#endif
template <int L, typename T> XARM_HD T lv_get(LV<T> x) { return lv_bcast<L>(x).v[0]; }
// y[l] = sum of x[k] over the lanes k >= l of the row (Hillis-Steele with DPP row_shl 1, 2, 4, 8; lanes shifted in
// from beyond the row read 0)
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
XARM_HD LV<float> lv_suffix_sum(LV<float> x) {
    XC_NO_CONTRACT
    float s = x.v[0];
    s += dpp<0x101>(s);
    s += dpp<0x102>(s);
    s += dpp<0x104>(s);
    s += dpp<0x108>(s);
    LV<float> r; r.v[0] = s; return r;
}
#else
template <typename T> XARM_HD LV<T> lv_suffix_sum(LV<T> x) {
    T a[GL], b[GL];
    for (int i = 0; i < GL; i++) a[i] = x.v[i];
    for (int sh = 1; sh < GL; sh *= 2) {
        for (int i = 0; i < GL; i++) b[i] = a[i] + (i + sh < GL ? a[i + sh] : (T)0);
        for (int i = 0; i < GL; i++) a[i] = b[i];
    }
    LV<T> r; for (int i = 0; i < GL; i++) r.v[i] = a[i]; return r;
}
#endif

// two values per lane that are always updated together: the table row and the single-joint row a lane owns (slots 0
// and 1 are never coupled to each other, so row i of both can be processed in ONE step).  Device: an aligned VGPR pair,
// v_pk_fma_f32 / v_pk_add_f32 / v_mov_b64_dpp row_newbcast - half the instructions of two scalar row steps.
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
template <typename T> struct LV2 { xk::xf2 v; };
XARM_HD LV2<float> lv2_make(LV<float> x, LV<float> y) { LV2<float> r; r.v.x = x.v[0]; r.v.y = y.v[0]; return r; }
XARM_HD LV<float> lv2_x(LV2<float> a) { LV<float> r; r.v[0] = a.v.x; return r; }
XARM_HD LV<float> lv2_y(LV2<float> a) { LV<float> r; r.v[0] = a.v.y; return r; }
XARM_HD LV2<float> lv2_fma(LV2<float> a, LV2<float> b, LV2<float> c) { LV2<float> r; r.v = __builtin_elementwise_fma(a.v, b.v, c.v); return r; }
XARM_HD LV2<float> lv2_sub(LV2<float> a, LV2<float> b) { XC_NO_CONTRACT LV2<float> r; r.v = a.v - b.v; return r; }
XARM_HD LV2<float> lv2_mul(LV2<float> a, LV2<float> b) { XC_NO_CONTRACT LV2<float> r; r.v = a.v * b.v; return r; }
template <int L> XARM_HD void lv2_commit(const Grp &G, LV2<float> &dst, LV2<float> src) { dst.v = G.l == L ? src.v : dst.v; }
template <int L> XARM_HD LV2<float> lv2_bcast(LV2<float> x) {
    LV2<float> r;
    const long long in = __builtin_bit_cast(long long, x.v);
    const long long out = __builtin_amdgcn_update_dpp(0ll, in, 0x150 + L, 0xf, 0xf, true);   // v_mov_b64_dpp row_newbcast
    r.v = __builtin_bit_cast(xk::xf2, out);
    return r;
}
#else
template <typename T> struct LV2 { LV<T> x, y; };
template <typename T> XARM_HD LV2<T> lv2_make(LV<T> x, LV<T> y) { LV2<T> r; r.x = x; r.y = y; return r; }
template <typename T> XARM_HD LV<T> lv2_x(LV2<T> a) { return a.x; }
template <typename T> XARM_HD LV<T> lv2_y(LV2<T> a) { return a.y; }
template <typename T> XARM_HD LV2<T> lv2_fma(LV2<T> a, LV2<T> b, LV2<T> c) { return lv2_make(lv_fma(a.x, b.x, c.x), lv_fma(a.y, b.y, c.y)); }
template <typename T> XARM_HD LV2<T> lv2_sub(LV2<T> a, LV2<T> b) { return lv2_make(lv_sub(a.x, b.x), lv_sub(a.y, b.y)); }
template <typename T> XARM_HD LV2<T> lv2_mul(LV2<T> a, LV2<T> b) { return lv2_make(lv_mul(a.x, b.x), lv_mul(a.y, b.y)); }
template <int L, typename T> XARM_HD void lv2_commit(const Grp &G, LV2<T> &dst, LV2<T> src) { lv_commit<L>(G, dst.x, src.x); lv_commit<L>(G, dst.y, src.y); }
template <int L, typename T> XARM_HD LV2<T> lv2_bcast(LV2<T> x) { return lv2_make(lv_bcast<L>(x.x), lv_bcast<L>(x.y)); }
#endif

template <typename T> XARM_HD T sel3(int k, T a, T b, T c) { return k == 0 ? a : (k == 1 ? b : c); }
template <typename T> XARM_HD T sel4(int k, T a, T b, T c, T d) { return k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d)); }
template <typename T> XARM_HD V3<T> selv3(int k, V3<T> a, V3<T> b, V3<T> c) { return mk<T>(sel3(k, a.x, b.x, c.x), sel3(k, a.y, b.y, c.y), sel3(k, a.z, b.z, c.z)); }
template <typename T> XARM_HD V3<T> selv4(int k, V3<T> a, V3<T> b, V3<T> c, V3<T> d) {
    return mk<T>(sel4(k, a.x, b.x, c.x, d.x), sel4(k, a.y, b.y, c.y, d.y), sel4(k, a.z, b.z, c.z, d.z));
}

// ---------------------------------------------------------------------------------------------
// group-uniform quantities of one substep (every lane of the row computes the same values)
template <typename T> struct Setup {
    T Minv[45], dq[9];             // arm: inverse joint-space inertia, unconstrained joint velocities
    V3<T> hc1;                     // hand y axis (finger slide direction)
    V3<T> cb, vb, wb;              // object centre, unconstrained velocities
    T Iinv[6];
    V3<T> tr[NTS];                 // table slots: corner offset, target velocity, warm start, corner id
    T tvt[NTS], tl0[NTS];
    int tid[NTS];
    T m_vt[9], la_vt[7], la_sg[7], lf_vt[2][2], g_vt;
    // pad points, DEALT to the lanes: lane l holds pad l / 3 (the pad whose row (l / 3, l % 3) it owns; lanes 12-15 repeat pad 3):
    // position, normal, first tangent, target velocity, warm start.  Only the flags are shared by the row.
    LV<T> lpp[3], lpn[3], lpt1[3], lpvt, lpl0;
    bool pact[NP];
    bool pad_any, la_any;
    T pad_cfm, mu_p;
};

// per-lane row data (scalar fields gathered into LV on the host)
enum { R_J0 = 0,      // slot 0 Jacobian, object part: d (3), r x d (3)
       R_J1 = 6,      // slot 1 Jacobian pattern over the 9 joints
       R_J2A = 15,    // slot 2 Jacobian, joints (9)
       R_J2B = 24,    // slot 2 Jacobian, object (6)
       R_J3 = 30,     // slot 3 Jacobian over the 7 arm joints
       R_G = 37,      // g[4] = target - J u_free
       R_L0 = 41,     // initial impulses [4]
       R_LO1 = 45, R_HI1 = 46, // fixed limits of the slot-1 rows
       R_CFM = 47,    // constraint-force mixing of the owned slot-2 row (pad normals)
       R_N = 48 };

// table row directions: a = 0 normal +z, 1 t1 = -y, 2 t2 = +x (btPlaneSpace1 of (0,0,1))
template <typename T> XARM_HD V3<T> tdir(int a) { return mk<T>(a == 2 ? (T)1 : (T)0, a == 1 ? (T)-1 : (T)0, a == 0 ? (T)1 : (T)0); }

// Slots 0 and 1 (table rows, single-joint rows) exist in every instantiation and are therefore built by ONE piece of
// code that runs before the row set is chosen: whatever the neighbours in the wavefront need, these values and the
// sweep's single-rounding operations on them are the same bits.
template <typename T, typename Scene = xk::PnpScene>
XARM_HD void lane_rows_base(const Setup<T> &S, int l, T (&R)[R_N]) {
    const T inf = (T)3.0e38;
#pragma unroll
    for (int k = 0; k < R_N; k++) R[k] = (T)0;
    // slot 0: table row (s, a) on lanes 0..11
    {
        const int s = l / 3, a = l - 3 * s;
        const bool own = l < NT;
        const V3<T> d = tdir<T>(a);
        const V3<T> r = selv4(s, S.tr[0], S.tr[1], S.tr[2], S.tr[3]);
        const V3<T> rd = cross(r, d);
        R[R_J0 + 0] = own ? d.x : (T)0; R[R_J0 + 1] = own ? d.y : (T)0; R[R_J0 + 2] = own ? d.z : (T)0;
        R[R_J0 + 3] = own ? rd.x : (T)0; R[R_J0 + 4] = own ? rd.y : (T)0; R[R_J0 + 5] = own ? rd.z : (T)0;
        const T vt = a == 0 ? sel4(s, S.tvt[0], S.tvt[1], S.tvt[2], S.tvt[3]) : (T)0;
        const T ju = R[R_J0 + 0] * S.vb.x + R[R_J0 + 1] * S.vb.y + R[R_J0 + 2] * S.vb.z + R[R_J0 + 3] * S.wb.x + R[R_J0 + 4] * S.wb.y + R[R_J0 + 5] * S.wb.z;
        R[R_G + 0] = own ? vt - ju : (T)0;
        R[R_L0 + 0] = (own && a == 0) ? sel4(s, S.tl0[0], S.tl0[1], S.tl0[2], S.tl0[3]) : (T)0;
    }
    // slot 1: motors 0..8, finger limits (k, side) on lanes 9..12, gear on lane 13
    {
#pragma unroll
        for (int d = 0; d < 7; d++) R[R_J1 + d] = l == d ? (T)1 : (T)0;
        R[R_J1 + 7] = (l == 7 || l == 9 || l == 13) ? (T)1 : (l == 10 ? (T)-1 : (T)0);
        R[R_J1 + 8] = (l == 8 || l == 11) ? (T)1 : ((l == 12 || l == 13) ? (T)-1 : (T)0);
        T vt = (T)0, ju = (T)0;
#pragma unroll
        for (int d = 0; d < 9; d++) { vt = l == d ? S.m_vt[d] : vt; ju += R[R_J1 + d] * S.dq[d]; }
        vt = l == 9 ? S.lf_vt[0][0] : (l == 10 ? S.lf_vt[0][1] : (l == 11 ? S.lf_vt[1][0] : (l == 12 ? S.lf_vt[1][1] : (l == 13 ? S.g_vt : vt))));
        R[R_G + 1] = l < NA1 ? vt - ju : (T)0;
        const T m_hi_arm = (T)(xm::ARM_MOTOR_FORCE * Scene::TIME_STEP), m_hi_fin = (T)(Scene::FINGER_MOTOR_FORCE * Scene::TIME_STEP);
        const T g_hi = (T)(xm::GEAR_MAX_FORCE * Scene::TIME_STEP);
        const T hi = l < 7 ? m_hi_arm : (l < 9 ? m_hi_fin : (l < 13 ? inf : (l == 13 ? g_hi : (T)0)));
        R[R_HI1] = hi;
        R[R_LO1] = (l >= 9 && l < 13) ? (T)0 : -hi;
    }
}

template <typename T, typename Lds, bool PAD, bool LA>
XARM_HD void lane_rows_extra(const Setup<T> &S, Lds lds, int l, int li, T (&R)[R_N]) {
    // slot 2: pad row (idx, a) on lanes 0..11; li = the lane's slot in the LV fields (0 on the device, l on the host)
    if (PAD) {
        const int idx = l / 3, a = l - 3 * idx;
        const bool own = l < NF;
        const int fk = idx / xm::NPAD;
        const V3<T> p = mk<T>(S.lpp[0].v[li], S.lpp[1].v[li], S.lpp[2].v[li]);
        const V3<T> n = mk<T>(S.lpn[0].v[li], S.lpn[1].v[li], S.lpn[2].v[li]);
        const V3<T> t1 = mk<T>(S.lpt1[0].v[li], S.lpt1[1].v[li], S.lpt1[2].v[li]);
        const V3<T> t2 = cross(n, t1);
        const V3<T> d = selv3(a, n, t1, t2);
        const V3<T> mo = cross(p, d);
        const V3<T> af = S.hc1 * (fk == 0 ? (T)1 : (T)-1);
        const T wf = dot(af, d);
        T ju = (T)0;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const T j = lds[LDS_S + i * 6 + 0] * mo.x + lds[LDS_S + i * 6 + 1] * mo.y + lds[LDS_S + i * 6 + 2] * mo.z +
                        lds[LDS_S + i * 6 + 3] * d.x + lds[LDS_S + i * 6 + 4] * d.y + lds[LDS_S + i * 6 + 5] * d.z;
            R[R_J2A + i] = own ? j : (T)0;
            ju += R[R_J2A + i] * S.dq[i];
        }
        R[R_J2A + 7] = (own && fk == 0) ? wf : (T)0;
        R[R_J2A + 8] = (own && fk == 1) ? wf : (T)0;
        ju += R[R_J2A + 7] * S.dq[7] + R[R_J2A + 8] * S.dq[8];
        const V3<T> rb = p - S.cb;
        const V3<T> rd = cross(rb, d);
        R[R_J2B + 0] = own ? -d.x : (T)0; R[R_J2B + 1] = own ? -d.y : (T)0; R[R_J2B + 2] = own ? -d.z : (T)0;
        R[R_J2B + 3] = own ? -rd.x : (T)0; R[R_J2B + 4] = own ? -rd.y : (T)0; R[R_J2B + 5] = own ? -rd.z : (T)0;
        ju += R[R_J2B + 0] * S.vb.x + R[R_J2B + 1] * S.vb.y + R[R_J2B + 2] * S.vb.z + R[R_J2B + 3] * S.wb.x + R[R_J2B + 4] * S.wb.y + R[R_J2B + 5] * S.wb.z;
        const T vt = a == 0 ? S.lpvt.v[li] : (T)0;
        R[R_G + 2] = own ? vt - ju : (T)0;
        R[R_L0 + 2] = (own && a == 0) ? S.lpl0.v[li] : (T)0;
        R[R_CFM] = (own && a == 0) ? S.pad_cfm : (T)0;
    }
    // slot 3: arm joint limit i on lanes 0..6
    if (LA) {
        T vt = (T)0, ju = (T)0;
#pragma unroll
        for (int d = 0; d < 7; d++) {
            R[R_J3 + d] = l == d ? S.la_sg[d] : (T)0;
            vt = l == d ? S.la_vt[d] : vt;
            ju += R[R_J3 + d] * S.dq[d];
        }
        R[R_G + 3] = l < NLA ? vt - ju : (T)0;
    }
}

// rows of -A owned by lane l against the table and single-joint rows, and the reciprocal diagonals of those rows;
// the uniform M^-1 J_r^T of such a row is cheap (object inverse inertia / columns of Minv) and formed by every lane
template <typename T> XARM_HD void arm_row_B(const Setup<T> &S, int r, T (&B)[9]) {
#pragma unroll
    for (int d = 0; d < 9; d++) {
        if (r < 9) B[d] = S.Minv[symi(d, r)];
        else if (r == 9) B[d] = S.Minv[symi(d, 7)];
        else if (r == 10) B[d] = -S.Minv[symi(d, 7)];
        else if (r == 11) B[d] = S.Minv[symi(d, 8)];
        else if (r == 12) B[d] = -S.Minv[symi(d, 8)];
        else if (r == 13) B[d] = S.Minv[symi(d, 7)] - S.Minv[symi(d, 8)];
        else B[d] = S.la_sg[r - NA1] * S.Minv[symi(d, r - NA1)];
    }
}
template <typename T, typename Scene = xk::PnpScene>
XARM_HD void lane_delassus_base(const Setup<T> &S, int l, const T (&R)[R_N], T (&nA0)[C0_N], T (&nA1)[C1_N], T (&invd)[4]) {
    const T imb = (T)(1.0 / Scene::OBJ_MASS);
#pragma unroll
    for (int k = 0; k < 4; k++) invd[k] = (T)0;
    T diag0 = (T)1, diag1 = (T)1;
    bool act0 = false;
    // ---- table rows: M^-1 J^T touches the object only
#pragma unroll
    for (int s = 0; s < NTS; s++)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int r = 3 * s + a;
            const V3<T> d = tdir<T>(a);
            const V3<T> bl = d * imb, ba = symmul(S.Iinv, cross(S.tr[s], d));
            const T a0 = R[R_J0 + 0] * bl.x + R[R_J0 + 1] * bl.y + R[R_J0 + 2] * bl.z + R[R_J0 + 3] * ba.x + R[R_J0 + 4] * ba.y + R[R_J0 + 5] * ba.z;
            nA0[C0_T + r] = -a0;
            diag0 = l == r ? a0 : diag0;
            act0 = l == r ? S.tid[s] >= 0 : act0;
        }
    invd[0] = act0 ? (T)1 / diag0 : (T)0;   // one division for the row this lane owns
    // ---- single-joint rows: M^-1 J^T is a signed combination of columns of Minv
#pragma unroll
    for (int r = 0; r < NA1; r++) {
        T B[9];
        arm_row_B(S, r, B);
        T a1 = (T)0;
#pragma unroll
        for (int d = 0; d < 9; d++) a1 += R[R_J1 + d] * B[d];
        nA1[C1_A + r] = -a1;
        diag1 = l == r ? a1 : diag1;
    }
    invd[1] = l < NA1 ? (T)1 / diag1 : (T)0;
}
// the entries that exist only with pad rows (slot 2) and / or arm-limit rows (slot 3)
template <typename T, bool PAD, bool LA, typename Scene = xk::PnpScene>
XARM_HD void lane_delassus_extra(const Setup<T> &S, int l, const T (&R)[R_N], T (&nA1)[C1_N], T (&nA2)[C2_N], T (&nA3)[C1_N], T (&invd)[4]) {
    const T imb = (T)(1.0 / Scene::OBJ_MASS);
    T diag3 = (T)1;
    bool act3 = false;
    if (PAD) {
#pragma unroll
        for (int s = 0; s < NTS; s++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int r = 3 * s + a;
                const V3<T> d = tdir<T>(a);
                const V3<T> bl = d * imb, ba = symmul(S.Iinv, cross(S.tr[s], d));
                nA2[C2_T + r] = -(R[R_J2B + 0] * bl.x + R[R_J2B + 1] * bl.y + R[R_J2B + 2] * bl.z + R[R_J2B + 3] * ba.x + R[R_J2B + 4] * ba.y + R[R_J2B + 5] * ba.z);
            }
    }
#pragma unroll
    for (int r = 0; r < NA1 + NLA; r++) {
        if (!LA && r >= NA1) continue;
        T B[9];
        arm_row_B(S, r, B);
        T a1 = (T)0, a2 = (T)0, a3 = (T)0;
#pragma unroll
        for (int d = 0; d < 9; d++) {
            if (r >= NA1) a1 += R[R_J1 + d] * B[d];
            if (PAD) a2 += R[R_J2A + d] * B[d];
            if (LA && d < 7) a3 += R[R_J3 + d] * B[d];
        }
        const int c1 = r < NA1 ? C1_A + r : C1_L + (r - NA1);
        if (r >= NA1) nA1[c1] = -a1;
        if (PAD) nA2[(r < NA1 ? C2_A + r : C2_L + (r - NA1))] = -a2;
        if (LA) nA3[c1] = -a3;
        if (r >= NA1) { diag3 = l == r - NA1 ? a3 : diag3; act3 = l == r - NA1 ? S.la_sg[r - NA1] != (T)0 : act3; }
    }
    if (LA) invd[3] = act3 ? (T)1 / diag3 : (T)0;
}

// ---------------------------------------------------------------------------------------------
// one solver row, owned by lane LANE in slot SLOT; column indices of the row in each coupled slot
template <typename T> struct Sweep {
    LV<T> g[4], lam[4], invd[4];
    LV<T> lo1, hi1;
    LV<T> nA0[C0_N], nA1[C1_N], nA2[C2_N], nA3[C1_N];
};

// Columns of A that belong to the pad rows.  M^-1 J_r^T of a pad row is expensive (full 9x9 product), so the lane
// that owns row r forms it once for its own row and the row is then handed round by DPP broadcast: every lane dots
// it with the Jacobians of the rows it owns.  padw[p]: pad p has a live row somewhere in the wavefront.
// M^-1 J_r^T of the pad row each lane owns: joint part Ba (9), object part Bb (linear 3, angular 3)
template <typename T, typename Scene = xk::PnpScene>
XARM_HD void pad_minv_jt(const Setup<T> &S, const LV<T> (&J)[R_G], LV<T> (&Ba)[9], LV<T> (&Bb)[6]) {
    const T imb = (T)(1.0 / Scene::OBJ_MASS);
#pragma unroll
    for (int q = 0; q < 9; q++) {
        LV<T> s = lv_scale(J[R_J2A + 0], S.Minv[symi(q, 0)]);
#pragma unroll
        for (int i = 1; i < 9; i++) s = lv_fmas(J[R_J2A + i], S.Minv[symi(q, i)], s);
        Ba[q] = s;
    }
#pragma unroll
    for (int d = 0; d < 3; d++) Bb[d] = lv_scale(J[R_J2B + d], imb);
    Bb[3] = lv_fmas(J[R_J2B + 5], S.Iinv[2], lv_fmas(J[R_J2B + 4], S.Iinv[1], lv_scale(J[R_J2B + 3], S.Iinv[0])));
    Bb[4] = lv_fmas(J[R_J2B + 5], S.Iinv[4], lv_fmas(J[R_J2B + 4], S.Iinv[3], lv_scale(J[R_J2B + 3], S.Iinv[1])));
    Bb[5] = lv_fmas(J[R_J2B + 5], S.Iinv[5], lv_fmas(J[R_J2B + 4], S.Iinv[4], lv_scale(J[R_J2B + 3], S.Iinv[2])));
}
template <typename T, bool LA, typename Scene = xk::PnpScene>
XARM_HD void pad_columns(const Grp &G, const Setup<T> &S, const LV<T> (&J)[R_G], LV<T> cfm, Sweep<T> &W, const bool (&padw)[NP]) {
    LV<T> Ba[9], Bb[6];
    pad_minv_jt<T, Scene>(S, J, Ba, Bb);
#define XC_PAD_COL(r)                                                                                        \
    if (padw[(r) / 3]) {                                                                                     \
        LV<T> ca[9], cb[6];                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 9; q++) ca[q] = lv_bcast<r>(Ba[q]);                            \
        _Pragma("unroll") for (int d = 0; d < 6; d++) cb[d] = lv_bcast<r>(Bb[d]);                            \
        LV<T> a0 = lv_mul(J[R_J0], cb[0]), a1 = lv_mul(J[R_J1], ca[0]), a2 = lv_mul(J[R_J2A], ca[0]);        \
        _Pragma("unroll") for (int d = 1; d < 6; d++) a0 = lv_fma(J[R_J0 + d], cb[d], a0);                   \
        _Pragma("unroll") for (int q = 1; q < 9; q++) { a1 = lv_fma(J[R_J1 + q], ca[q], a1); a2 = lv_fma(J[R_J2A + q], ca[q], a2); } \
        _Pragma("unroll") for (int d = 0; d < 6; d++) a2 = lv_fma(J[R_J2B + d], cb[d], a2);                  \
        LV<T> diag = a2;                                                                                     \
        lv_commit<r>(G, diag, lv_add(a2, cfm)); /* the owner of a pad normal adds its cfm: g_r carries -cfm lam_r */ \
        W.nA0[C0_F + r] = lv_neg(a0);                                                                        \
        W.nA1[C1_F + r] = lv_neg(a1);                                                                        \
        W.nA2[C2_F + r] = lv_neg(diag);                                                                      \
        if (LA) {                                                                                            \
            LV<T> a3 = lv_mul(J[R_J3], ca[0]);                                                               \
            _Pragma("unroll") for (int q = 1; q < 7; q++) a3 = lv_fma(J[R_J3 + q], ca[q], a3);               \
            W.nA3[C1_F + r] = lv_neg(a3);                                                                    \
        }                                                                                                    \
        lv_commit<r>(G, W.invd[2], lv_rcp_if(diag, S.pact[(r) / 3]));                                        \
    }
    XC_PAD_COL(0) XC_PAD_COL(1) XC_PAD_COL(2) XC_PAD_COL(3) XC_PAD_COL(4) XC_PAD_COL(5)
    XC_PAD_COL(6) XC_PAD_COL(7) XC_PAD_COL(8) XC_PAD_COL(9) XC_PAD_COL(10) XC_PAD_COL(11)
#undef XC_PAD_COL
}

// The NUM_ITERATIONS sweeps.  Row order of the oracle: T (table points), M (motors), L (arm limits, finger limits),
// G (gear), F (pad points).  The table rows never couple to the single-joint rows (A is block diagonal there; only
// the pad rows touch both), so table row i and single-joint row i - both owned by lane i - are advanced in one PAIR
// step on packed registers: same arithmetic per row, same order within each block, 14 steps instead of 26.
#if defined(XC_SWEEP_ITERS) && !defined(XARM_SWEEP_VARIANT)
#define XARM_SWEEP_VARIANT XC_SWEEP_ITERS      // xarm_version() reports it (xarm_hip.hip)
#endif
#ifndef XC_SWEEP_ITERS
#define XC_SWEEP_ITERS xm::NUM_ITERATIONS   // timing probes only (tools/coop_split.sh) build with fewer sweeps
#endif
// The loop carries c = lam + g / d per row (the unclamped impulse the row would take now) instead of the residual g:
// a row step is then clamp(c) -> d lam -> broadcast -> c += (A / d) d lam for every other row, FOUR dependent
// instructions where the residual form needs five (it re-derives lam + g / d first) - and the sweep is bound by exactly
// that chain (tools/probes/dpp_probe.hip).  The columns are scaled by the receiving row's 1 / d once per substep; a
// row's own entry becomes zero, because its c does not move when its own impulse does (lam + dl + (g - d dl) / d).
template <typename T, bool PAD, bool LA>
XARM_HD void sweep_all(const Grp &G, Sweep<T> &W, T mu_t, T mu_p, const bool (&padw)[NP]) {
    LV2<T> lam01 = lv2_make(W.lam[0], W.lam[1]);
    const LV2<T> invd01 = lv2_make(W.invd[0], W.invd[1]), zero2 = lv2_make(lv_fill((T)0), lv_fill((T)0));
    LV2<T> c01 = lv2_fma(lv2_make(W.g[0], W.g[1]), invd01, lam01);
    LV2<T> A01[NA1], AF01[NF];
#define XC_COL01(i)                                                                                          \
    A01[i] = lv2_mul(lv2_make((i) < NT ? W.nA0[C0_T + (i)] : lv_fill((T)0), W.nA1[C1_A + (i)]), invd01);    \
    lv2_commit<i>(G, A01[i], zero2);
    XC_COL01(0) XC_COL01(1) XC_COL01(2) XC_COL01(3) XC_COL01(4) XC_COL01(5) XC_COL01(6) XC_COL01(7) XC_COL01(8) XC_COL01(9)
    XC_COL01(10) XC_COL01(11) XC_COL01(12) XC_COL01(13)
#undef XC_COL01
    LV<T> AL1[7];   // arm-limit columns of the single-joint rows
    if (LA) {
#pragma unroll
        for (int i = 0; i < 7; i++) AL1[i] = lv_mul(W.nA1[C1_L + i], W.invd[1]);
    }
    LV<T> c2 = lv_fill((T)0), c3 = lv_fill((T)0);
    if (PAD) {
        c2 = lv_fma(W.g[2], W.invd[2], W.lam[2]);
#pragma unroll
        for (int i = 0; i < NT; i++) W.nA2[C2_T + i] = lv_mul(W.nA2[C2_T + i], W.invd[2]);
#pragma unroll
        for (int i = 0; i < NA1; i++) W.nA2[C2_A + i] = lv_mul(W.nA2[C2_A + i], W.invd[2]);
        if (LA) {
#pragma unroll
            for (int i = 0; i < 7; i++) W.nA2[C2_L + i] = lv_mul(W.nA2[C2_L + i], W.invd[2]);
        }
#define XC_COLF(r)                                                                                           \
        if (padw[(r) / 3]) {                                                                                 \
            AF01[r] = lv2_mul(lv2_make(W.nA0[C0_F + r], W.nA1[C1_F + r]), invd01);                           \
            W.nA2[C2_F + r] = lv_mul(W.nA2[C2_F + r], W.invd[2]);                                            \
            lv_commit<r>(G, W.nA2[C2_F + r], lv_fill((T)0));                                                 \
            if (LA) W.nA3[C1_F + r] = lv_mul(W.nA3[C1_F + r], W.invd[3]);                                    \
        }
        XC_COLF(0) XC_COLF(1) XC_COLF(2) XC_COLF(3) XC_COLF(4) XC_COLF(5) XC_COLF(6) XC_COLF(7) XC_COLF(8) XC_COLF(9) XC_COLF(10) XC_COLF(11)
#undef XC_COLF
    }
    if (LA) {
        c3 = lv_fma(W.g[3], W.invd[3], W.lam[3]);
#pragma unroll
        for (int i = 0; i < NA1; i++) W.nA3[C1_A + i] = lv_mul(W.nA3[C1_A + i], W.invd[3]);
#define XC_COLL(i)                                                                                           \
        W.nA3[C1_L + i] = lv_mul(W.nA3[C1_L + i], W.invd[3]);                                                \
        lv_commit<i>(G, W.nA3[C1_L + i], lv_fill((T)0));
        XC_COLL(0) XC_COLL(1) XC_COLL(2) XC_COLL(3) XC_COLL(4) XC_COLL(5) XC_COLL(6)
#undef XC_COLL
    }
    const LV<T> mu_tv = lv_fill_vgpr(mu_t), mu_pv = lv_fill(mu_p);
    // NL[i]: the impulse pair pair step i last produced.  Only lane i's entry means anything (the other lanes clamp their own c),
    // and only lane i's entry is ever used: its difference to the new pair is what step i broadcasts.  Keeping the 14 pairs
    // in registers replaces the per-step write of lane i into ONE pair (two v_cndmask_b32 per step: 8 of the 48 cycles of a pair
    // step, tools/probes/commit_probe.hip) by 14 writes after the last sweep - same values, same arithmetic.
    LV2<T> NL[NA1];
#pragma unroll
    for (int i = 0; i < NA1; i++) NL[i] = lam01;
    // which arm-limit rows are live somewhere in the wavefront (usually one joint near one limit): the others are
    // exact no-ops and are skipped, the launch lasts as long as its slowest wavefront
    bool law[NLA];
#define XC_LAW(i) law[i] = LA && XARM_ANY_X(lv_get<i>(W.invd[3]) != (T)0);
    XC_LAW(0) XC_LAW(1) XC_LAW(2) XC_LAW(3) XC_LAW(4) XC_LAW(5) XC_LAW(6)
#undef XC_LAW
#pragma unroll 1
    for (int it = 0; it < XC_SWEEP_ITERS; it++) {
        LV<T> lim = lv_fill((T)0);
        // pair step i: table row i (normal of point i/3 when i % 3 == 0, else friction; none for i >= 12) + slot-1 row i
#define XC_PAIR(i)                                                                                           \
        {                                                                                                    \
            LV<T> nx = lv2_x(c01);                                                                           \
            if ((i) % 3 == 0 || (i) >= NT) nx = lv_max0(nx);                                                 \
            else nx = lv_med3(nx, lv_neg(lim), lim);                                                         \
            const LV2<T> nl = lv2_make(nx, lv_med3(lv2_y(c01), W.lo1, W.hi1));                               \
            const LV2<T> dl = lv2_sub(nl, NL[i]);                                                            \
            NL[i] = nl; /* no commit inside the loop: see NL above */                                        \
            if ((i) % 3 == 0 && (i) < NT) lim = lv_mul(lv_bcast<i>(nx), mu_tv); /* friction limit of this point */ \
            const LV2<T> b = lv2_bcast<i>(dl);                                                               \
            c01 = lv2_fma(A01[i], b, c01);                                                                   \
            if (PAD) {                                                                                       \
                if ((i) < NT) c2 = lv_fma(W.nA2[C2_T + (i)], lv2_x(b), c2);                                  \
                c2 = lv_fma(W.nA2[C2_A + (i)], lv2_y(b), c2);                                                \
            }                                                                                                \
            if (LA) c3 = lv_fma(W.nA3[C1_A + (i)], lv2_y(b), c3);                                            \
        }
#define XC_L_ROW(i)                                                                                          \
        if (law[i]) {                                                                                        \
            const LV<T> nl = lv_max0(c3);                                                                    \
            const LV<T> dl = lv_sub(nl, W.lam[3]);                                                           \
            lv_commit<i>(G, W.lam[3], nl);                                                                   \
            const LV<T> b = lv_bcast<i>(dl);                                                                 \
            c01 = lv2_make(lv2_x(c01), lv_fma(AL1[i], b, lv2_y(c01)));                                       \
            if (PAD) c2 = lv_fma(W.nA2[C2_L + i], b, c2);                                                    \
            c3 = lv_fma(W.nA3[C1_L + i], b, c3);                                                             \
        }
#define XC_F_ROW(p, a)                                                                                       \
        {                                                                                                    \
            LV<T> nl;                                                                                        \
            if ((a) == 0) nl = lv_max0(c2);                                                                  \
            else {                                                                                           \
                const LV<T> flim = lv_mul(lv_bcast<3 * p>(W.lam[2]), mu_pv);                                 \
                nl = lv_med3(c2, lv_neg(flim), flim);                                                        \
            }                                                                                                \
            const LV<T> dl = lv_sub(nl, W.lam[2]);                                                           \
            lv_commit<3 * p + a>(G, W.lam[2], nl);                                                           \
            const LV<T> b = lv_bcast<3 * p + a>(dl);                                                         \
            c01 = lv2_fma(AF01[3 * p + a], lv2_make(b, b), c01);                                             \
            c2 = lv_fma(W.nA2[C2_F + 3 * p + a], b, c2);                                                     \
            if (LA) c3 = lv_fma(W.nA3[C1_F + 3 * p + a], b, c3);                                             \
        }
#define XC_F_PAD(p) if (padw[p]) { XC_F_ROW(p, 0) XC_F_ROW(p, 1) XC_F_ROW(p, 2) }
        XC_PAIR(0) XC_PAIR(1) XC_PAIR(2) XC_PAIR(3) XC_PAIR(4) XC_PAIR(5) XC_PAIR(6) XC_PAIR(7) XC_PAIR(8)
        if (LA) { XC_L_ROW(0) XC_L_ROW(1) XC_L_ROW(2) XC_L_ROW(3) XC_L_ROW(4) XC_L_ROW(5) XC_L_ROW(6) }
        XC_PAIR(9) XC_PAIR(10) XC_PAIR(11) XC_PAIR(12) XC_PAIR(13)
        if (PAD) { XC_F_PAD(0) XC_F_PAD(1) XC_F_PAD(2) XC_F_PAD(3) }
#undef XC_PAIR
#undef XC_L_ROW
#undef XC_F_ROW
#undef XC_F_PAD
    }
#define XC_KEEP(i) lv2_commit<i>(G, lam01, NL[i]);
    XC_KEEP(0) XC_KEEP(1) XC_KEEP(2) XC_KEEP(3) XC_KEEP(4) XC_KEEP(5) XC_KEEP(6) XC_KEEP(7) XC_KEEP(8) XC_KEEP(9)
    XC_KEEP(10) XC_KEEP(11) XC_KEEP(12) XC_KEEP(13)
#undef XC_KEEP
    W.lam[0] = lv2_x(lam01);
    W.lam[1] = lv2_y(lam01);
}

// warm start: the impulses the rows start with have already acted on the velocities, g -= A lam0
template <typename T, bool PAD, bool LA> XARM_HD void apply_warm_start(Sweep<T> &W, const bool (&padw)[NP]) {
#define XC_WS_T(s)                                                                                           \
    {                                                                                                        \
        const LV<T> b = lv_bcast<3 * s>(W.lam[0]);                                                           \
        W.g[0] = lv_fma(W.nA0[C0_T + 3 * s], b, W.g[0]);                                                    \
        if (PAD) W.g[2] = lv_fma(W.nA2[C2_T + 3 * s], b, W.g[2]);                                           \
    }
    XC_WS_T(0) XC_WS_T(1) XC_WS_T(2) XC_WS_T(3)
    if (PAD) {
#define XC_WS_F(p)                                                                                           \
    if (padw[p]) {                                                                                                     \
        const LV<T> b = lv_bcast<3 * p>(W.lam[2]);                                                           \
        W.g[0] = lv_fma(W.nA0[C0_F + 3 * p], b, W.g[0]);                                                    \
        W.g[1] = lv_fma(W.nA1[C1_F + 3 * p], b, W.g[1]);                                                    \
        W.g[2] = lv_fma(W.nA2[C2_F + 3 * p], b, W.g[2]);                                                    \
        if (LA) W.g[3] = lv_fma(W.nA3[C1_F + 3 * p], b, W.g[3]);                                            \
    }
        XC_WS_F(0) XC_WS_F(1) XC_WS_F(2) XC_WS_F(3)
    }
#undef XC_WS_T
#undef XC_WS_F
}

// ---------------------------------------------------------------------------------------------
// Arm dynamics with the per-body work dealt to the lanes of the row (lane b = body b: links 1-7, the two fingers).
// xk::arm_dynamics spends most of its instructions on nine rigid-body inertias about the world origin, their bias
// forces and the composite-rigid-body products; here the kinematic chain (frames, velocities, accelerations: a
// serial recursion) is still walked by every lane, each lane captures the frame of ITS body on the way, forms that
// body's inertia and bias force, a DPP suffix scan over the lanes turns them into the composites of the subtree, and
// lane j computes row j of the joint-space inertia.  The rows are then broadcast and the 9x9 factorisation runs
// redundantly as before.  Same quantities as xk::arm_dynamics (summation order of the composites differs).
template <typename T> struct ArmLane { LV<T> mass, com[3], inertia[6], damping; };   // per-lane body constants
template <typename T> XARM_HD ArmLane<T> arm_lane_consts(const Grp &G) {
    ArmLane<T> C;
    XC_LANES {
        const int l = lane_of(G, i_);
        T m = (T)0, dmp = (T)0, c[3] = {(T)0, (T)0, (T)0}, in[6] = {(T)0, (T)0, (T)0, (T)0, (T)0, (T)0};
#pragma unroll
        for (int b = 0; b < 9; b++) {
            m = l == b ? (T)xm::MASS[b] : m;
            dmp = l == b ? (T)xm::DAMPING[b] : dmp;
#pragma unroll
            for (int k = 0; k < 3; k++) c[k] = l == b ? (T)xm::COM[b][k] : c[k];
#pragma unroll
            for (int k = 0; k < 6; k++) in[k] = l == b ? (T)xm::INERTIA[b][k] : in[k];
        }
        C.mass.v[i_] = m; C.damping.v[i_] = dmp;
#pragma unroll
        for (int k = 0; k < 3; k++) C.com[k].v[i_] = c[k];
#pragma unroll
        for (int k = 0; k < 6; k++) C.inertia[k].v[i_] = in[k];
    }
    return C;
}

template <typename T, typename Lds, typename Scene = xk::PnpScene>
XARM_HD void arm_dynamics_coop(const Grp &G, const ArmLane<T> &C, const T (&q_in)[9], const T (&qd_in)[9], const T dt, Lds lds, xk::ArmDyn<T> &A, const int arm = 0) {
    using xk::SV; using xk::RBI; using xk::Frame;
    // ---- the chain, walked by every lane; lane i captures body i's frame / spatial velocity / acceleration / joint rate
    SV<T> S[7];
    Frame<T> f = Scene::template base_frame<T>(arm);
    SV<T> vel, acc;
    vel.w = mk<T>(0, 0, 0); vel.v = mk<T>(0, 0, 0);
    acc.w = mk<T>(0, 0, 0); acc.v = mk<T>(0, 0, (T)xm::GRAVITY);
    LV<T> cap[24], capqd = lv_fill((T)0);
#pragma unroll
    for (int k = 0; k < 24; k++) cap[k] = lv_fill((T)0);
#define XC_CAPTURE(i, fr, vv, aa, qdv)                                                                       \
    {                                                                                                        \
        const T vals[24] = {fr.c0.x, fr.c0.y, fr.c0.z, fr.c1.x, fr.c1.y, fr.c1.z, fr.c2.x, fr.c2.y, fr.c2.z, fr.o.x, fr.o.y, fr.o.z, \
                            vv.w.x, vv.w.y, vv.w.z, vv.v.x, vv.v.y, vv.v.z, aa.w.x, aa.w.y, aa.w.z, aa.v.x, aa.v.y, aa.v.z}; \
        _Pragma("unroll") for (int k = 0; k < 24; k++) lv_commit<i>(G, cap[k], lv_fill(vals[k]));            \
        lv_commit<i>(G, capqd, lv_fill(qdv));                                                                \
    }
#define XC_CHAIN(i)                                                                                          \
    {                                                                                                        \
        xk::fk_advance(f, i, q_in[i]);                                                                       \
        S[i].w = f.c2;                                                                                       \
        S[i].v = cross(f.o, f.c2);                                                                           \
        const T qd = qd_in[i];                                                                               \
        acc.w = acc.w + cross(vel.w, S[i].w) * qd;                                                           \
        acc.v = acc.v + (cross(vel.w, S[i].v) + cross(vel.v, S[i].w)) * qd;                                  \
        vel.w = vel.w + S[i].w * qd;                                                                         \
        vel.v = vel.v + S[i].v * qd;                                                                         \
        XC_CAPTURE(i, f, vel, acc, qd)                                                                       \
    }
    XC_CHAIN(0) XC_CHAIN(1) XC_CHAIN(2) XC_CHAIN(3) XC_CHAIN(4) XC_CHAIN(5) XC_CHAIN(6)
#undef XC_CHAIN
    const V3<T> hc0 = f.c0, hc1 = f.c1, hc2 = f.c2, ho = f.o;
    A.hc0 = hc0; A.hc1 = hc1; A.hc2 = hc2;
    // fingers slide along +/- hand y: frame = hand axes at the finger origin, prismatic velocity / acceleration terms
#define XC_FINGER(k)                                                                                         \
    {                                                                                                        \
        const T sg = k == 0 ? (T)1 : (T)-1;                                                                  \
        const V3<T> af = hc1 * sg;                                                                           \
        A.fo[k] = ho + hc2 * (T)xm::FINGER_Z + af * q_in[7 + k];                                             \
        const T qd = qd_in[7 + k];                                                                           \
        SV<T> v = vel, a = acc;                                                                              \
        a.v = a.v + cross(vel.w, af) * qd;                                                                   \
        v.v = v.v + af * qd;                                                                                 \
        Frame<T> ff; ff.c0 = hc0; ff.c1 = hc1; ff.c2 = hc2; ff.o = A.fo[k];                                  \
        XC_CAPTURE(7 + k, ff, v, a, qd)                                                                      \
    }
    XC_FINGER(0) XC_FINGER(1)
#undef XC_FINGER
#undef XC_CAPTURE
    // ---- per lane: inertia of the lane's body about the world origin, bias force, then composites by suffix scan
    LV<T> body[16], comp[16], Mrow[9], taul;
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
        const SV<T> Iv = xk::rbi_mul(I, v), Ia = xk::rbi_mul(I, a);
        SV<T> fbias;
        fbias.w = Ia.w + cross(v.w, Iv.w) + cross(v.v, Iv.v);
        fbias.v = Ia.v + cross(v.w, Iv.v);
        const T vals[16] = {I.m, I.h.x, I.h.y, I.h.z, I.I[0], I.I[1], I.I[2], I.I[3], I.I[4], I.I[5],
                            fbias.w.x, fbias.w.y, fbias.w.z, fbias.v.x, fbias.v.y, fbias.v.z};
#pragma unroll
        for (int k = 0; k < 16; k++) body[k].v[i_] = l < 9 ? vals[k] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) comp[k] = lv_suffix_sum(body[k]);
    // ---- per lane: row l of the joint-space inertia and the bias torque (arm lanes: composite of the subtree, revolute
    // axis; finger lanes: their own body, prismatic axis along +/- hand y)
    XC_LANES {
        const int l = lane_of(G, i_);
        const bool fin = l >= 7;
        RBI<T> Ic;
        SV<T> fc, Sl;
        const LV<T> *src = comp;
        Ic.m = fin ? body[0].v[i_] : src[0].v[i_];
        Ic.h = mk<T>(fin ? body[1].v[i_] : src[1].v[i_], fin ? body[2].v[i_] : src[2].v[i_], fin ? body[3].v[i_] : src[3].v[i_]);
#pragma unroll
        for (int k = 0; k < 6; k++) Ic.I[k] = fin ? body[4 + k].v[i_] : src[4 + k].v[i_];
        fc.w = mk<T>(fin ? body[10].v[i_] : src[10].v[i_], fin ? body[11].v[i_] : src[11].v[i_], fin ? body[12].v[i_] : src[12].v[i_]);
        fc.v = mk<T>(fin ? body[13].v[i_] : src[13].v[i_], fin ? body[14].v[i_] : src[14].v[i_], fin ? body[15].v[i_] : src[15].v[i_]);
        const V3<T> c1 = mk<T>(cap[3].v[i_], cap[4].v[i_], cap[5].v[i_]), c2 = mk<T>(cap[6].v[i_], cap[7].v[i_], cap[8].v[i_]),
                    o = mk<T>(cap[9].v[i_], cap[10].v[i_], cap[11].v[i_]);
        const V3<T> ax = c1 * (l == 8 ? (T)-1 : (T)1);                          // finger slide direction
        Sl.w = fin ? mk<T>(0, 0, 0) : c2;
        Sl.v = fin ? ax : cross(o, c2);
        const SV<T> F = xk::rbi_mul(Ic, Sl);
#pragma unroll
        for (int i = 0; i < 7; i++) Mrow[i].v[i_] = xk::sdot(S[i], F);
        Mrow[7].v[i_] = l == 7 ? dot(ax, F.v) : (T)0;                           // finger 2 is not carried by finger 1: M[8][7] = 0
        Mrow[8].v[i_] = l == 8 ? dot(ax, F.v) : (T)0;
        taul.v[i_] = -xk::sdot(Sl, fc) - C.damping.v[i_] * capqd.v[i_];
    }
    // ---- rows and torques to every lane, then the factorisation as in xk::arm_dynamics
    T M[45], tau[9];
#define XC_ROW(r)                                                                                            \
    {                                                                                                        \
        _Pragma("unroll") for (int c = 0; c <= r; c++) M[tri(r, c)] = lv_get<r>(Mrow[c]);                    \
        tau[r] = lv_get<r>(taul);                                                                            \
    }
    XC_ROW(0) XC_ROW(1) XC_ROW(2) XC_ROW(3) XC_ROW(4) XC_ROW(5) XC_ROW(6) XC_ROW(7) XC_ROW(8)
#undef XC_ROW
    T (&Minv)[45] = A.Minv;
    {
        T rd[9];
#pragma unroll
        for (int c = 0; c < 9; c++) {
#pragma unroll
            for (int r = c; r < 9; r++) {
                T s = M[tri(r, c)];
#pragma unroll
                for (int k = 0; k < c; k++) s -= M[tri(r, k)] * M[tri(c, k)];
                if (r == c) { M[tri(c, c)] = xsqrt(s); rd[c] = (T)1 / M[tri(c, c)]; }
                else M[tri(r, c)] = s * rd[c];
            }
        }
        T Li[45];
#pragma unroll
        for (int c = 0; c < 9; c++) {
            Li[tri(c, c)] = rd[c];
#pragma unroll
            for (int r = c + 1; r < 9; r++) {
                T s = (T)0;
#pragma unroll
                for (int k = c; k < r; k++) s -= M[tri(r, k)] * Li[tri(k, c)];
                Li[tri(r, c)] = s * rd[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 9; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) {
                T s = (T)0;
#pragma unroll
                for (int k = r; k < 9; k++) s += Li[tri(k, r)] * Li[tri(k, c)];
                Minv[tri(r, c)] = s;
            }
    }
#pragma unroll
    for (int r = 0; r < 9; r++) {
        T s = (T)0;
#pragma unroll
        for (int c = 0; c < 9; c++) s += Minv[symi(r, c)] * tau[c];
        A.dq[r] = qd_in[r] + dt * s;
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        lds[LDS_S + i * 6 + 0] = S[i].w.x; lds[LDS_S + i * 6 + 1] = S[i].w.y; lds[LDS_S + i * 6 + 2] = S[i].w.z;
        lds[LDS_S + i * 6 + 3] = S[i].v.x; lds[LDS_S + i * 6 + 4] = S[i].v.y; lds[LDS_S + i * 6 + 5] = S[i].v.z;
    }
    XARM_LDS_FENCE();
}

// ---------------------------------------------------------------------------------------------
// collision + row constants of a substep: same arithmetic as the first half of xk::substep, minus the
// operational-space K blocks (the Delassus rows replace them)
template <typename T, typename Lds, typename Scene = xk::PnpScene>
XARM_HD void substep_setup(const Grp &G, const ArmLane<T> &C, EnvState<T> &st, const T (&qt)[9], const T dt, Lds lds, Setup<T> &S, const int arm = 0) {
    const T idt = (T)1 / dt;
    xk::ArmDyn<T> AD;
    arm_dynamics_coop<T, Lds, Scene>(G, C, st.q, st.qd, dt, lds, AD, arm);
#pragma unroll
    for (int k = 0; k < 45; k++) S.Minv[k] = AD.Minv[k];
#pragma unroll
    for (int k = 0; k < 9; k++) S.dq[k] = AD.dq[k];
    S.hc1 = AD.hc1;
    const V3<T> hc0 = AD.hc0, hc1 = AD.hc1, hc2 = AD.hc2;
    // object frame, inverse inertia, unconstrained motion (gyroscopic step, gravity, damping)
    V3<T> b0, b1, b2;
    {
        const T x = st.bq[0], y = st.bq[1], z = st.bq[2], w = st.bq[3];
        b0 = mk<T>((T)1 - (T)2 * (y * y + z * z), (T)2 * (x * y + z * w), (T)2 * (x * z - y * w));
        b1 = mk<T>((T)2 * (x * y - z * w), (T)1 - (T)2 * (x * x + z * z), (T)2 * (y * z + x * w));
        b2 = mk<T>((T)2 * (x * z + y * w), (T)2 * (y * z - x * w), (T)1 - (T)2 * (x * x + y * y));
    }
    const V3<T> cb = mk<T>(st.bp[0], st.bp[1], st.bp[2]);
    S.cb = cb;
    const T hx = (T)Scene::OBJ_HX, hy = (T)Scene::OBJ_HY, hz = (T)Scene::OBJ_HZ;
    const T Ibx = (T)(Scene::OBJ_MASS / 3.0 * (Scene::OBJ_HY * Scene::OBJ_HY + Scene::OBJ_HZ * Scene::OBJ_HZ));
    const T Iby = (T)(Scene::OBJ_MASS / 3.0 * (Scene::OBJ_HX * Scene::OBJ_HX + Scene::OBJ_HZ * Scene::OBJ_HZ));
    const T Ibz = (T)(Scene::OBJ_MASS / 3.0 * (Scene::OBJ_HX * Scene::OBJ_HX + Scene::OBJ_HY * Scene::OBJ_HY));
    {
        const T ix = (T)1 / Ibx, iy = (T)1 / Iby, iz = (T)1 / Ibz;
        S.Iinv[0] = b0.x * b0.x * ix + b1.x * b1.x * iy + b2.x * b2.x * iz;
        S.Iinv[1] = b0.x * b0.y * ix + b1.x * b1.y * iy + b2.x * b2.y * iz;
        S.Iinv[2] = b0.x * b0.z * ix + b1.x * b1.z * iy + b2.x * b2.z * iz;
        S.Iinv[3] = b0.y * b0.y * ix + b1.y * b1.y * iy + b2.y * b2.y * iz;
        S.Iinv[4] = b0.y * b0.z * ix + b1.y * b1.z * iy + b2.y * b2.z * iz;
        S.Iinv[5] = b0.z * b0.z * ix + b1.z * b1.z * iy + b2.z * b2.z * iz;
    }
    V3<T> vb = mk<T>(st.bv[0], st.bv[1], st.bv[2]), wb = mk<T>(st.bw[0], st.bw[1], st.bw[2]);
    {
        const V3<T> wl = mk<T>(dot(b0, wb), dot(b1, wb), dot(b2, wb));
        const V3<T> iw = mk<T>(Ibx * wl.x, Iby * wl.y, Ibz * wl.z);
        const V3<T> f = cross(wl, iw) * dt;
        const T J00 = Ibx, J01 = dt * (-wl.z * Iby + iw.z), J02 = dt * (wl.y * Ibz - iw.y);
        const T J10 = dt * (wl.z * Ibx - iw.z), J11 = Iby, J12 = dt * (-wl.x * Ibz + iw.x);
        const T J20 = dt * (-wl.y * Ibx + iw.y), J21 = dt * (wl.x * Iby - iw.x), J22 = Ibz;
        const T c00 = J11 * J22 - J12 * J21, c01 = J12 * J20 - J10 * J22, c02 = J10 * J21 - J11 * J20;
        const T id = (T)1 / (J00 * c00 + J01 * c01 + J02 * c02);
        const V3<T> x = mk<T>((f.x * c00 + f.y * (J02 * J21 - J01 * J22) + f.z * (J01 * J12 - J02 * J11)) * id,
                              (f.x * c01 + f.y * (J00 * J22 - J02 * J20) + f.z * (J02 * J10 - J00 * J12)) * id,
                              (f.x * c02 + f.y * (J01 * J20 - J00 * J21) + f.z * (J00 * J11 - J01 * J10)) * id);
        const V3<T> wn = wl - x;
        wb = b0 * wn.x + b1 * wn.y + b2 * wn.z;
        vb.z -= dt * (T)xm::GRAVITY;
        vb = vb * (T)Scene::LIN_DAMP_FACTOR;
        wb = wb * (T)Scene::ANG_DAMP_FACTOR;
    }
    S.vb = vb; S.wb = wb;
    // (T) object corners against the table: first <= NTS active corners
#pragma unroll
    for (int s = 0; s < NTS; s++) { S.tr[s] = mk<T>(0, 0, 0); S.tvt[s] = (T)0; S.tl0[s] = (T)0; S.tid[s] = -1; }
    {
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const V3<T> r = b0 * ((i & 1) ? hx : -hx) + b1 * ((i & 2) ? hy : -hy) + b2 * ((i & 4) ? hz : -hz);
            const V3<T> p = cb + r;
            T hsup;
            const bool sup = Scene::template support<T>(p, hsup);
            const T dist = p.z - hsup;
            const bool act = dist < (T)xm::SOLVER_MARGIN && sup && cnt < NTS;
            const T vt = dist < (T)0 ? -(T)xm::CONTACT_ERP * dist * idt : -dist * idt;
            const T l0 = (T)xm::WARMSTART * st.lam_t[i];
#pragma unroll
            for (int s = 0; s < NTS; s++) {
                const bool put = act && cnt == s;
                S.tr[s] = xk::selv(put, r, S.tr[s]);
                S.tvt[s] = put ? vt : S.tvt[s];
                S.tl0[s] = put ? l0 : S.tl0[s];
                S.tid[s] = put ? i : S.tid[s];
            }
            cnt += act ? 1 : 0;
        }
        if constexpr (Scene::HAS_STAND) {
            // the static stand under the goal: up to four more support points (ids 8..11, no warm start) after the corners
            V3<T> sp[4];
            T sd[4];
            Scene::template stand_points<T>(st.goal, cb, b0, b1, b2, sp, sd);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const T dist = sd[v];
                const bool act = dist < (T)xm::SOLVER_MARGIN && dist > (T)Scene::STAND_MIN_GAP && cnt < NTS;
                const V3<T> r = sp[v] - cb;
                const T vt = dist < (T)0 ? -(T)xm::CONTACT_ERP * dist * idt : -dist * idt;
#pragma unroll
                for (int s = 0; s < NTS; s++) {
                    const bool put = act && cnt == s;
                    S.tr[s] = xk::selv(put, r, S.tr[s]);
                    S.tvt[s] = put ? vt : S.tvt[s];
                    S.tl0[s] = put ? (T)0 : S.tl0[s];
                    S.tid[s] = put ? 8 + v : S.tid[s];
                }
                cnt += act ? 1 : 0;
            }
        }
    }
    // (M) motors, (L) limits, (G) gear: targets
#pragma unroll
    for (int i = 0; i < 9; i++) S.m_vt[i] = (T)xm::MOTOR_KP * (qt[i] - st.q[i]) * idt + (T)(1.0 - xm::MOTOR_KD) * S.dq[i];
    bool la_any = false;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const T g0 = st.q[i] - (T)xm::LOWER[i], g1 = (T)xm::UPPER[i] - st.q[i];
        const bool lo = g0 < (T)xm::LIMIT_WINDOW, hi = g1 < (T)xm::LIMIT_WINDOW;
        const T g = lo ? g0 : g1;
        S.la_sg[i] = lo ? (T)1 : (hi ? (T)-1 : (T)0);
        S.la_vt[i] = g < (T)0 ? -(T)xm::GLOBAL_ERP * g * idt : -g * idt;
        la_any = la_any || S.la_sg[i] != (T)0;
    }
    S.la_any = la_any;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const T g0 = st.q[7 + k] - (T)xm::LOWER[7 + k], g1 = (T)xm::UPPER[7 + k] - st.q[7 + k];
        S.lf_vt[k][0] = g0 < (T)0 ? -(T)xm::GLOBAL_ERP * g0 * idt : -g0 * idt;
        S.lf_vt[k][1] = g1 < (T)0 ? -(T)xm::GLOBAL_ERP * g1 * idt : -g1 * idt;
    }
    S.g_vt = -(T)(xm::GEAR_ERP * xm::GLOBAL_ERP) * (st.q[7] - st.q[8]) * idt;
    // (F) finger pad spheres against the object
    const T pad_denom = dt * (T)xm::FINGER_CONTACT_STIFFNESS + (T)(xm::FINGER_CONTACT_DAMPING + xm::OBJECT_CONTACT_DAMPING);
    const T pad_erp = dt * (T)xm::FINGER_CONTACT_STIFFNESS / pad_denom;
    S.pad_cfm = ((T)1 / pad_denom) * idt;
    // DEALT: each lane tests the ONE pad whose rows it owns (lane l: pad l / 3; lanes 12-15 repeat pad 3) instead of all four in
    // every lane, and keeps the point for its own row; only the flags travel (four row broadcasts each)
    static_assert(xm::NPAD == 2 && NP == 4, "pad dealing below");
    LV<T> actf, tchf;
    XC_LANES {
        const int l = lane_of(G, i_);
        const int idx = l / 3 < NP ? l / 3 : NP - 1;
        const int fk = idx >> 1, j = idx & 1;
        const T sg = fk == 0 ? (T)1 : (T)-1;
        const T pc0 = j == 0 ? (T)xm::PAD_C[0][0] : (T)xm::PAD_C[1][0], pc1 = j == 0 ? (T)xm::PAD_C[0][1] : (T)xm::PAD_C[1][1];
        const T pc2 = j == 0 ? (T)xm::PAD_C[0][2] : (T)xm::PAD_C[1][2];
        const V3<T> c = xk::selv(fk == 0, AD.fo[0], AD.fo[1]) + hc0 * pc0 + hc1 * (sg * pc1) + hc2 * pc2;
        const V3<T> d = c - cb;
        const V3<T> cl = mk<T>(dot(b0, d), dot(b1, d), dot(b2, d));
        const V3<T> ql = mk<T>(clampT(cl.x, -hx, hx), clampT(cl.y, -hy, hy), clampT(cl.z, -hz, hz));
        const V3<T> dl = cl - ql;
        const T d2 = dot(dl, dl);
        V3<T> nl, pl;
        T dist;
        if (d2 > (T)1e-12) {
            const T len = xsqrt(d2);
            nl = dl * ((T)1 / len);
            dist = len - (T)xm::PAD_RADIUS;
            pl = ql;
        } else {
            const T px = hx - xabs(cl.x), py = hy - xabs(cl.y), pz = hz - xabs(cl.z);
            int k = 0;
            T best = px;
            if (py < best) { best = py; k = 1; }
            if (pz < best) { best = pz; k = 2; }
            const T clk = k == 0 ? cl.x : (k == 1 ? cl.y : cl.z);
            const T s1 = clk < (T)0 ? (T)-1 : (T)1;
            nl = mk<T>(k == 0 ? s1 : (T)0, k == 1 ? s1 : (T)0, k == 2 ? s1 : (T)0);
            dist = -best - (T)xm::PAD_RADIUS;
            pl = mk<T>(k == 0 ? s1 * hx : cl.x, k == 1 ? s1 * hy : cl.y, k == 2 ? s1 * hz : cl.z);
        }
        const bool act = dist < (T)xm::SOLVER_MARGIN;
        actf.v[i_] = act ? (T)1 : (T)0;
        tchf.v[i_] = dist < (T)xm::CONTACT_MARGIN ? (T)1 : (T)0;
        const V3<T> pn = b0 * nl.x + b1 * nl.y + b2 * nl.z;
        const V3<T> pp = cb + b0 * pl.x + b1 * pl.y + b2 * pl.z;
        const V3<T> pt1 = xk::plane_space(pn);
        S.lpn[0].v[i_] = pn.x; S.lpn[1].v[i_] = pn.y; S.lpn[2].v[i_] = pn.z;
        S.lpp[0].v[i_] = pp.x; S.lpp[1].v[i_] = pp.y; S.lpp[2].v[i_] = pp.z;
        S.lpt1[0].v[i_] = pt1.x; S.lpt1[1].v[i_] = pt1.y; S.lpt1[2].v[i_] = pt1.z;
        S.lpvt.v[i_] = dist < (T)0 ? -pad_erp * dist * idt : -dist * idt;
        S.lpl0.v[i_] = act ? (T)xm::WARMSTART * sel4(idx, st.lam_p[0], st.lam_p[1], st.lam_p[2], st.lam_p[3]) : (T)0;
    }
    S.pact[0] = lv_get<0>(actf) != (T)0; S.pact[1] = lv_get<3>(actf) != (T)0;
    S.pact[2] = lv_get<6>(actf) != (T)0; S.pact[3] = lv_get<9>(actf) != (T)0;
    S.pad_any = S.pact[0] || S.pact[1] || S.pact[2] || S.pact[3];
    const bool touch_f[2] = {lv_get<0>(tchf) != (T)0 || lv_get<3>(tchf) != (T)0, lv_get<6>(tchf) != (T)0 || lv_get<9>(tchf) != (T)0};
    S.mu_p = (T)xm::MU_OBJECT * (st.mug > (T)0.5 ? (T)xm::MU_FINGER_GRASP : (T)xm::MU_FINGER);
    // the touch flag is an output of the collision pass
    st.touch = (touch_f[0] && touch_f[1]) ? (T)1 : (T)0;
}

// rows -> Sweep (host: gather the 16 per-lane evaluations); base = slots 0 and 1, before the row set is chosen
template <typename T, typename Scene = xk::PnpScene>
XARM_HD void build_base(const Grp &G, const Setup<T> &S, Sweep<T> &W, LV<T> (&J)[R_G]) {
    XC_LANES {
        const int l = lane_of(G, i_);
        T R[R_N], a0[C0_N], a1[C1_N], iv[4];
        lane_rows_base<T, Scene>(S, l, R);
        lane_delassus_base<T, Scene>(S, l, R, a0, a1, iv);
#pragma unroll
        for (int k = 0; k < R_J2A; k++) J[k].v[i_] = R[k];
#pragma unroll
        for (int k = 0; k < 2; k++) { W.g[k].v[i_] = R[R_G + k]; W.lam[k].v[i_] = R[R_L0 + k]; W.invd[k].v[i_] = iv[k]; }
        W.lo1.v[i_] = R[R_LO1]; W.hi1.v[i_] = R[R_HI1];
#pragma unroll
        for (int k = 0; k < NT; k++) W.nA0[C0_T + k].v[i_] = a0[C0_T + k];
#pragma unroll
        for (int k = 0; k < NA1; k++) W.nA1[C1_A + k].v[i_] = a1[C1_A + k];
    }
}
template <typename T, typename Lds, bool PAD, bool LA, typename Scene = xk::PnpScene>
XARM_HD void build_extra(const Grp &G, const Setup<T> &S, Lds lds, Sweep<T> &W, LV<T> (&J)[R_G], LV<T> &cfm) {
    XC_LANES {
        const int l = lane_of(G, i_);
        T R[R_N], a1[C1_N], a2[C2_N], a3[C1_N], iv[4];
#pragma unroll
        for (int k = 0; k < R_N; k++) R[k] = k < R_J2A ? J[k].v[i_] : (T)0;
#pragma unroll
        for (int k = 0; k < C1_N; k++) a1[k] = a3[k] = (T)0;
#pragma unroll
        for (int k = 0; k < C2_N; k++) a2[k] = (T)0;
#pragma unroll
        for (int k = 0; k < 4; k++) iv[k] = (T)0;
        lane_rows_extra<T, Lds, PAD, LA>(S, lds, l, i_, R);
        lane_delassus_extra<T, PAD, LA, Scene>(S, l, R, a1, a2, a3, iv);
#pragma unroll
        for (int k = R_J2A; k < R_G; k++) J[k].v[i_] = R[k];
#pragma unroll
        for (int k = 2; k < 4; k++) { W.g[k].v[i_] = R[R_G + k]; W.lam[k].v[i_] = R[R_L0 + k]; W.invd[k].v[i_] = iv[k]; }
        cfm.v[i_] = R[R_CFM];
#pragma unroll
        for (int k = C1_L; k < C1_N; k++) W.nA1[k].v[i_] = a1[k];
#pragma unroll
        for (int k = NT; k < C0_N; k++) W.nA0[k].v[i_] = (T)0;
#pragma unroll
        for (int k = 0; k < C1_N; k++) W.nA3[k].v[i_] = a3[k];
#pragma unroll
        for (int k = 0; k < C2_N; k++) W.nA2[k].v[i_] = a2[k];
    }
}

template <typename T, typename Lds, bool PAD, bool LA>
XARM_HD void solve(const Grp &G, const Setup<T> &S, Lds lds, EnvState<T> &st, Sweep<T> &W, LV<T> (&J)[R_G], T (&tau)[15]) {
    LV<T> cfm = lv_fill((T)0);
    if (PAD || LA) build_extra<T, Lds, PAD, LA>(G, S, lds, W, J, cfm);
    bool padw[NP];
#pragma unroll
#if defined(XC_FORCE_FULL) || defined(XC_FORCE_PADW)
    for (int p = 0; p < NP; p++) padw[p] = PAD;
#else
    for (int p = 0; p < NP; p++) padw[p] = PAD && XARM_ANY_X(S.pact[p]);
#endif
    if (PAD) pad_columns<T, LA>(G, S, J, cfm, W, padw);
    apply_warm_start<T, PAD, LA>(W, padw);
    const T mu_t = (T)(xm::MU_OBJECT * xm::MU_TABLE);
    sweep_all<T, PAD, LA>(G, W, mu_t, S.mu_p, padw);
    // generalized impulse tau = sum_r J_r^T lam_r, reduced over the row
#pragma unroll
    for (int d = 0; d < 9; d++) {
        LV<T> c = lv_mul(J[R_J1 + d], W.lam[1]);
        if (PAD) c = lv_fma(J[R_J2A + d], W.lam[2], c);
        if (LA && d < 7) c = lv_fma(J[R_J3 + d], W.lam[3], c);
        tau[d] = lv_allsum(c);
    }
#pragma unroll
    for (int d = 0; d < 6; d++) {
        LV<T> c = lv_mul(J[R_J0 + d], W.lam[0]);
        if (PAD) c = lv_fma(J[R_J2B + d], W.lam[2], c);
        tau[9 + d] = lv_allsum(c);
    }
    // normal impulses for the next substep's warm start
    const T lt[NTS] = {lv_get<0>(W.lam[0]), lv_get<3>(W.lam[0]), lv_get<6>(W.lam[0]), lv_get<9>(W.lam[0])};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        T l = (T)0;
#pragma unroll
        for (int s = 0; s < NTS; s++) l = S.tid[s] == i ? lt[s] : l;
        st.lam_t[i] = l;
        st.lam_p[i] = (T)0;
    }
    if (PAD) {
        st.lam_p[0] = lv_get<0>(W.lam[2]); st.lam_p[1] = lv_get<3>(W.lam[2]);
        st.lam_p[2] = lv_get<6>(W.lam[2]); st.lam_p[3] = lv_get<9>(W.lam[2]);
    }
}

// one internal substep of one environment, executed by the 16 lanes of its row
template <typename T, typename Lds>
XARM_HD void substep(const Grp &G, const ArmLane<T> &C, EnvState<T> &st, const T (&qt)[9], const T dt, Lds lds) {
    Setup<T> S;
    substep_setup<T, Lds>(G, C, st, qt, dt, lds, S);
    XARM_LDS_FENCE();
    T tau[15];
    Sweep<T> W;
    LV<T> J[R_G];
    build_base<T>(G, S, W, J);
    // wave-uniform choice of the row set: absent rows carry exactly zero impulse and every operation the row sets
    // share is a single correctly rounded one on bits built above, so the choice does not change an environment's result
#ifdef XC_FORCE_FULL
    const bool pad = true, la = true;   // test hook: every row set present
#elif defined(XC_FORCE_PAD)
    const bool pad = true, la = XARM_ANY_X(S.la_any);
#elif defined(XC_FORCE_LA)
    const bool pad = XARM_ANY_X(S.pad_any), la = true;
#else
    const bool pad = XARM_ANY_X(S.pad_any), la = XARM_ANY_X(S.la_any);
#endif
    if (pad) {
        if (la) solve<T, Lds, true, true>(G, S, lds, st, W, J, tau);
        else solve<T, Lds, true, false>(G, S, lds, st, W, J, tau);
    } else {
        if (la) solve<T, Lds, false, true>(G, S, lds, st, W, J, tau);
        else solve<T, Lds, false, false>(G, S, lds, st, W, J, tau);
    }
    // constrained velocities u = u_free + Minv_tot tau, then semi-implicit Euler
    const T imb = (T)(1.0 / xk::PnpScene::OBJ_MASS);
#pragma unroll
    for (int r = 0; r < 9; r++) {
        T s = S.dq[r];
#pragma unroll
        for (int c = 0; c < 9; c++) s += S.Minv[symi(r, c)] * tau[c];
        st.qd[r] = s;
        st.q[r] += dt * s;
    }
    const V3<T> vb = S.vb + mk<T>(tau[9], tau[10], tau[11]) * imb;
    const V3<T> wb = S.wb + symmul(S.Iinv, mk<T>(tau[12], tau[13], tau[14]));
    st.bp[0] += dt * vb.x; st.bp[1] += dt * vb.y; st.bp[2] += dt * vb.z;
    {
        const T idt = (T)1 / dt;
        T ang = xsqrt(dot(wb, wb));
        if (ang * dt > (T)0.7853981633974483) ang = (T)0.7853981633974483 * idt;
        T sw, cw;
        xk::xsincos((T)0.5 * ang * dt, sw, cw);
        const T k = ang < (T)0.001 ? (T)0.5 * dt - dt * dt * dt * (T)0.020833333333 * ang * ang : sw / ang;
        const V3<T> ax = wb * k;
        const T x = st.bq[0], y = st.bq[1], z = st.bq[2], w0 = st.bq[3];
        const T nx = cw * x + ax.x * w0 + ax.y * z - ax.z * y;
        const T ny = cw * y + ax.y * w0 + ax.z * x - ax.x * z;
        const T nz = cw * z + ax.z * w0 + ax.x * y - ax.y * x;
        const T nw = cw * w0 - ax.x * x - ax.y * y - ax.z * z;
        const T inv = (T)1 / xsqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        st.bq[0] = nx * inv; st.bq[1] = ny * inv; st.bq[2] = nz * inv; st.bq[3] = nw * inv;
    }
    st.bv[0] = vb.x; st.bv[1] = vb.y; st.bv[2] = vb.z;
    st.bw[0] = wb.x; st.bw[1] = wb.y; st.bw[2] = wb.z;
}

template <typename T, typename Lds> XARM_HD void sim_tick(const Grp &G, const ArmLane<T> &C, EnvState<T> &st, const T (&qt)[9], Lds lds) {
    const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
#pragma unroll 1
    for (int k = 0; k < xm::PNP_N_SUBSTEPS; k++) substep<T, Lds>(G, C, st, qt, dt, lds);
}

// XarmPickAndPlace.reset (:121-127) = _reset_sim (:250-267) + _sample_goal (:269-287); same sequence as xk::env_reset
template <typename T, typename Lds> XARM_HD void env_reset(const Grp &G, const EnvCfg &cfg, int64_t env, EnvState<T> &st, Lds lds) {
    const ArmLane<T> C = arm_lane_consts<T>(G);
    T qt[9];
    const int64_t episode = (int64_t)st.episode + 1;
    const V3<T> start = mk<T>((T)xm::PNP_START_GRIPPER_POS[0], (T)xm::PNP_START_GRIPPER_POS[1], (T)xm::PNP_START_GRIPPER_POS[2]);
#pragma unroll 1
    for (int k = 0; k < xm::PNP_RESET_TICKS + 1; k++) {
        if (k < xm::PNP_RESET_TICKS) {
            xk::ik_solve(st.q, start, qt);
            qt[7] = qt[8] = (T)xm::PNP_RESET_FINGER_TARGET;
        } else {
            T u[8];
            xk::sample_draws(cfg, env, episode, u);
            xk::sample_object(cfg, u, st);
            xk::sample_goal(cfg, u, st);
        }
        sim_tick<T, Lds>(G, C, st, qt, lds);
    }
    st.steps = (T)0;
    st.episode = (T)episode;
}

// XarmPickAndPlace.step (:107-119) on a 16-lane row; same sequence as xk::env_step
template <typename T, typename Lds>
XARM_HD void env_step(const Grp &G, const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&obs)[xk::OBS_DIM], T &reward, bool &done,
                      bool &success, Lds lds) {
    const ArmLane<T> C = arm_lane_consts<T>(G);
    st.steps += (T)1;
    T a[4], qt[9];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    xk::Frame<T> f = xk::frame_identity<T>();
#pragma unroll
    for (int i = 0; i < 7; i++) xk::fk_advance(f, i, st.q[i]);
    const T sc = (T)(xm::PNP_MAX_VEL * xm::PNP_ACTION_DT);
    const V3<T> target = mk<T>(clampT(f.o.x + a[0] * sc, (T)xm::PNP_POS_LOW[0], (T)xm::PNP_POS_HIGH[0]),
                               clampT(f.o.y + a[1] * sc, (T)xm::PNP_POS_LOW[1], (T)xm::PNP_POS_HIGH[1]),
                               clampT(f.o.z + a[2] * sc, (T)xm::PNP_POS_LOW[2], (T)xm::PNP_POS_HIGH[2]));
    const T g = clampT(st.q[7] + a[3] * (T)(xm::PNP_ACTION_DT * xm::PNP_MAX_GRIPPER_VEL), (T)xm::PNP_GRIPPER_LOW, (T)xm::PNP_GRIPPER_HIGH);
    xk::ik_solve(st.q, target, qt);
    qt[7] = qt[8] = g;
    st.mug = st.touch;
    sim_tick<T, Lds>(G, C, st, qt, lds);
    xk::get_obs(st, obs);
    const T dx = st.bp[0] - st.goal[0], dy = st.bp[1] - st.goal[1], dz = st.bp[2] - st.goal[2];
    const T dist = xsqrt(dx * dx + dy * dy + dz * dz);
    success = dist < (T)xm::PNP_DISTANCE_THRESHOLD;
    reward = cfg.reward_type == 2 ? xk::dense_reward<T>(st, obs, dist) : xk::reward_of<T>(cfg.reward_type, dist);
    done = success || ((int)st.steps == xm::PNP_MAX_EPISODE_STEPS);
}

// substeps [k0, 15) of a step on a 16-lane row: k0 == 0 is env_step, a later one continues the step a fast stage opened (xk::env_step_fast_range)
template <typename T, typename Lds>
XARM_HD void env_step_from(const Grp &G, const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&qt)[9], int k0, T (&obs)[xk::OBS_DIM], T &reward,
                           bool &done, bool &success, Lds lds) {
    const ArmLane<T> C = arm_lane_consts<T>(G);
    if (k0 == 0) xk::step_open(st, act, qt);
    const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
#pragma unroll 1
    for (int k = k0; k < xm::PNP_N_SUBSTEPS; k++) substep<T, Lds>(G, C, st, qt, dt, lds);
    xk::step_close(cfg, st, obs, reward, done, success);
}

} // namespace xc
