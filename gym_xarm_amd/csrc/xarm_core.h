// xarm_core.h - per-environment simulation core of the batched XarmPickAndPlace environment.
//
// One thread owns one environment.  Everything here is straight-line, fully unrolled scalar
// code over compile-time model constants (xarm7_pd_model.h) so that the 9-dof state, the
// inverse mass matrix and the contact blocks live in VGPRs; the two matrices that are used
// once per solver sweep (the 6x7 hand Jacobian S and T = M^-1 S^T) are staged in LDS through
// the `Lds` accessor (one column of 64 lanes per value -> conflict-free ds_read_b32).
//
// What is computed (reference: /root/reference/gym_xarm/envs/xarm_pick_and_place.py):
//   env_step   = XarmPickAndPlace.step          :107-119  (+ _set_action :199-218, _get_obs :220-248)
//   env_reset  = XarmPickAndPlace.reset         :121-127  (+ _reset_sim :250-267, _sample_goal :269-287)
//   reward     = XarmPickAndPlace.compute_reward :155-177 (sparse, dense_o2g)
//   sim_tick   = p.stepSimulation() with numSubSteps = 15 (:64,:111)
//   ik_solve   = p.calculateInverseKinematics(..., [1,0,0,0], maxNumIterations=15) (:207,:253)
//
// The algorithm is NOT the oracle's: the oracle (oracle/xarm_oracle.c) walks a generic link
// tree with ABA in link coordinates and keeps one Jacobian + unit-impulse response per solver
// row.  Here the joint-space inertia comes from a world-frame composite-rigid-body pass and a
// 9x9 Cholesky factorisation, M^-1 is formed once per substep, and every finger/object contact
// point is solved as a 3x3 block in the 8-dimensional operational space (hand twist + two
// finger slides), which is algebraically the same Gauss-Seidel sweep in the same row order.
//
// The file compiles for gfx950 (hipcc) and, for the CPU-side unit tests and sanitizer runs
// only, for the host (g++ -DXARM_HOST_BUILD, tests/hostbuild/).  The host build is never part
// of the product: gym_xarm_amd loads libxarm_hip.so or raises.
#pragma once
#include <stdint.h>
#include <math.h>
#include "xarm7_pd_model.h"

#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
#define XARM_HD __device__ __forceinline__
// wave-uniform "does any lane need this block" (64-wide ballot)
#define XARM_ANY(p) (__builtin_amdgcn_ballot_w64(p) != 0ull)
// same ballot, for decisions that select between two exact code paths (the host build then follows the
// per-environment predicate, so both paths are exercised by the CPU tests)
#define XARM_ANY_X(p) (__builtin_amdgcn_ballot_w64(p) != 0ull)
// Compiler-only fence.  The LDS columns are lane-private and never cross a barrier, so LLVM would
// otherwise forward the staged values through registers (and then spill them to scratch).
#define XARM_LDS_FENCE() asm volatile("" ::: "memory")
#else
#define XARM_HD inline
// host build: always run the masked path so that the predication logic itself is tested (tools/flopcount defines
// XARM_HOST_ANY_PER_ENV to follow the per-environment predicate instead: the algorithmic operation count)
#ifdef XARM_HOST_ANY_PER_ENV
#define XARM_ANY(p) (p)
#else
#define XARM_ANY(p) (true)
#endif
#define XARM_ANY_X(p) (p)
#define XARM_LDS_FENCE() ((void)0)
#endif

// timing probes only (tools/variant_time.sh): -DXK_SWEEP_ITERS=n builds the one-env-per-lane cores with n solver sweeps, to
// split a kernel's time into per-substep setup and sweeps; xarm_version() then reports a TIMING VARIANT (xarm_hip.hip)
#ifdef XK_SWEEP_ITERS
#define XARM_SWEEP_VARIANT XK_SWEEP_ITERS
#else
#define XK_SWEEP_ITERS xm::NUM_ITERATIONS
#endif

namespace xk {

constexpr int STATE_DIM = 54;
constexpr int OBS_DIM = 24;
constexpr int GOAL_DIM = 3;
constexpr int ACT_DIM = 4;
// per-env LDS columns: S (6x7 hand Jacobian) | T = Minv[:,0:7] S^T (9x6) | A_hh (6x6 sym) | table slots
constexpr int LDS_S = 0, LDS_T = 42, LDS_AHH = 96, LDS_TBL = 117;
constexpr int LDS_FLOATS = 117 + 8 * 4; // 149 floats = 596 B per env (160 KiB / 256 envs per CU = 640 B)
constexpr int NTS = 4;         // object/table manifold slots (Bullet keeps <= 4 points)
constexpr int NP = 2 * xm::NPAD; // finger pad spheres (both fingers)

// state row layout (API edge, row-major [E, 54]; internal device storage is [54][E])
enum { S_Q = 0, S_QD = 9, S_BP = 18, S_BQ = 21, S_BV = 25, S_BW = 28, S_GOAL = 31, S_LT = 34, S_LP = 42,
       S_TOUCH = 50, S_MUG = 51, S_STEPS = 52, S_EPISODE = 53 };

template <typename T> struct V3 { T x, y, z; };
template <typename T> XARM_HD V3<T> mk(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename T> XARM_HD V3<T> operator+(V3<T> a, V3<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> XARM_HD V3<T> operator-(V3<T> a, V3<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> XARM_HD V3<T> operator*(V3<T> a, T s) { return mk<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> XARM_HD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> XARM_HD V3<T> cross(V3<T> a, V3<T> b) {
    return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// symmetric 3x3 (xx, xy, xz, yy, yz, zz) times vector
template <typename T> XARM_HD V3<T> symmul(const T (&s)[6], V3<T> v) {
    return mk<T>(s[0] * v.x + s[1] * v.y + s[2] * v.z, s[1] * v.x + s[3] * v.y + s[4] * v.z,
                 s[2] * v.x + s[4] * v.y + s[5] * v.z);
}
// component-wise select (a conditional expression on two V3 lvalues selects an ADDRESS, which keeps both in memory)
template <typename T> XARM_HD V3<T> selv(bool c, V3<T> a, V3<T> b) { return mk<T>(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
// two values that are always updated together (pairs of joint velocities / of Minv column entries): on the device
// a float pair is an aligned VGPR pair and `pkfma` one v_pk_fma_f32 with the scalar broadcast by op_sel - a lone
// wavefront issues a packed fma at the rate of a scalar one (tools/probes/lane_probe.hip), so this halves the
// instruction count of the column updates.  Host / double: two scalars.
template <typename T> struct Pk { T a, b; };
template <typename T> XARM_HD T pklo(const Pk<T> &p) { return p.a; }
template <typename T> XARM_HD T pkhi(const Pk<T> &p) { return p.b; }
template <typename T> XARM_HD Pk<T> mkpk(T a, T b) { Pk<T> r; r.a = a; r.b = b; return r; }
template <typename T> XARM_HD Pk<T> pkfma(const Pk<T> &m, T s, const Pk<T> &c) { Pk<T> r; r.a = m.a * s + c.a; r.b = m.b * s + c.b; return r; }
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
typedef float xf2 __attribute__((ext_vector_type(2)));
template <> struct Pk<float> { xf2 v; };
XARM_HD float pklo(const Pk<float> &p) { return p.v.x; }
XARM_HD float pkhi(const Pk<float> &p) { return p.v.y; }
template <> XARM_HD Pk<float> mkpk<float>(float a, float b) { Pk<float> r; r.v.x = a; r.v.y = b; return r; }
XARM_HD Pk<float> pkfma(const Pk<float> &m, float s, const Pk<float> &c) {
    Pk<float> r;
    xf2 sp; sp.x = s; sp.y = s;
    r.v = __builtin_elementwise_fma(m.v, sp, c.v);
    return r;
}
#endif
template <typename T> XARM_HD T clampT(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
// The solver's clamps (lo <= hi always holds there).  On the device one v_med3_f32 / v_max_f32 each: written as comparisons
// they compile to two v_cmp + two v_cndmask, ~75 of the ~960 instructions of a k_step sweep.  Same value for finite input
// (the sign of a zero result may differ); the host build and float64 keep the comparisons.
template <typename T> XARM_HD T sclamp(T v, T lo, T hi) { return clampT(v, lo, hi); }
template <typename T> XARM_HD T smax0(T v) { return v < (T)0 ? (T)0 : v; }
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
XARM_HD float sclamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
XARM_HD float smax0(float v) { return __builtin_fmaxf(v, 0.0f); }
#endif
template <typename T> XARM_HD T comp(V3<T> v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

XARM_HD float xsqrt(float x) { return sqrtf(x); }
XARM_HD double xsqrt(double x) { return sqrt(x); }
XARM_HD float xsin(float x) { return sinf(x); }
XARM_HD double xsin(double x) { return sin(x); }
XARM_HD float xcos(float x) { return cosf(x); }
XARM_HD double xcos(double x) { return cos(x); }
// joint angles stay within +-2pi: the hardware v_sin_f32 / v_cos_f32 (abs. error ~1e-6) are accurate
// enough for the 5e-4 parity tolerance and ~40x cheaper than the ocml sinf/cosf with their large-
// argument reduction (which made up a third of the kernel's instructions)
#if defined(__HIPCC__) && !defined(XARM_HOST_BUILD)
XARM_HD void xsincos(float x, float &s, float &c) { s = __sinf(x); c = __cosf(x); }
#else
XARM_HD void xsincos(float x, float &s, float &c) { s = sinf(x); c = cosf(x); }
#endif
XARM_HD void xsincos(double x, double &s, double &c) { s = sin(x); c = cos(x); }
XARM_HD float xatan2(float y, float x) { return atan2f(y, x); }
XARM_HD double xatan2(double y, double x) { return atan2(y, x); }
XARM_HD float xremainder(float x, float y) { return remainderf(x, y); }
XARM_HD double xremainder(double x, double y) { return remainder(x, y); }
XARM_HD float xasin(float x) { return asinf(x); }
XARM_HD double xasin(double x) { return asin(x); }
XARM_HD float xabs(float x) { return fabsf(x); }
XARM_HD double xabs(double x) { return fabs(x); }
XARM_HD float xpow(float x, float y) { return powf(x, y); }
XARM_HD double xpow(double x, double y) { return pow(x, y); }

// packed lower-triangular index, i >= j
XARM_HD constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
XARM_HD constexpr int symi(int i, int j) { return i >= j ? tri(i, j) : tri(j, i); }

template <typename T> struct EnvState {
    T q[9], qd[9];
    T bp[3], bq[4], bv[3], bw[3];
    T goal[3];
    T lam_t[8], lam_p[8];
    T touch, mug, steps, episode;
};

struct EnvCfg {
    uint64_t seed;
    int64_t env_id_offset;
    float init_grasp_rate, goal_ground_rate;
    int goal_shape;  // 0 air, 1 ground
    int reward_type; // 0 sparse, 1 dense_o2g, 2 dense (staged, uses the contact flags)
};

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (bit-identical to oracle/xarm_oracle.c:xo_philox)
XARM_HD void philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
template <typename T> XARM_HD T u01(uint32_t x) { return (T)(x >> 8) * (T)(1.0 / 16777216.0); }

// ---------------------------------------------------------------------------------------------
// forward kinematics of the 7 revolute joints: frame columns c0,c1,c2 and origin o after joint i
template <typename T> struct Frame { V3<T> c0, c1, c2, o; };

template <typename T> XARM_HD void fk_advance(Frame<T> &f, int i, T q) {
    // translate by the joint origin, roll about x by the URDF rpy, rotate about local z by q
    f.o = f.o + f.c0 * (T)xm::ORG_P[i][0] + f.c1 * (T)xm::ORG_P[i][1] + f.c2 * (T)xm::ORG_P[i][2];
    const T c = (T)xm::ORG_C[i], s = (T)xm::ORG_S[i];
    V3<T> n1 = f.c1 * c + f.c2 * s, n2 = f.c2 * c - f.c1 * s;
    T cq, sq;
    xsincos(q, sq, cq);
    V3<T> m0 = f.c0 * cq + n1 * sq, m1 = n1 * cq - f.c0 * sq;
    f.c0 = m0; f.c1 = m1; f.c2 = n2;
}
template <typename T> XARM_HD Frame<T> frame_identity() {
    Frame<T> f;
    f.c0 = mk<T>(1, 0, 0); f.c1 = mk<T>(0, 1, 0); f.c2 = mk<T>(0, 0, 1); f.o = mk<T>(0, 0, 0);
    return f;
}

// ---------------------------------------------------------------------------------------------
// damped-least-squares IK for link_eef, orientation target quaternion (1,0,0,0) = diag(1,-1,-1)
template <typename T> XARM_HD V3<T> rot_error(const Frame<T> &f) {
    // Re = Rt * Rc^T with Rt = diag(1,-1,-1); Rc columns c0,c1,c2 -> Rc^T rows are the columns
    // Re[r][c] = Rt[r][r] * Rc[c][r]
    const T r00 = f.c0.x, r01 = f.c0.y, r02 = f.c0.z;       // row 0 of Re = (Rc[0][0], Rc[1][0], Rc[2][0])
    const T r10 = -f.c1.x, r11 = -f.c1.y, r12 = -f.c1.z;    // row 1 = -(Rc[0][1], Rc[1][1], Rc[2][1])
    const T r20 = -f.c2.x, r21 = -f.c2.y, r22 = -f.c2.z;
    V3<T> vee = mk<T>((T)0.5 * (r21 - r12), (T)0.5 * (r02 - r20), (T)0.5 * (r10 - r01));
    const T s = xsqrt(dot(vee, vee)), c = (T)0.5 * (r00 + r11 + r22 - (T)1);
    if (s > (T)1e-6) return vee * (xatan2(s, c) / s);
    if (c > (T)0) return vee;
    T ax[3];
    const T dg[3] = {r00, r11, r22};
#pragma unroll
    for (int i = 0; i < 3; i++) { T d = (T)0.5 * (dg[i] + (T)1); ax[i] = xsqrt(d > (T)0 ? d : (T)0); }
    int k = 0;
    if (ax[1] > ax[k]) k = 1;
    if (ax[2] > ax[k]) k = 2;
    const T Re[9] = {r00, r01, r02, r10, r11, r12, r20, r21, r22};
#pragma unroll
    for (int i = 0; i < 3; i++)
        if (i != k && (Re[k * 3 + i] + Re[i * 3 + k]) < (T)0) ax[i] = -ax[i];
    const T pi = (T)3.14159265358979323846;
    return mk<T>(pi * ax[0], pi * ax[1], pi * ax[2]);
}

template <typename T, int MAXIT> XARM_HD void ik_arm(const T (&q_in)[7], V3<T> target, T (&q_out)[7], const Frame<T> base = frame_identity<T>()) {
    T q[7];
#pragma unroll
    for (int i = 0; i < 7; i++) q[i] = q_in[i];
    bool done = false;
#pragma unroll 1
    for (int it = 0; it < MAXIT; it++) {
        V3<T> o[7], a[7];
        Frame<T> f = base;
#pragma unroll
        for (int i = 0; i < 7; i++) { fk_advance(f, i, q[i]); o[i] = f.o; a[i] = f.c2; }
        V3<T> ep = target - f.o;
        if (xsqrt(dot(ep, ep)) < (T)xm::IK_RESIDUAL) done = true;
        if (!XARM_ANY(!done)) break;
        V3<T> eo = rot_error(f);
        const T err[6] = {ep.x, ep.y, ep.z, eo.x, eo.y, eo.z};
        T J[6][7];
#pragma unroll
        for (int i = 0; i < 7; i++) {
            V3<T> c = cross(a[i], f.o - o[i]);
            J[0][i] = c.x; J[1][i] = c.y; J[2][i] = c.z;
            J[3][i] = a[i].x; J[4][i] = a[i].y; J[5][i] = a[i].z;
        }
        T L[21]; // A = J J^T + lambda^2 I, then its Cholesky factor in place (packed lower)
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) {
                T s = (r == c) ? (T)(xm::IK_LAMBDA * xm::IK_LAMBDA) : (T)0;
#pragma unroll
                for (int k = 0; k < 7; k++) s += J[r][k] * J[c][k];
                L[tri(r, c)] = s;
            }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c <= r; c++) {
                T s = L[tri(r, c)];
#pragma unroll
                for (int k = 0; k < c; k++) s -= L[tri(r, k)] * L[tri(c, k)];
                L[tri(r, c)] = (r == c) ? xsqrt(s) : s / L[tri(c, c)];
            }
        T y[6], x[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            T s = err[r];
#pragma unroll
            for (int k = 0; k < r; k++) s -= L[tri(r, k)] * y[k];
            y[r] = s / L[tri(r, r)];
        }
#pragma unroll
        for (int r = 5; r >= 0; r--) {
            T s = y[r];
#pragma unroll
            for (int k = r + 1; k < 6; k++) s -= L[tri(k, r)] * x[k];
            x[r] = s / L[tri(r, r)];
        }
        T dq[7], mx = (T)0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            T s = (T)0;
#pragma unroll
            for (int r = 0; r < 6; r++) s += J[r][k] * x[r];
            dq[k] = s;
            mx = xabs(s) > mx ? xabs(s) : mx;
        }
        const T sc = mx > (T)xm::IK_MAX_DTHETA ? (T)xm::IK_MAX_DTHETA / mx : (T)1;
        if (!done) {
#pragma unroll
            for (int k = 0; k < 7; k++) q[k] += sc * dq[k];
        }
    }
#pragma unroll
    for (int i = 0; i < 7; i++) q_out[i] = q[i];
}
// PickAndPlace: maxNumIterations = n_substeps = 15 (:207); finger entries are passed through
template <typename T> XARM_HD void ik_solve(const T (&q_in)[9], V3<T> target, T (&q_out)[9]) {
    T qa[7], qo[7];
#pragma unroll
    for (int i = 0; i < 7; i++) qa[i] = q_in[i];
    ik_arm<T, xm::PNP_N_SUBSTEPS>(qa, target, qo);
#pragma unroll
    for (int i = 0; i < 7; i++) q_out[i] = qo[i];
    q_out[7] = q_in[7];
    q_out[8] = q_in[8];
}

// btPlaneSpace1: t1; t2 = n x t1
template <typename T> XARM_HD V3<T> plane_space(V3<T> n) {
    if (xabs(n.z) > (T)0.7071067811865475244) {
        const T a = n.y * n.y + n.z * n.z, k = (T)1 / xsqrt(a);
        return mk<T>((T)0, -n.z * k, n.y * k);
    }
    const T a = n.x * n.x + n.y * n.y, k = (T)1 / xsqrt(a);
    return mk<T>(-n.y * k, n.x * k, (T)0);
}

// spatial helpers; a spatial vector is (w = angular, v = linear about the WORLD origin)
template <typename T> struct SV { V3<T> w, v; };
// rigid-body inertia about the world origin: mass, h = m*c, Ibar (sym 6)
template <typename T> struct RBI { T m; V3<T> h; T I[6]; };
template <typename T> XARM_HD SV<T> rbi_mul(const RBI<T> &I, SV<T> x) {
    SV<T> r;
    r.w = symmul(I.I, x.w) + cross(I.h, x.v);
    r.v = x.v * I.m - cross(I.h, x.w);
    return r;
}
template <typename T> XARM_HD T sdot(SV<T> a, SV<T> b) { return dot(a.w, b.w) + dot(a.v, b.v); }

template <typename T> struct PadPoint {
    V3<T> p, n, t1;
    V3<T> Kn, Kt1, Kt2;
    T invd[3];
    T lam[3];
    T vt;
};
template <typename T> struct TablePoint {
    V3<T> r;
    T lam[3];
    T vt;
    int id;
};

// ---------------------------------------------------------------------------------------------
// Scene traits: what differs between the single-arm PickAndPlace scene and the dual-arm Handover scene.
struct PnpScene {
    static constexpr int NARMS = 1;
    static constexpr bool HAS_STAND = false;   // no static support box besides the table
    static constexpr double OBJ_HX = xm::PNP_OBJ_HALF[0], OBJ_HY = xm::PNP_OBJ_HALF[1], OBJ_HZ = xm::PNP_OBJ_HALF[2];
    static constexpr double OBJ_MASS = xm::PNP_OBJ_MASS;
    static constexpr double TIME_STEP = xm::PNP_TIME_STEP;              // p.setTimeStep: motor / gear max impulse = force * timeStep
    static constexpr double FINGER_MOTOR_FORCE = xm::PNP_FINGER_MOTOR_FORCE;
    static constexpr double LIN_DAMP_FACTOR = xm::LIN_DAMP_FACTOR, ANG_DAMP_FACTOR = xm::ANG_DAMP_FACTOR;
    template <typename T> static XARM_HD Frame<T> base_frame(int) { return frame_identity<T>(); }
    // support surface under the point p: the table top (:66), nothing beside it
    template <typename T> static XARM_HD bool support(V3<T> p, T &height) {
        height = (T)xm::TABLE_TOP_Z;
        return xabs(p.x) <= (T)xm::TABLE_HALF_X && xabs(p.y) <= (T)xm::TABLE_HALF_Y;
    }
};
// lane-pair exchange used by the dual-arm scene; the single-arm scene never exchanges
struct NoXchg {
    template <typename T> XARM_HD T from0(T v) const { return v; }   // value held by the lane of arm 0
    template <typename T> XARM_HD T from1(T v) const { return v; }   // ... of arm 1
    template <typename T> XARM_HD T partner(T v) const { return v; } // ... of the other arm
};

// ---------------------------------------------------------------------------------------------
// Arm part of a substep, shared by every scene: FK, world-frame RNEA / CRBA, 9x9 Cholesky -> Minv, unconstrained
// joint velocities, and the operational-space blocks S (hand Jacobian), T = Minv S^T, A_hh = S T staged in LDS.
template <typename T> struct ArmDyn {
    T Minv[45];        // packed lower 9x9
    T dq[9];           // unconstrained joint velocities after dt
    V3<T> hc0, hc1, hc2; // hand (link7) frame axes
    V3<T> fo[2];       // finger frame origins
};
// OPSPACE = false (cooperative core, xarm_coop_core.h): only S is staged, T and A_hh are not formed.
// STAGE_S = false (the pad-free fast step, substep<.., FAST>): nothing of the arm goes to LDS at all.
template <typename T, typename Lds, typename Scene, bool OPSPACE = true, bool STAGE_S = true>
XARM_HD void arm_dynamics(const T (&q_in)[9], const T (&qd_in)[9], const T dt, Lds lds, const int arm, ArmDyn<T> &A) {
    // ---------------- kinematics + world-frame RNEA / CRBA
    SV<T> S[7];      // joint motion axes about the world origin
    RBI<T> Ib[9];    // per-body inertia, later suffix-summed into composite inertias
    SV<T> fb[9];     // per-body bias force, later suffix-summed
    Frame<T> f = Scene::template base_frame<T>(arm);
    SV<T> vel, acc;
    vel.w = mk<T>(0, 0, 0); vel.v = mk<T>(0, 0, 0);
    acc.w = mk<T>(0, 0, 0); acc.v = mk<T>(0, 0, (T)xm::GRAVITY);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        fk_advance(f, i, q_in[i]);
        S[i].w = f.c2;
        S[i].v = cross(f.o, f.c2);
        // acc += (vel_parent x S_i) qd_i ; vel += S_i qd_i
        const T qd = qd_in[i];
        acc.w = acc.w + cross(vel.w, S[i].w) * qd;
        acc.v = acc.v + (cross(vel.w, S[i].v) + cross(vel.v, S[i].w)) * qd;
        vel.w = vel.w + S[i].w * qd;
        vel.v = vel.v + S[i].v * qd;
        // body inertia in world axes about the world origin
        const T m = (T)xm::MASS[i];
        V3<T> c = f.o + f.c0 * (T)xm::COM[i][0] + f.c1 * (T)xm::COM[i][1] + f.c2 * (T)xm::COM[i][2];
        const T ixx = (T)xm::INERTIA[i][0], ixy = (T)xm::INERTIA[i][1], ixz = (T)xm::INERTIA[i][2],
                iyy = (T)xm::INERTIA[i][3], iyz = (T)xm::INERTIA[i][4], izz = (T)xm::INERTIA[i][5];
        V3<T> m0 = f.c0 * ixx + f.c1 * ixy + f.c2 * ixz;
        V3<T> m1 = f.c0 * ixy + f.c1 * iyy + f.c2 * iyz;
        V3<T> m2 = f.c0 * ixz + f.c1 * iyz + f.c2 * izz;
        const T cc = dot(c, c);
        RBI<T> &I = Ib[i];
        I.m = m;
        I.h = c * m;
        I.I[0] = m0.x * f.c0.x + m1.x * f.c1.x + m2.x * f.c2.x + m * (cc - c.x * c.x);
        I.I[1] = m0.x * f.c0.y + m1.x * f.c1.y + m2.x * f.c2.y - m * c.x * c.y;
        I.I[2] = m0.x * f.c0.z + m1.x * f.c1.z + m2.x * f.c2.z - m * c.x * c.z;
        I.I[3] = m0.y * f.c0.y + m1.y * f.c1.y + m2.y * f.c2.y + m * (cc - c.y * c.y);
        I.I[4] = m0.y * f.c0.z + m1.y * f.c1.z + m2.y * f.c2.z - m * c.y * c.z;
        I.I[5] = m0.z * f.c0.z + m1.z * f.c1.z + m2.z * f.c2.z + m * (cc - c.z * c.z);
        // bias force f = I a + v x* (I v)
        SV<T> Iv = rbi_mul(I, vel), Ia = rbi_mul(I, acc);
        fb[i].w = Ia.w + cross(vel.w, Iv.w) + cross(vel.v, Iv.v);
        fb[i].v = Ia.v + cross(vel.w, Iv.v);
    }
    // hand frame = link7 frame; fingers slide along +/- hand y
    const V3<T> hc0 = f.c0, hc1 = f.c1, hc2 = f.c2, ho = f.o;
    A.hc0 = hc0; A.hc1 = hc1; A.hc2 = hc2;
    const SV<T> vh = vel, ah = acc;
    V3<T> (&fo)[2] = A.fo; // finger frame origins
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const T sg = k == 0 ? (T)1 : (T)-1;
        const V3<T> af = hc1 * sg;
        fo[k] = ho + hc2 * (T)xm::FINGER_Z + af * q_in[7 + k];
        const T qd = qd_in[7 + k];
        SV<T> v = vh, a = ah;
        a.v = a.v + cross(vh.w, af) * qd;
        v.v = v.v + af * qd;
        const T m = (T)xm::MASS[7 + k];
        V3<T> c = fo[k] + hc0 * (T)xm::COM[7 + k][0] + hc1 * (T)xm::COM[7 + k][1] + hc2 * (T)xm::COM[7 + k][2];
        const T cc = dot(c, c), ii = (T)xm::FINGER_INERTIA;
        RBI<T> &I = Ib[7 + k];
        I.m = m;
        I.h = c * m;
        I.I[0] = ii + m * (cc - c.x * c.x); I.I[1] = -m * c.x * c.y; I.I[2] = -m * c.x * c.z;
        I.I[3] = ii + m * (cc - c.y * c.y); I.I[4] = -m * c.y * c.z; I.I[5] = ii + m * (cc - c.z * c.z);
        SV<T> Iv = rbi_mul(I, v), Ia = rbi_mul(I, a);
        fb[7 + k].w = Ia.w + cross(v.w, Iv.w) + cross(v.v, Iv.v);
        fb[7 + k].v = Ia.v + cross(v.w, Iv.v);
    }
    // joint-space inertia (packed lower 9x9) and bias torques
    T M[45], tau[9];
    {
        const V3<T> af1 = hc1;
        // finger columns use the single-body finger inertias
        SV<T> F7, F8;
        F7.w = cross(Ib[7].h, af1); F7.v = af1 * Ib[7].m;
        F8.w = cross(af1, Ib[8].h); F8.v = af1 * (-Ib[8].m); // axis of finger 2 is -af1
        tau[7] = -dot(af1, fb[7].v);
        tau[8] = dot(af1, fb[8].v);
        M[tri(7, 7)] = Ib[7].m; M[tri(8, 8)] = Ib[8].m; M[tri(8, 7)] = (T)0;
#pragma unroll
        for (int i = 0; i < 7; i++) { M[tri(7, i)] = sdot(S[i], F7); M[tri(8, i)] = sdot(S[i], F8); }
        // composite of everything carried by the hand: body 6 + fingers
        RBI<T> Ic = Ib[6];
        SV<T> fc = fb[6];
#pragma unroll
        for (int k = 7; k < 9; k++) {
            Ic.m += Ib[k].m; Ic.h = Ic.h + Ib[k].h;
#pragma unroll
            for (int e = 0; e < 6; e++) Ic.I[e] += Ib[k].I[e];
            fc.w = fc.w + fb[k].w; fc.v = fc.v + fb[k].v;
        }
#pragma unroll
        for (int j = 6; j >= 0; j--) {
            if (j < 6) {
                Ic.m += Ib[j].m; Ic.h = Ic.h + Ib[j].h;
#pragma unroll
                for (int e = 0; e < 6; e++) Ic.I[e] += Ib[j].I[e];
                fc.w = fc.w + fb[j].w; fc.v = fc.v + fb[j].v;
            }
            SV<T> F = rbi_mul(Ic, S[j]);
#pragma unroll
            for (int i = 0; i <= j; i++) M[tri(j, i)] = sdot(S[i], F);
            tau[j] = -sdot(S[j], fc) - (T)xm::DAMPING[j] * qd_in[j];
        }
    }
    // Cholesky M = L L^T (in place), Linv, Minv = Linv^T Linv
    T (&Minv)[45] = A.Minv;
    {
        T rd[9]; // reciprocals of the diagonal of L: one division per column instead of one per entry
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
        T Li[45]; // inverse of L (lower)
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
    // unconstrained joint velocities
    T (&dq)[9] = A.dq;
#pragma unroll
    for (int r = 0; r < 9; r++) {
        T s = (T)0;
#pragma unroll
        for (int c = 0; c < 9; c++) s += Minv[symi(r, c)] * tau[c];
        dq[r] = qd_in[r] + dt * s;
    }
    // T = Minv[:, 0:7] S^T (9 x 6) -> LDS; A_hh = S T_a (6x6 sym) -> registers; T_f rows -> registers
    if (OPSPACE) {
        T Tm[9][6];
#pragma unroll
        for (int r = 0; r < 9; r++)
#pragma unroll
            for (int k = 0; k < 6; k++) {
                T s = (T)0;
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const T sik = k < 3 ? comp(S[i].w, k) : comp(S[i].v, k - 3);
                    s += Minv[symi(r, i)] * sik;
                }
                Tm[r][k] = s;
            }
#pragma unroll
        for (int k = 0; k < 6; k++)
#pragma unroll
            for (int l = 0; l <= k; l++) {
                T s = (T)0;
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const T sik = k < 3 ? comp(S[i].w, k) : comp(S[i].v, k - 3);
                    s += sik * Tm[i][l];
                }
                lds[LDS_AHH + tri(k, l)] = s;
            }
#pragma unroll
        for (int i = 0; i < 7; i++) {
            lds[LDS_S + i * 6 + 0] = S[i].w.x; lds[LDS_S + i * 6 + 1] = S[i].w.y; lds[LDS_S + i * 6 + 2] = S[i].w.z;
            lds[LDS_S + i * 6 + 3] = S[i].v.x; lds[LDS_S + i * 6 + 4] = S[i].v.y; lds[LDS_S + i * 6 + 5] = S[i].v.z;
        }
#pragma unroll
        for (int r = 0; r < 9; r++)
#pragma unroll
            for (int k = 0; k < 6; k++) lds[LDS_T + r * 6 + k] = Tm[r][k];
        XARM_LDS_FENCE();
    } else if (STAGE_S) {
#pragma unroll
        for (int i = 0; i < 7; i++) {
            lds[LDS_S + i * 6 + 0] = S[i].w.x; lds[LDS_S + i * 6 + 1] = S[i].w.y; lds[LDS_S + i * 6 + 2] = S[i].w.z;
            lds[LDS_S + i * 6 + 3] = S[i].v.x; lds[LDS_S + i * 6 + 4] = S[i].v.y; lds[LDS_S + i * 6 + 5] = S[i].v.z;
        }
        XARM_LDS_FENCE();
    }
    (void)ho;
}


// ---------------------------------------------------------------------------------------------
// one internal substep (dt = timeStep / numSubSteps): collide, unconstrained dynamics, rows, PGS, integrate.
// Dual-arm scenes run one arm per lane: `arm` selects the base frame, `xchg.from(a, v)` returns the copy of v
// held by the lane of arm a of the same environment (object velocities are handed over between the two
// finger/object phases of a sweep; everything else is either per-arm or computed identically by both lanes).
//
// FAST = true: the same substep with every finger-pad ROW compiled out - no operational-space blocks, no pad warm start,
// no F phase, nothing of the arm staged in LDS.  It is exact (the arithmetic of the sweep below: an inactive pad carries
// 1/diag = 0 and contributes nothing; the same bits in the host build and in a device build with -ffp-contract=on, the
// default device build contracts the two instantiations differently: last bits) for as long as no pad of the environment
// is within the solver margin of the object, and it says so: the return value is "a pad row of this environment is active in this substep".  The fast step kernel
// runs it on every environment and hands the ones that answer true to a kernel that solves pad rows (xarm_hip.hip,
// k_step_fast): ~2 % of the environments hold a finger contact, but that is >= 1 lane in most wavefronts, and a wavefront
// with one such lane sweeps the pad blocks for all 64 (k_step 1.88 ms against 0.74 ms for a contact-free batch).
template <typename T, typename Lds, typename Scene = PnpScene, typename Xchg = NoXchg, bool FAST = false>
XARM_HD bool substep(EnvState<T> &st, const T (&qt)[9], const T dt, Lds lds, const int arm = 0, const Xchg xchg = Xchg()) {
    const T idt = (T)1 / dt;
    ArmDyn<T> AD;
    arm_dynamics<T, Lds, Scene, !FAST, !FAST>(st.q, st.qd, dt, lds, arm, AD);
    T (&Minv)[45] = AD.Minv;
    T (&dq)[9] = AD.dq;
    const V3<T> hc0 = AD.hc0, hc1 = AD.hc1, hc2 = AD.hc2;
    const V3<T> (&fo)[2] = AD.fo;

    // ---------------- object: frame, inverse inertia, unconstrained motion
    V3<T> b0, b1, b2; // columns of Rb
    {
        const T x = st.bq[0], y = st.bq[1], z = st.bq[2], w = st.bq[3];
        b0 = mk<T>((T)1 - (T)2 * (y * y + z * z), (T)2 * (x * y + z * w), (T)2 * (x * z - y * w));
        b1 = mk<T>((T)2 * (x * y - z * w), (T)1 - (T)2 * (x * x + z * z), (T)2 * (y * z + x * w));
        b2 = mk<T>((T)2 * (x * z + y * w), (T)2 * (y * z - x * w), (T)1 - (T)2 * (x * x + y * y));
    }
    const V3<T> cb = mk<T>(st.bp[0], st.bp[1], st.bp[2]);
    const T hx = (T)Scene::OBJ_HX, hy = (T)Scene::OBJ_HY, hz = (T)Scene::OBJ_HZ;
    const T mb = (T)Scene::OBJ_MASS, imb = (T)(1.0 / Scene::OBJ_MASS);
    const T Ibx = (T)(Scene::OBJ_MASS / 3.0 * (Scene::OBJ_HY * Scene::OBJ_HY + Scene::OBJ_HZ * Scene::OBJ_HZ));
    const T Iby = (T)(Scene::OBJ_MASS / 3.0 * (Scene::OBJ_HX * Scene::OBJ_HX + Scene::OBJ_HZ * Scene::OBJ_HZ));
    const T Ibz = (T)(Scene::OBJ_MASS / 3.0 * (Scene::OBJ_HX * Scene::OBJ_HX + Scene::OBJ_HY * Scene::OBJ_HY));
    T Iinv[6];
    {
        const T ix = (T)1 / Ibx, iy = (T)1 / Iby, iz = (T)1 / Ibz;
        Iinv[0] = b0.x * b0.x * ix + b1.x * b1.x * iy + b2.x * b2.x * iz;
        Iinv[1] = b0.x * b0.y * ix + b1.x * b1.y * iy + b2.x * b2.y * iz;
        Iinv[2] = b0.x * b0.z * ix + b1.x * b1.z * iy + b2.x * b2.z * iz;
        Iinv[3] = b0.y * b0.y * ix + b1.y * b1.y * iy + b2.y * b2.y * iz;
        Iinv[4] = b0.y * b0.z * ix + b1.y * b1.z * iy + b2.y * b2.z * iz;
        Iinv[5] = b0.z * b0.z * ix + b1.z * b1.z * iy + b2.z * b2.z * iz;
    }
    V3<T> vb = mk<T>(st.bv[0], st.bv[1], st.bv[2]), wb = mk<T>(st.bw[0], st.bw[1], st.bw[2]);
    {
        // gyroscopic term as btRigidBody::computeGyroscopicImpulseImplicit_Body (Bullet's default): one Newton step of
        // I (w' - w) + dt w' x (I w') = 0 in body axes, J = I + dt ([w]x I - [I w]x), w' = w - J^-1 (dt w x I w).
        // (the explicit form gains energy every step and runs away once |w| dt ~ 1); then gravity and damping
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
        }
        vb.z -= dt * (T)xm::GRAVITY;
        // Bullet's pow(1 - damping, dt); dt is always timeStep / numSubSteps, folded by the header generator
        const T dl = (T)Scene::LIN_DAMP_FACTOR, da = (T)Scene::ANG_DAMP_FACTOR;
        vb = vb * dl;
        wb = wb * da;
    }

    // ---------------- (T) object corners against the table: first <= NTS active corners
    TablePoint<T> tp[NTS];
#pragma unroll
    for (int s = 0; s < NTS; s++) {
        tp[s].r = mk<T>(0, 0, 0); tp[s].vt = (T)0; tp[s].id = -1;
        tp[s].lam[0] = tp[s].lam[1] = tp[s].lam[2] = (T)0;
    }
    {
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            V3<T> r = b0 * ((i & 1) ? hx : -hx) + b1 * ((i & 2) ? hy : -hy) + b2 * ((i & 4) ? hz : -hz);
            V3<T> p = cb + r;
            T hsup;
            const bool sup = Scene::template support<T>(p, hsup);
            const T dist = p.z - hsup;
            const bool act = dist < (T)xm::SOLVER_MARGIN && sup && cnt < NTS;
            const T vt = dist < (T)0 ? -(T)xm::CONTACT_ERP * dist * idt : -dist * idt;
            const T l0 = (T)xm::WARMSTART * st.lam_t[i];
#pragma unroll
            for (int s = 0; s < NTS; s++) {
                const bool put = act && cnt == s;
                tp[s].r.x = put ? r.x : tp[s].r.x; tp[s].r.y = put ? r.y : tp[s].r.y; tp[s].r.z = put ? r.z : tp[s].r.z;
                tp[s].vt = put ? vt : tp[s].vt;
                tp[s].lam[0] = put ? l0 : tp[s].lam[0];
                tp[s].id = put ? i : tp[s].id;
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
                    tp[s].r.x = put ? r.x : tp[s].r.x; tp[s].r.y = put ? r.y : tp[s].r.y; tp[s].r.z = put ? r.z : tp[s].r.z;
                    tp[s].vt = put ? vt : tp[s].vt;
                    tp[s].lam[0] = put ? (T)0 : tp[s].lam[0];
                    tp[s].id = put ? 8 + v : tp[s].id;
                }
                cnt += act ? 1 : 0;
            }
        }
    }
    const T mu_t = (T)(xm::MU_OBJECT * xm::MU_TABLE);
#pragma unroll
    for (int s = 0; s < NTS; s++) {
        TablePoint<T> &P = tp[s];
        const bool act = P.id >= 0;
        // K = 1/m + C^T Iinv C, C = [r]x, columns c_j = r x e_j
        const V3<T> cx = mk<T>((T)0, P.r.z, -P.r.y), cy = mk<T>(-P.r.z, (T)0, P.r.x), cz = mk<T>(P.r.y, -P.r.x, (T)0);
        const V3<T> wx = symmul(Iinv, cx), wy = symmul(Iinv, cy), wz = symmul(Iinv, cz);
        const T K0 = imb + dot(cx, wx), K3 = imb + dot(cy, wy), K5 = imb + dot(cz, wz);
        // slot columns in LDS: K_xy K_xz K_yy K_yz K_zz, then 1/diag of the rows n = +z, t1 = -y, t2 = +x
        // (btPlaneSpace1 of (0,0,1))
        lds[LDS_TBL + s * 8 + 0] = dot(cx, wy); lds[LDS_TBL + s * 8 + 1] = dot(cx, wz); lds[LDS_TBL + s * 8 + 2] = K3;
        lds[LDS_TBL + s * 8 + 3] = dot(cy, wz); lds[LDS_TBL + s * 8 + 4] = K5;
        lds[LDS_TBL + s * 8 + 5] = act ? (T)1 / K5 : (T)0;
        lds[LDS_TBL + s * 8 + 6] = act ? (T)1 / K3 : (T)0;
        lds[LDS_TBL + s * 8 + 7] = act ? (T)1 / K0 : (T)0;
        // warm start: impulse lam0 * n on the object at r
        const V3<T> fi = mk<T>((T)0, (T)0, P.lam[0]);
        vb = vb + fi * imb;
        wb = wb + symmul(Iinv, cross(P.r, fi));
    }

    // ---------------- (M) motors, (L) limits, (G) gear: row constants
    T m_vt[9], m_invd[9], m_lam[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        m_vt[i] = (T)xm::MOTOR_KP * (qt[i] - st.q[i]) * idt + (T)(1.0 - xm::MOTOR_KD) * dq[i];
        m_invd[i] = (T)1 / Minv[tri(i, i)];
        m_lam[i] = (T)0;
    }
    const T m_hi_arm = (T)(xm::ARM_MOTOR_FORCE * Scene::TIME_STEP), m_hi_fin = (T)(Scene::FINGER_MOTOR_FORCE * Scene::TIME_STEP);
    // arm joints: range > 2 * window, so at most one side is inside the window
    T la_vt[7], la_sg[7], la_lam[7];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        const T g0 = st.q[i] - (T)xm::LOWER[i], g1 = (T)xm::UPPER[i] - st.q[i];
        const bool lo = g0 < (T)xm::LIMIT_WINDOW, hi = g1 < (T)xm::LIMIT_WINDOW;
        const T g = lo ? g0 : g1;
        la_sg[i] = lo ? (T)1 : (hi ? (T)-1 : (T)0);
        la_vt[i] = g < (T)0 ? -(T)xm::GLOBAL_ERP * g * idt : -g * idt;
        la_lam[i] = (T)0;
    }
    // finger joints: range (0.04) < window, both sides always present
    T lf_vt[2][2], lf_lam[2][2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const T g0 = st.q[7 + k] - (T)xm::LOWER[7 + k], g1 = (T)xm::UPPER[7 + k] - st.q[7 + k];
        lf_vt[k][0] = g0 < (T)0 ? -(T)xm::GLOBAL_ERP * g0 * idt : -g0 * idt;
        lf_vt[k][1] = g1 < (T)0 ? -(T)xm::GLOBAL_ERP * g1 * idt : -g1 * idt;
        lf_lam[k][0] = lf_lam[k][1] = (T)0;
    }
    const T g_vt = -(T)(xm::GEAR_ERP * xm::GLOBAL_ERP) * (st.q[7] - st.q[8]) * idt;
    const T g_hi = (T)(xm::GEAR_MAX_FORCE * Scene::TIME_STEP);
    const T g_invd = (T)1 / (Minv[tri(7, 7)] - (T)2 * Minv[tri(8, 7)] + Minv[tri(8, 8)]);
    T g_lam = (T)0;

    // ---------------- (F) finger pad spheres against the object
    static_assert(xm::NPAD == 2, "the finger block below fuses exactly two pad points per finger");
    PadPoint<T> pp[NP];
    // K21[fk]: change of the relative velocity at the finger's second pad point per unit impulse at its first one
    // (3x3, row-major); lets both points of a finger be swept before ONE operational-space update
    T K21[2][9];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int e = 0; e < 9; e++) K21[k][e] = (T)0;
    bool pad_any = false;
    bool touch_f[2] = {false, false};
    const T pad_denom = dt * (T)xm::FINGER_CONTACT_STIFFNESS + (T)(xm::FINGER_CONTACT_DAMPING + xm::OBJECT_CONTACT_DAMPING);
    const T pad_cfm = ((T)1 / pad_denom) * idt, pad_erp = dt * (T)xm::FINGER_CONTACT_STIFFNESS / pad_denom;
    {
        T wtot[8];
#pragma unroll
        for (int k = 0; k < 8; k++) wtot[k] = (T)0;
        const V3<T> vb_pre = vb, wb_pre = wb;
#pragma unroll
        for (int idx = 0; idx < NP; idx++) {
            const int fk = idx / xm::NPAD, j = idx % xm::NPAD;
            const T sg = fk == 0 ? (T)1 : (T)-1;
            PadPoint<T> &P = pp[idx];
            const V3<T> c = fo[fk] + hc0 * (T)xm::PAD_C[j][0] + hc1 * (sg * (T)xm::PAD_C[j][1]) + hc2 * (T)xm::PAD_C[j][2];
            // sphere against box, in box axes
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
            // touch flag: inside Bullet's 0.02 contact-breaking margin (getContactPoints); solver rows
            // only for points that can receive an impulse within one substep (dist < SOLVER_MARGIN)
            const bool act = dist < (T)xm::SOLVER_MARGIN;
            touch_f[fk] = touch_f[fk] || (dist < (T)xm::CONTACT_MARGIN);
            pad_any = pad_any || act;
            P.n = b0 * nl.x + b1 * nl.y + b2 * nl.z;
            P.p = cb + b0 * pl.x + b1 * pl.y + b2 * pl.z;
            P.t1 = plane_space(P.n);
            P.vt = dist < (T)0 ? -pad_erp * dist * idt : -dist * idt;
            P.lam[0] = act ? (T)xm::WARMSTART * st.lam_p[idx] : (T)0;
            P.lam[1] = P.lam[2] = (T)0;
            P.invd[0] = P.invd[1] = P.invd[2] = (T)0;
            P.Kn = P.Kt1 = P.Kt2 = mk<T>(0, 0, 0);
            if (!FAST && XARM_ANY(act)) {
                // 3x3 point Delassus block K = K_A (arm side, through A) + K_B (object side)
                const V3<T> af = hc1 * sg;
                const V3<T> r = P.p - cb;
                T K[3][3];
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const V3<T> ej = mk<T>(e == 0 ? (T)1 : (T)0, e == 1 ? (T)1 : (T)0, e == 2 ? (T)1 : (T)0);
                    const V3<T> mo = cross(P.p, ej);
                    const T W[6] = {mo.x, mo.y, mo.z, ej.x, ej.y, ej.z};
                    const T wf = comp(af, e);
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
                    const V3<T> vbj = ej * imb - cross(r, symmul(Iinv, cross(r, ej)));
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
                // warm start: +lam0 n on the finger, -lam0 n on the object
                const V3<T> fi = P.n * P.lam[0];
                const V3<T> mo = cross(P.p, fi);
                wtot[0] += mo.x; wtot[1] += mo.y; wtot[2] += mo.z;
                wtot[3] += fi.x; wtot[4] += fi.y; wtot[5] += fi.z;
                wtot[6 + fk] += dot(af, fi);
                vb = vb - fi * imb;
                wb = wb - symmul(Iinv, cross(r, fi));
            }
        }
        if (!FAST && XARM_ANY(pad_any)) {
#pragma unroll
            for (int r = 0; r < 9; r++) {
                T s = Minv[symi(r, 7)] * wtot[6] + Minv[symi(r, 8)] * wtot[7];
#pragma unroll
                for (int k = 0; k < 6; k++) s += lds[LDS_T + r * 6 + k] * wtot[k];
                dq[r] += s;
            }
            // coupling of the two pad points of each finger: relative velocity at point 2 per unit impulse at point 1
#pragma unroll
            for (int fk = 0; fk < 2; fk++) {
                const PadPoint<T> &P1 = pp[2 * fk], &P2 = pp[2 * fk + 1];
                if (!XARM_ANY(P1.invd[0] != (T)0 && P2.invd[0] != (T)0)) continue;
                const V3<T> af = hc1 * (fk == 0 ? (T)1 : (T)-1);
                const V3<T> r1 = P1.p - cb, r2 = P2.p - cb;
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const V3<T> ej = mk<T>(e == 0 ? (T)1 : (T)0, e == 1 ? (T)1 : (T)0, e == 2 ? (T)1 : (T)0);
                    const V3<T> mo = cross(P1.p, ej);
                    const T W[6] = {mo.x, mo.y, mo.z, ej.x, ej.y, ej.z};
                    const T wf = comp(af, e);
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
                    const V3<T> vbj = ej * imb - cross(r2, symmul(Iinv, cross(r1, ej)));
                    K21[fk][0 * 3 + e] = va.x + vbj.x; K21[fk][1 * 3 + e] = va.y + vbj.y; K21[fk][2 * 3 + e] = va.z + vbj.z;
                }
            }
        }
        if (Scene::NARMS == 2 && !FAST) {
            // the object also receives the warm-start impulses of the other arm's pads
            const V3<T> dv = vb - vb_pre, dw = wb - wb_pre;
            vb = vb + mk<T>(xchg.partner(dv.x), xchg.partner(dv.y), xchg.partner(dv.z));
            wb = wb + mk<T>(xchg.partner(dw.x), xchg.partner(dw.y), xchg.partner(dw.z));
            // both lanes must hold bit-identical object velocities from here on: take arm 0's sum
            vb = mk<T>(xchg.from0(vb.x), xchg.from0(vb.y), xchg.from0(vb.z));
            wb = mk<T>(xchg.from0(wb.x), xchg.from0(wb.y), xchg.from0(wb.z));
        }
    }
    st.touch = (touch_f[0] && touch_f[1]) ? (T)1 : (T)0;

    // two arms: the finger phases of the two arms commute unless both arms touch the object; `seq` is wave-uniform
    bool other_any = false, seq = false;
    if (Scene::NARMS == 2 && !FAST) {   // (the fast substep has no pad rows: the arms never interact)
        other_any = xchg.partner(pad_any ? (T)1 : (T)0) != (T)0;
        seq = XARM_ANY_X(pad_any && other_any);
    }
    // does any lane of the wavefront have an arm joint inside its limit window? (rare; one test instead of seven per sweep)
    bool la_lane = false;
#pragma unroll
    for (int i = 0; i < 7; i++) la_lane = la_lane || la_sg[i] != (T)0;
    const bool la_wave = XARM_ANY(la_lane);
    // ... and which of the seven: at 65 536 envs a handful always have ONE joint near a limit, and the launch lasts as
    // long as its slowest wavefront - that wavefront now sweeps the one row, not all seven
    bool la_row[7];
#pragma unroll
    for (int i = 0; i < 7; i++) la_row[i] = la_wave && XARM_ANY(la_sg[i] != (T)0);
    // packed working set of the sweep: joint velocities as 4 pairs + dq[8], full columns of Minv as pairs
    Pk<T> dqp[4], MC[9][4];
    T dq8 = dq[8], ML[9];
#pragma unroll
    for (int k = 0; k < 4; k++) dqp[k] = mkpk<T>(dq[2 * k], dq[2 * k + 1]);
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) MC[i][k] = mkpk<T>(Minv[symi(2 * k, i)], Minv[symi(2 * k + 1, i)]);
        ML[i] = Minv[symi(8, i)];
    }
#define XARM_DQ(i) ((i) == 8 ? dq8 : (((i) & 1) ? pkhi(dqp[(i) >> 1]) : pklo(dqp[(i) >> 1])))
#define XARM_DQ_AXPY(col, dl_) do { _Pragma("unroll") for (int k_ = 0; k_ < 4; k_++) dqp[k_] = pkfma(MC[col][k_], (dl_), dqp[k_]); dq8 += ML[col] * (dl_); } while (0)
    // ---------------- projected Gauss-Seidel, rows in the order T, M, L, G, F
    const T mu_p = (T)xm::MU_OBJECT * (st.mug > (T)0.5 ? (T)xm::MU_FINGER_GRASP : (T)xm::MU_FINGER);
#pragma unroll 1
    for (int it = 0; it < XK_SWEEP_ITERS; it++) {
        XARM_LDS_FENCE(); // keep the S / T reads of this sweep as LDS reads inside the loop
        // (T) object / table points: n = +z, t1 = -y, t2 = +x
#pragma unroll
        for (int s = 0; s < NTS; s++) {
            TablePoint<T> &P = tp[s];
            // no wave-level skip: the slot math is a no-op for inactive lanes (1/diag = 0) and, being independent of the
            // motor rows that follow, fills their dependency stalls when both sit in one basic block
            const T Kxy = lds[LDS_TBL + s * 8 + 0], Kxz = lds[LDS_TBL + s * 8 + 1], Kyy = lds[LDS_TBL + s * 8 + 2],
                    Kyz = lds[LDS_TBL + s * 8 + 3], Kzz = lds[LDS_TBL + s * 8 + 4];
            V3<T> u = vb + cross(wb, P.r);
            T dl = (P.vt - u.z) * lds[LDS_TBL + s * 8 + 5];
            T nl = P.lam[0] + dl;
            nl = smax0(nl);
            dl = nl - P.lam[0];
            P.lam[0] = nl;
            V3<T> fi = mk<T>((T)0, (T)0, dl);
            u = u + mk<T>(Kxz, Kyz, Kzz) * dl;
            const T lim = mu_t * P.lam[0];
            dl = u.y * lds[LDS_TBL + s * 8 + 6]; // t1 = -y: jv = -u.y, target 0
            nl = sclamp(P.lam[1] + dl, -lim, lim);
            dl = nl - P.lam[1];
            P.lam[1] = nl;
            fi.y = -dl;
            u = u - mk<T>(Kxy, Kyy, Kyz) * dl;
            dl = -u.x * lds[LDS_TBL + s * 8 + 7]; // t2 = +x
            nl = sclamp(P.lam[2] + dl, -lim, lim);
            dl = nl - P.lam[2];
            P.lam[2] = nl;
            fi.x = dl;
            vb = vb + fi * imb;
            wb = wb + symmul(Iinv, cross(P.r, fi));
        }
        // (M) velocity-level PD motors
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const T hi = i < 7 ? m_hi_arm : m_hi_fin;
            T dl = (m_vt[i] - XARM_DQ(i)) * m_invd[i];
            const T nl = sclamp(m_lam[i] + dl, -hi, hi);
            dl = nl - m_lam[i];
            m_lam[i] = nl;
            XARM_DQ_AXPY(i, dl);
        }
        // (L) joint limits: arm (one side at most), then fingers (lower, upper)
#pragma unroll
        for (int i = 0; i < 7; i++) {
            if (!la_row[i]) continue;   // wave-uniform, decided once per substep
            const T sg = la_sg[i];
            T dl = (la_vt[i] - sg * XARM_DQ(i)) * (sg != (T)0 ? m_invd[i] : (T)0);
            T nl = la_lam[i] + dl;
            nl = smax0(nl);
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
                nl = smax0(nl);
                dl = (nl - lf_lam[k][side]) * sg;
                lf_lam[k][side] = nl;
                XARM_DQ_AXPY(7 + k, dl);
            }
        // (G) gear row, q7' - q8' = 0
        {
            T dl = (g_vt - (XARM_DQ(7) - dq8)) * g_invd;
            const T nl = sclamp(g_lam + dl, -g_hi, g_hi);
            dl = nl - g_lam;
            g_lam = nl;
            XARM_DQ_AXPY(7, dl);
            XARM_DQ_AXPY(8, -dl);
        }
        // (F) pad points, each solved as a 3x3 block in operational space; with two arms the pads of arm 0 are
        // swept first, the object velocity is handed to the other lane, then the pads of arm 1
#pragma unroll
        for (int ph = 0; ph < Scene::NARMS; ph++) {
        // two arms: sequential (phase = arm) when both touch the object; otherwise the touching arm sweeps in phase 0
        const bool mine = Scene::NARMS == 1 || (seq ? arm == ph : ph == 0);
        if (!FAST && XARM_ANY(pad_any && mine)) {   // NOT a rare path at wave level: ~2 % of the envs hold a finger contact, i.e. 1 - 0.98^64 = 73 % of the wavefronts (a __builtin_expect(.., 0) here took k_step from 2.02 to 3.15 ms)
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
                PadPoint<T> &P1 = pp[2 * fk], &P2 = pp[2 * fk + 1];
                if (!XARM_ANY((P1.invd[0] != (T)0 || P2.invd[0] != (T)0) && mine)) continue;
                const V3<T> af = hc1 * (fk == 0 ? (T)1 : (T)-1);
                const V3<T> yw = mk<T>(y[0], y[1], y[2]);
                const V3<T> base = mk<T>(y[3], y[4], y[5]) + af * yf[fk] - vb;
                V3<T> fsum = mk<T>(0, 0, 0), msum = mk<T>(0, 0, 0), bsum = mk<T>(0, 0, 0); // sum f, sum p x f, sum r x f
                V3<T> f1 = mk<T>(0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    PadPoint<T> &P = j == 0 ? P1 : P2;
                    const T e0 = mine ? P.invd[0] : (T)0, e1 = mine ? P.invd[1] : (T)0, e2 = mine ? P.invd[2] : (T)0;
                    const V3<T> r = P.p - cb;
                    const V3<T> t2 = cross(P.n, P.t1);
                    V3<T> u = base + cross(yw, P.p) - cross(wb, r);
                    if (j == 1) // effect of the impulse just applied at this finger's first point
                        u = u + mk<T>(K21[fk][0] * f1.x + K21[fk][1] * f1.y + K21[fk][2] * f1.z,
                                      K21[fk][3] * f1.x + K21[fk][4] * f1.y + K21[fk][5] * f1.z,
                                      K21[fk][6] * f1.x + K21[fk][7] * f1.y + K21[fk][8] * f1.z);
                    T dl = (P.vt - pad_cfm * P.lam[0] - dot(P.n, u)) * e0;
                    T nl = P.lam[0] + dl;
                    nl = smax0(nl);
                    dl = nl - P.lam[0];
                    P.lam[0] = nl;
                    V3<T> fi = P.n * dl;
                    u = u + P.Kn * dl;
                    const T lim = mu_p * P.lam[0];
                    dl = -dot(P.t1, u) * e1;
                    nl = sclamp(P.lam[1] + dl, -lim, lim);
                    dl = nl - P.lam[1];
                    P.lam[1] = nl;
                    fi = fi + P.t1 * dl;
                    u = u + P.Kt1 * dl;
                    dl = -dot(t2, u) * e2;
                    nl = sclamp(P.lam[2] + dl, -lim, lim);
                    dl = nl - P.lam[2];
                    P.lam[2] = nl;
                    fi = fi + t2 * dl;
                    if (j == 0) f1 = fi;
                    fsum = fsum + fi;
                    msum = msum + cross(P.p, fi);
                    bsum = bsum + cross(r, fi);
                }
                // ONE operational-space update for the finger: +fsum on finger fk (moment msum about the world
                // origin), -fsum on the object (moment bsum about its centre)
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
                vb = vb - fsum * imb;
                wb = wb - symmul(Iinv, bsum);
            }
            XARM_DQ_AXPY(7, wtot[6]);
            XARM_DQ_AXPY(8, wtot[7]);
#pragma unroll
            for (int k = 0; k < 6; k++) {
#pragma unroll
                for (int r2 = 0; r2 < 4; r2++)
                    dqp[r2] = pkfma(mkpk<T>(lds[LDS_T + (2 * r2) * 6 + k], lds[LDS_T + (2 * r2 + 1) * 6 + k]), wtot[k], dqp[r2]);
                dq8 += lds[LDS_T + 8 * 6 + k] * wtot[k];
            }
        }
        if (Scene::NARMS == 2 && !FAST) {
            if (ph == 0) {
                const V3<T> f0 = mk<T>(xchg.from0(vb.x), xchg.from0(vb.y), xchg.from0(vb.z)), g0 = mk<T>(xchg.from0(wb.x), xchg.from0(wb.y), xchg.from0(wb.z));
                const V3<T> pv = mk<T>(xchg.partner(vb.x), xchg.partner(vb.y), xchg.partner(vb.z)), pw = mk<T>(xchg.partner(wb.x), xchg.partner(wb.y), xchg.partner(wb.z));
                // concurrent form: the object is taken from the lane whose pads touched it
                vb = selv(seq, f0, selv(other_any, pv, vb));
                wb = selv(seq, g0, selv(other_any, pw, wb));
            } else {
                const V3<T> f1 = mk<T>(xchg.from1(vb.x), xchg.from1(vb.y), xchg.from1(vb.z)), g1 = mk<T>(xchg.from1(wb.x), xchg.from1(wb.y), xchg.from1(wb.z));
                vb = selv(seq, f1, vb);
                wb = selv(seq, g1, wb);
            }
        }
        }
    }

#pragma unroll
    for (int k = 0; k < 4; k++) { dq[2 * k] = pklo(dqp[k]); dq[2 * k + 1] = pkhi(dqp[k]); }
    dq[8] = dq8;
#undef XARM_DQ
#undef XARM_DQ_AXPY

    // ---------------- store warm-start impulses, integrate (semi-implicit Euler)
#pragma unroll
    for (int i = 0; i < 8; i++) {
        T l = (T)0;
#pragma unroll
        for (int s = 0; s < NTS; s++) l = tp[s].id == i ? tp[s].lam[0] : l;
        st.lam_t[i] = l;
        st.lam_p[i] = (!FAST && i < NP) ? pp[i < NP ? i : 0].lam[0] : (T)0;
    }
#pragma unroll
    for (int i = 0; i < 9; i++) { st.qd[i] = dq[i]; st.q[i] += dt * dq[i]; }
    st.bp[0] += dt * vb.x; st.bp[1] += dt * vb.y; st.bp[2] += dt * vb.z;
    {
        // btTransformUtil::integrateTransform exponential map
        T ang = xsqrt(dot(wb, wb));
        if (ang * dt > (T)0.7853981633974483) ang = (T)0.7853981633974483 * idt;
        T sw, cw;
        xsincos((T)0.5 * ang * dt, sw, cw);
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
    (void)mb;
    return pad_any;
}

// p.stepSimulation() with numSubSteps = 15
template <typename T, typename Lds> XARM_HD void sim_tick(EnvState<T> &st, const T (&qt)[9], Lds lds) {
    const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
#pragma unroll 1
    for (int k = 0; k < xm::PNP_N_SUBSTEPS; k++) substep<T, Lds>(st, qt, dt, lds);
}

// ---------------------------------------------------------------------------------------------
// observation, xarm_pick_and_place.py:220-248; also returns the EEF (link_eef origin) position
template <typename T> XARM_HD void get_obs(const EnvState<T> &st, T (&obs)[OBS_DIM]) {
    Frame<T> f = frame_identity<T>();
    V3<T> w = mk<T>(0, 0, 0), v = mk<T>(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 7; i++) {
        fk_advance(f, i, st.q[i]);
        w = w + f.c2 * st.qd[i];
        v = v + cross(f.o, f.c2) * st.qd[i];
    }
    const V3<T> hp = f.o + f.c0 * (T)xm::HAND_COM[0] + f.c1 * (T)xm::HAND_COM[1] + f.c2 * (T)xm::HAND_COM[2];
    const V3<T> hv = v + cross(w, hp);
    obs[0] = hp.x; obs[1] = hp.y; obs[2] = hp.z;
    obs[3] = hv.x; obs[4] = hv.y; obs[5] = hv.z;
    obs[6] = st.q[7]; obs[7] = st.qd[7];
    obs[8] = st.bp[0]; obs[9] = st.bp[1]; obs[10] = st.bp[2];
    obs[11] = st.bq[0]; obs[12] = st.bq[1]; obs[13] = st.bq[2]; obs[14] = st.bq[3];
    obs[15] = st.bv[0] - hv.x; obs[16] = st.bv[1] - hv.y; obs[17] = st.bv[2] - hv.z;
    obs[18] = st.bw[0]; obs[19] = st.bw[1]; obs[20] = st.bw[2];
    obs[21] = st.bp[0] - hp.x; obs[22] = st.bp[1] - hp.y; obs[23] = st.bp[2] - hp.z;
}

template <typename T> XARM_HD T reward_of(int reward_type, T dist) {
    return reward_type == 0 ? (dist < (T)xm::PNP_DISTANCE_THRESHOLD ? (T)1 : (T)0) : -dist;
}
XARM_HD float xtanh(float x) { return tanhf(x); }
XARM_HD double xtanh(double x) { return tanh(x); }
// staged dense reward, xarm_pick_and_place.py:166-175: reach [0,0.25] / grasped low 0.5 / lifted
// 1 + hover bonus.  if_grasp = both fingers report contact points with the object after the step.
template <typename T> XARM_HD T dense_reward(const EnvState<T> &st, const T (&obs)[OBS_DIM], T d_og) {
    const T gx = obs[0] - (T)xm::PNP_EEF2GRIP[0] - st.bp[0] + (T)0.06;
    const T gy = obs[1] - (T)xm::PNP_EEF2GRIP[1] - st.bp[1];
    const T gz = obs[2] - (T)xm::PNP_EEF2GRIP[2] - st.bp[2];
    const T d_ao = xsqrt(gx * gx + gy * gy + gz * gz);
    if (!(st.touch > (T)0.5)) return (T)0.25 * ((T)1 - xtanh(d_ao));
    if (st.bp[2] > (T)0.05) return (T)1 + (T)0.25 * ((T)1 - xtanh(d_og));
    return (T)0.5;
}

// ---------------------------------------------------------------------------------------------
// sampling: draws 0 init-grasp coin, 1-2 object xy, 3-5 goal xyz, 6 goal-on-ground coin
template <typename T> XARM_HD void sample_draws(const EnvCfg &cfg, int64_t env, int64_t episode, T (&u)[8]) {
    const uint64_t gid = (uint64_t)(cfg.env_id_offset + env);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        uint32_t o[4];
        philox(cfg.seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, (uint32_t)b, o);
#pragma unroll
        for (int k = 0; k < 4; k++) u[b * 4 + k] = u01<T>(o[k]);
    }
}
template <typename T> XARM_HD void sample_object(const EnvCfg &cfg, const T (&u)[8], EnvState<T> &st) {
    if (u[0] < (T)cfg.init_grasp_rate) {
        st.bp[0] = (T)xm::PNP_START_GRIPPER_POS[0];
        st.bp[1] = (T)xm::PNP_START_GRIPPER_POS[1];
    } else {
        st.bp[0] = (T)xm::PNP_OBJ_LOW[0] + u[1] * (T)(xm::PNP_OBJ_HIGH[0] - xm::PNP_OBJ_LOW[0]);
        st.bp[1] = (T)xm::PNP_OBJ_LOW[1] + u[2] * (T)(xm::PNP_OBJ_HIGH[1] - xm::PNP_OBJ_LOW[1]);
    }
    st.bp[2] = (T)xm::PNP_HEIGHT_OFFSET;
    st.bq[0] = st.bq[1] = st.bq[2] = (T)0; st.bq[3] = (T)1;
#pragma unroll
    for (int k = 0; k < 3; k++) { st.bv[k] = (T)0; st.bw[k] = (T)0; }
#pragma unroll
    for (int k = 0; k < 8; k++) { st.lam_t[k] = (T)0; st.lam_p[k] = (T)0; }
}
template <typename T> XARM_HD void sample_goal(const EnvCfg &cfg, const T (&u)[8], EnvState<T> &st) {
#pragma unroll
    for (int k = 0; k < 3; k++) st.goal[k] = (T)xm::PNP_GOAL_LOW[k] + u[3 + k] * (T)(xm::PNP_GOAL_HIGH[k] - xm::PNP_GOAL_LOW[k]);
    if (cfg.goal_shape == 0) {
        if (u[6] < (T)cfg.goal_ground_rate) st.goal[2] = (T)xm::PNP_GOAL_LOW[2];
    } else
        st.goal[2] = (T)xm::PNP_HEIGHT_OFFSET;
}
template <typename T> XARM_HD void env_init(const EnvCfg &cfg, int64_t env, EnvState<T> &st) {
#pragma unroll
    for (int i = 0; i < 9; i++) { st.q[i] = (T)0; st.qd[i] = (T)0; }
    st.touch = st.mug = st.steps = st.episode = (T)0;
    T u[8];
    sample_draws(cfg, env, 0, u);
    sample_object(cfg, u, st);
    sample_goal(cfg, u, st);
}

// XarmPickAndPlace.reset (:121-127) = _reset_sim (:250-267) + _sample_goal (:269-287)
template <typename T, typename Lds> XARM_HD void env_reset(const EnvCfg &cfg, int64_t env, EnvState<T> &st, Lds lds) {
    T qt[9];
    const int64_t episode = (int64_t)st.episode + 1;
    const V3<T> start = mk<T>((T)xm::PNP_START_GRIPPER_POS[0], (T)xm::PNP_START_GRIPPER_POS[1], (T)xm::PNP_START_GRIPPER_POS[2]);
#pragma unroll 1
    for (int k = 0; k < xm::PNP_RESET_TICKS + 1; k++) {
        if (k < xm::PNP_RESET_TICKS) {
            ik_solve(st.q, start, qt);
            qt[7] = qt[8] = (T)xm::PNP_RESET_FINGER_TARGET;
        } else {
            T u[8];
            sample_draws(cfg, env, episode, u);
            sample_object(cfg, u, st);
            sample_goal(cfg, u, st);
        }
        sim_tick<T, Lds>(st, qt, lds);
    }
    st.steps = (T)0;
    st.episode = (T)episode;
}

// XarmPickAndPlace.step (:107-119)
template <typename T, typename Lds>
XARM_HD void env_step(const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&obs)[OBS_DIM], T &reward, bool &done,
                      bool &success, Lds lds) {
    st.steps += (T)1;
    T a[4], qt[9];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    // current link_eef position
    Frame<T> f = frame_identity<T>();
#pragma unroll
    for (int i = 0; i < 7; i++) fk_advance(f, i, st.q[i]);
    const T sc = (T)(xm::PNP_MAX_VEL * xm::PNP_ACTION_DT);
    const V3<T> target = mk<T>(clampT(f.o.x + a[0] * sc, (T)xm::PNP_POS_LOW[0], (T)xm::PNP_POS_HIGH[0]),
                               clampT(f.o.y + a[1] * sc, (T)xm::PNP_POS_LOW[1], (T)xm::PNP_POS_HIGH[1]),
                               clampT(f.o.z + a[2] * sc, (T)xm::PNP_POS_LOW[2], (T)xm::PNP_POS_HIGH[2]));
    const T g = clampT(st.q[7] + a[3] * (T)(xm::PNP_ACTION_DT * xm::PNP_MAX_GRIPPER_VEL), (T)xm::PNP_GRIPPER_LOW, (T)xm::PNP_GRIPPER_HIGH);
    ik_solve(st.q, target, qt);
    qt[7] = qt[8] = g;
    st.mug = st.touch; // friction toggle from the LAST step's contacts (:212-218)
    sim_tick<T, Lds>(st, qt, lds);
    get_obs(st, obs);
    const T dx = st.bp[0] - st.goal[0], dy = st.bp[1] - st.goal[1], dz = st.bp[2] - st.goal[2];
    const T dist = xsqrt(dx * dx + dy * dy + dz * dz);
    success = dist < (T)xm::PNP_DISTANCE_THRESHOLD;
    reward = cfg.reward_type == 2 ? dense_reward<T>(st, obs, dist) : reward_of<T>(cfg.reward_type, dist);
    done = success || ((int)st.steps == xm::PNP_MAX_EPISODE_STEPS);
}

// XarmPickAndPlace.step on the pad-free fast substep.  Returns false when a finger-pad row of this environment was
// active in any of the 15 substeps: the outputs are then meaningless and the caller must not store them (the
// environment is stepped again, from its untouched state, by a kernel that solves pad rows).
template <typename T, typename Lds>
XARM_HD bool env_step_fast(const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&obs)[OBS_DIM], T &reward, bool &done,
                           bool &success, Lds lds) {
    st.steps += (T)1;
    T a[4], qt[9];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    Frame<T> f = frame_identity<T>();
#pragma unroll
    for (int i = 0; i < 7; i++) fk_advance(f, i, st.q[i]);
    const T sc = (T)(xm::PNP_MAX_VEL * xm::PNP_ACTION_DT);
    const V3<T> target = mk<T>(clampT(f.o.x + a[0] * sc, (T)xm::PNP_POS_LOW[0], (T)xm::PNP_POS_HIGH[0]),
                               clampT(f.o.y + a[1] * sc, (T)xm::PNP_POS_LOW[1], (T)xm::PNP_POS_HIGH[1]),
                               clampT(f.o.z + a[2] * sc, (T)xm::PNP_POS_LOW[2], (T)xm::PNP_POS_HIGH[2]));
    const T g = clampT(st.q[7] + a[3] * (T)(xm::PNP_ACTION_DT * xm::PNP_MAX_GRIPPER_VEL), (T)xm::PNP_GRIPPER_LOW, (T)xm::PNP_GRIPPER_HIGH);
    ik_solve(st.q, target, qt);
    qt[7] = qt[8] = g;
    st.mug = st.touch;
    const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
    bool pad = false;
#pragma unroll 1
    for (int k = 0; k < xm::PNP_N_SUBSTEPS; k++) pad = substep<T, Lds, PnpScene, NoXchg, true>(st, qt, dt, lds) || pad;
    get_obs(st, obs);
    const T dx = st.bp[0] - st.goal[0], dy = st.bp[1] - st.goal[1], dz = st.bp[2] - st.goal[2];
    const T dist = xsqrt(dx * dx + dy * dy + dz * dz);
    success = dist < (T)xm::PNP_DISTANCE_THRESHOLD;
    reward = cfg.reward_type == 2 ? dense_reward<T>(st, obs, dist) : reward_of<T>(cfg.reward_type, dist);
    done = success || ((int)st.steps == xm::PNP_MAX_EPISODE_STEPS);
    return !pad;
}

// ---- the STAGED step (xarm_step, xarm_hip.hip; as for Handover, xarm_handover_core.h lane_step_fast_range): the 15 substeps of a
// step in stages.  step_open / step_close are the two ends of env_step above; a stage that begins at substep k0 > 0 continues the
// step with the joint targets qt the opening stage left.
template <typename T> XARM_HD void step_open(EnvState<T> &st, const T (&act)[4], T (&qt)[9]) {
    st.steps += (T)1;
    T a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    Frame<T> f = frame_identity<T>();
#pragma unroll
    for (int i = 0; i < 7; i++) fk_advance(f, i, st.q[i]);
    const T sc = (T)(xm::PNP_MAX_VEL * xm::PNP_ACTION_DT);
    const V3<T> target = mk<T>(clampT(f.o.x + a[0] * sc, (T)xm::PNP_POS_LOW[0], (T)xm::PNP_POS_HIGH[0]),
                               clampT(f.o.y + a[1] * sc, (T)xm::PNP_POS_LOW[1], (T)xm::PNP_POS_HIGH[1]),
                               clampT(f.o.z + a[2] * sc, (T)xm::PNP_POS_LOW[2], (T)xm::PNP_POS_HIGH[2]));
    const T g = clampT(st.q[7] + a[3] * (T)(xm::PNP_ACTION_DT * xm::PNP_MAX_GRIPPER_VEL), (T)xm::PNP_GRIPPER_LOW, (T)xm::PNP_GRIPPER_HIGH);
    ik_solve(st.q, target, qt);
    qt[7] = qt[8] = g;
    st.mug = st.touch;
}
template <typename T> XARM_HD void step_close(const EnvCfg &cfg, EnvState<T> &st, T (&obs)[OBS_DIM], T &reward, bool &done, bool &success) {
    get_obs(st, obs);
    const T dx = st.bp[0] - st.goal[0], dy = st.bp[1] - st.goal[1], dz = st.bp[2] - st.goal[2];
    const T dist = xsqrt(dx * dx + dy * dy + dz * dz);
    success = dist < (T)xm::PNP_DISTANCE_THRESHOLD;
    reward = cfg.reward_type == 2 ? dense_reward<T>(st, obs, dist) : reward_of<T>(cfg.reward_type, dist);
    done = success || ((int)st.steps == xm::PNP_MAX_EPISODE_STEPS);
}
// substeps [k0, k1) on the pad-free fast substep; false: a pad row was active in one of them (store nothing)
template <typename T, typename Lds>
XARM_HD bool env_step_fast_range(const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&qt)[9], int k0, int k1, T (&obs)[OBS_DIM], T &reward,
                                 bool &done, bool &success, Lds lds) {
    if (k0 == 0) step_open(st, act, qt);
    const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
    bool pad = false;
#pragma unroll 1
    for (int k = k0; k < k1; k++) pad = substep<T, Lds, PnpScene, NoXchg, true>(st, qt, dt, lds) || pad;
    if (k1 == xm::PNP_N_SUBSTEPS) step_close(cfg, st, obs, reward, done, success);
    return !pad;
}
// substeps [k0, 15) on the full substep (the long-list fall-back of a staged hand-off)
template <typename T, typename Lds>
XARM_HD void env_step_from(const EnvCfg &cfg, EnvState<T> &st, const T (&act)[4], T (&qt)[9], int k0, T (&obs)[OBS_DIM], T &reward, bool &done,
                           bool &success, Lds lds) {
    if (k0 == 0) step_open(st, act, qt);
    const T dt = (T)(xm::PNP_TIME_STEP / xm::PNP_N_SUBSTEPS);
#pragma unroll 1
    for (int k = k0; k < xm::PNP_N_SUBSTEPS; k++) substep<T, Lds>(st, qt, dt, lds);
    step_close(cfg, st, obs, reward, done, success);
}

// Lazy auto-reset (opt-in, NOT the reference's VecEnv semantics): instead of running the reference's
// PNP_RESET_TICKS + 1 reset ticks inside the step call in which an env finishes - six sequential ticks of latency for a
// handful of envs while the rest of the GPU idles - a finished env spends its next six step calls on
// those same ticks, one per call, next to the ordinary steps of all other envs.  The tick sequence and therefore the
// state after the sixth call are exactly those of env_reset.  While resetting, `st.steps` holds -(ticks still to
// run) and the env's action is ignored; phase = 1 marks the call in which the episode ended, 2 a reset tick
// (transition to be masked by the learner; the sixth returns the first observation of the new episode).
// The pose target of tick 5 is reused by the teleport tick (env_reset keeps it in a local); it is carried across the
// two calls in lam_p[4..7] and goal[0..2], which are dead during a reset.
template <typename T, typename Lds>
XARM_HD void env_step_lazy(const EnvCfg &cfg, int64_t env, EnvState<T> &st, const T (&act)[4], T (&obs)[OBS_DIM], T &reward,
                           bool &done, bool &success, int &phase, Lds lds) {
    const bool resetting = st.steps < (T)0;
    const int left = resetting ? -(int)st.steps : 0;          // reset ticks still to run, 6 .. 1
    const bool teleport = resetting && left == 1;
    T a[4], qt[9];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = clampT(act[k], (T)-1, (T)1);
    Frame<T> f = frame_identity<T>();
#pragma unroll
    for (int i = 0; i < 7; i++) fk_advance(f, i, st.q[i]);
    const T sc = (T)(xm::PNP_MAX_VEL * xm::PNP_ACTION_DT);
    const V3<T> tstep = mk<T>(clampT(f.o.x + a[0] * sc, (T)xm::PNP_POS_LOW[0], (T)xm::PNP_POS_HIGH[0]),
                              clampT(f.o.y + a[1] * sc, (T)xm::PNP_POS_LOW[1], (T)xm::PNP_POS_HIGH[1]),
                              clampT(f.o.z + a[2] * sc, (T)xm::PNP_POS_LOW[2], (T)xm::PNP_POS_HIGH[2]));
    const V3<T> start = mk<T>((T)xm::PNP_START_GRIPPER_POS[0], (T)xm::PNP_START_GRIPPER_POS[1], (T)xm::PNP_START_GRIPPER_POS[2]);
    ik_solve(st.q, resetting ? start : tstep, qt);
    const T g = clampT(st.q[7] + a[3] * (T)(xm::PNP_ACTION_DT * xm::PNP_MAX_GRIPPER_VEL), (T)xm::PNP_GRIPPER_LOW, (T)xm::PNP_GRIPPER_HIGH);
    qt[7] = qt[8] = resetting ? (T)xm::PNP_RESET_FINGER_TARGET : g;
    if (teleport) {
        // the pose target stays the one of the previous tick; then respawn the object and draw the goal (:259-266, :124)
#pragma unroll
        for (int k = 0; k < 4; k++) qt[k] = st.lam_p[4 + k];
#pragma unroll
        for (int k = 0; k < 3; k++) qt[4 + k] = st.goal[k];
#pragma unroll
        for (int k = 0; k < 4; k++) st.lam_p[4 + k] = (T)0;
        T u[8];
        sample_draws(cfg, env, (int64_t)st.episode + 1, u);
        sample_object(cfg, u, st);
        sample_goal(cfg, u, st);
    } else if (!resetting) {
        st.steps += (T)1;
        st.mug = st.touch; // friction toggle from the LAST step's contacts (:212-218)
    }
    sim_tick<T, Lds>(st, qt, lds);
    if (resetting && !teleport) {
        // stash this tick's pose target for the teleport tick (written after the tick: substep clears lam_p[4..7])
#pragma unroll
        for (int k = 0; k < 4; k++) st.lam_p[4 + k] = qt[k];
#pragma unroll
        for (int k = 0; k < 3; k++) st.goal[k] = qt[4 + k];
    }
    get_obs(st, obs);
    const T dx = st.bp[0] - st.goal[0], dy = st.bp[1] - st.goal[1], dz = st.bp[2] - st.goal[2];
    const T dist = xsqrt(dx * dx + dy * dy + dz * dz);
    if (resetting) {
        success = false; done = false; reward = (T)0; phase = 2;
        st.steps = teleport ? (T)0 : (T)(-(left - 1));
        if (teleport) st.episode += (T)1;
    } else {
        success = dist < (T)xm::PNP_DISTANCE_THRESHOLD;
        reward = cfg.reward_type == 2 ? dense_reward<T>(st, obs, dist) : reward_of<T>(cfg.reward_type, dist);
        done = success || ((int)st.steps == xm::PNP_MAX_EPISODE_STEPS);
        phase = done ? 1 : 0;
        if (done) st.steps = (T)(-(xm::PNP_RESET_TICKS + 1));
    }
}

} // namespace xk
