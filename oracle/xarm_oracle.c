/*
 * xarm_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see xarm_oracle.h for the parity status).
 *
 * Restates, in double precision and in the most literal form available, the reference hot path
 *   XarmPickAndPlace.step        /root/reference/gym_xarm/envs/xarm_pick_and_place.py:107-119
 *   XarmPickAndPlace._set_action                                              :199-218
 *   XarmPickAndPlace._get_obs                                                 :220-248
 *   XarmPickAndPlace.reset/_reset_sim/_sample_goal                            :121-127,250-287
 *   XarmPickAndPlace.compute_reward/_is_success                               :155-177,289-291
 * and the PyBullet calls they make.  The physics is organised the way Bullet's multibody world
 * is (generic kinematic tree, ABA in link coordinates, one Jacobian + one unit-impulse response
 * per solver row, sequential PGS sweep) — deliberately NOT the way the HIP kernel is organised
 * (world-frame CRBA + Cholesky, operational-space block solver), so that agreement between the
 * two is an independent check.
 */
#include "xarm_oracle.h"
#include <math.h>
#include <string.h>

typedef double real;

/* ------------------------------------------------------------------ small linear algebra */
static void v3_set(real *o, real x, real y, real z) { o[0] = x; o[1] = y; o[2] = z; }
static void v3_copy(real *o, const real *a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static real v3_dot(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void v3_cross(real *o, const real *a, const real *b) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static void v3_sub(real *o, const real *a, const real *b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void v3_add(real *o, const real *a, const real *b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static void v3_axpy(real *o, real s, const real *a) { o[0] += s * a[0]; o[1] += s * a[1]; o[2] += s * a[2]; }
static real v3_norm(const real *a) { return sqrt(v3_dot(a, a)); }
/* 3x3 row-major */
static void m3_mul(real *o, const real *a, const real *b) {
    real t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    memcpy(o, t, sizeof t);
}
static void m3_vec(real *o, const real *a, const real *v) {
    real x = a[0] * v[0] + a[1] * v[1] + a[2] * v[2];
    real y = a[3] * v[0] + a[4] * v[1] + a[5] * v[2];
    real z = a[6] * v[0] + a[7] * v[1] + a[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void m3_tvec(real *o, const real *a, const real *v) { /* a^T v */
    real x = a[0] * v[0] + a[3] * v[1] + a[6] * v[2];
    real y = a[1] * v[0] + a[4] * v[1] + a[7] * v[2];
    real z = a[2] * v[0] + a[5] * v[1] + a[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void m3_transpose(real *o, const real *a) {
    real t[9] = {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]};
    memcpy(o, t, sizeof t);
}
static void m3_ident(real *o) { memset(o, 0, 9 * sizeof(real)); o[0] = o[4] = o[8] = 1; }
/* URDF rpy: fixed-axis XYZ, R = Rz(y) Ry(p) Rx(r) */
static void m3_from_rpy(real *o, const real *rpy) {
    real cr = cos(rpy[0]), sr = sin(rpy[0]), cp = cos(rpy[1]), sp = sin(rpy[1]), cy = cos(rpy[2]), sy = sin(rpy[2]);
    o[0] = cy * cp; o[1] = cy * sp * sr - sy * cr; o[2] = cy * sp * cr + sy * sr;
    o[3] = sy * cp; o[4] = sy * sp * sr + cy * cr; o[5] = sy * sp * cr - cy * sr;
    o[6] = -sp;     o[7] = cp * sr;                o[8] = cp * cr;
}
static void m3_axis_angle(real *o, const real *k, real th) { /* Rodrigues, unit k */
    real c = cos(th), s = sin(th), v = 1 - c;
    o[0] = k[0] * k[0] * v + c;        o[1] = k[0] * k[1] * v - k[2] * s; o[2] = k[0] * k[2] * v + k[1] * s;
    o[3] = k[0] * k[1] * v + k[2] * s; o[4] = k[1] * k[1] * v + c;        o[5] = k[1] * k[2] * v - k[0] * s;
    o[6] = k[0] * k[2] * v - k[1] * s; o[7] = k[1] * k[2] * v + k[0] * s; o[8] = k[2] * k[2] * v + c;
}
static void m3_skew(real *o, const real *v) {
    o[0] = 0; o[1] = -v[2]; o[2] = v[1];
    o[3] = v[2]; o[4] = 0; o[5] = -v[0];
    o[6] = -v[1]; o[7] = v[0]; o[8] = 0;
}
static void quat_to_m3(real *o, const real *q) { /* q = x y z w */
    real x = q[0], y = q[1], z = q[2], w = q[3];
    o[0] = 1 - 2 * (y * y + z * z); o[1] = 2 * (x * y - z * w);     o[2] = 2 * (x * z + y * w);
    o[3] = 2 * (x * y + z * w);     o[4] = 1 - 2 * (x * x + z * z); o[5] = 2 * (y * z - x * w);
    o[6] = 2 * (x * z - y * w);     o[7] = 2 * (y * z + x * w);     o[8] = 1 - 2 * (x * x + y * y);
}
/* 6x6 row-major */
static void m6_mul(real *o, const real *a, const real *b) {
    real t[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            real s = 0;
            for (int k = 0; k < 6; k++) s += a[i * 6 + k] * b[k * 6 + j];
            t[i * 6 + j] = s;
        }
    memcpy(o, t, sizeof t);
}
static void m6_tmul(real *o, const real *a, const real *b) { /* a^T b */
    real t[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            real s = 0;
            for (int k = 0; k < 6; k++) s += a[k * 6 + i] * b[k * 6 + j];
            t[i * 6 + j] = s;
        }
    memcpy(o, t, sizeof t);
}
static void m6_vec(real *o, const real *a, const real *v) {
    real t[6];
    for (int i = 0; i < 6; i++) {
        real s = 0;
        for (int k = 0; k < 6; k++) s += a[i * 6 + k] * v[k];
        t[i] = s;
    }
    memcpy(o, t, sizeof t);
}
static void m6_tvec(real *o, const real *a, const real *v) {
    real t[6];
    for (int i = 0; i < 6; i++) {
        real s = 0;
        for (int k = 0; k < 6; k++) s += a[k * 6 + i] * v[k];
        t[i] = s;
    }
    memcpy(o, t, sizeof t);
}
static real v6_dot(const real *a, const real *b) {
    real s = 0;
    for (int i = 0; i < 6; i++) s += a[i] * b[i];
    return s;
}
/* spatial cross products, vectors are (angular; linear) */
static void crm(real *o, const real *v, const real *m) { /* v x m (motion) */
    real a[3], b[3], c[3];
    v3_cross(a, v, m);
    v3_cross(b, v, m + 3);
    v3_cross(c, v + 3, m);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
    o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(real *o, const real *v, const real *f) { /* v x* f (force) */
    real a[3], b[3], c[3];
    v3_cross(a, v, f);
    v3_cross(b, v + 3, f + 3);
    v3_cross(c, v, f + 3);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
    o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}

/* ------------------------------------------------------------------ Philox4x32-10 */
void xo_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static real u01(uint32_t x) { return (real)(x >> 8) * (1.0 / 16777216.0); }

/* ------------------------------------------------------------------ kinematic tree */
typedef struct {
    int nd;
    int dof[XO_MAXL];      /* dof index of link's joint or -1 */
    real R[XO_MAXL][9];    /* world rotation of link frame */
    real o[XO_MAXL][3];    /* world origin of link frame */
    real a[XO_MAXL][3];    /* world joint axis */
    real X[XO_MAXL][36];   /* motion transform parent->link */
    real S[XO_MAXL][6];    /* joint motion subspace, link coords */
    real I[XO_MAXL][36];   /* link spatial inertia, link coords */
    real IA[XO_MAXL][36], U[XO_MAXL][6], D[XO_MAXL]; /* ABA cache (depends on q only) */
    real Rbase[9], pbase[3]; /* fixed base pose in the world (identity for the single-arm envs) */
} tree_t;

static void tree_setup_base(const xo_model *m, const real *q, tree_t *t, const real *Rbase, const real *pbase) {
    int nd = 0;
    if (Rbase) { memcpy(t->Rbase, Rbase, 9 * sizeof(real)); v3_copy(t->pbase, pbase); }
    else { m3_ident(t->Rbase); v3_set(t->pbase, 0, 0, 0); }
    for (int i = 0; i < m->n_links; i++) t->dof[i] = (m->jtype[i] != 0) ? nd++ : -1;
    t->nd = nd;
    for (int i = 0; i < m->n_links; i++) {
        real Rorg[9], Rj[9], Rpc[9], r[3], E[9];
        m3_from_rpy(Rorg, m->org_rpy[i]);
        v3_copy(r, m->org_p[i]);
        m3_ident(Rj);
        memset(t->S[i], 0, sizeof t->S[i]);
        if (m->jtype[i] == 1) {
            m3_axis_angle(Rj, m->axis[i], q[t->dof[i]]);
            v3_copy(t->S[i], m->axis[i]);
        } else if (m->jtype[i] == 2) {
            real ax[3];
            m3_vec(ax, Rorg, m->axis[i]);
            v3_axpy(r, q[t->dof[i]], ax);
            v3_copy(t->S[i] + 3, m->axis[i]);
        }
        m3_mul(Rpc, Rorg, Rj);
        m3_transpose(E, Rpc);
        /* X = [[E,0],[-E rx, E]] */
        real rx[9], Erx[9];
        m3_skew(rx, r);
        m3_mul(Erx, E, rx);
        memset(t->X[i], 0, sizeof t->X[i]);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                t->X[i][a * 6 + b] = E[a * 3 + b];
                t->X[i][(a + 3) * 6 + b + 3] = E[a * 3 + b];
                t->X[i][(a + 3) * 6 + b] = -Erx[a * 3 + b];
            }
        /* world frames */
        int p = m->parent[i];
        if (p < 0) {
            real rw[3];
            m3_mul(t->R[i], t->Rbase, Rpc);
            m3_vec(rw, t->Rbase, r);
            v3_add(t->o[i], t->pbase, rw);
        } else {
            m3_mul(t->R[i], t->R[p], Rpc);
            real rw[3];
            m3_vec(rw, t->R[p], r);
            v3_add(t->o[i], t->o[p], rw);
        }
        m3_vec(t->a[i], t->R[i], m->axis[i]);
        /* spatial inertia about link origin */
        real cx[9], cxT[9], cc[9];
        const real *in = m->inertia[i];
        real Ic[9] = {in[0], in[1], in[2], in[1], in[3], in[4], in[2], in[4], in[5]};
        real mass = m->mass[i];
        m3_skew(cx, m->com[i]);
        m3_transpose(cxT, cx);
        m3_mul(cc, cx, cxT);
        memset(t->I[i], 0, sizeof t->I[i]);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                t->I[i][a * 6 + b] = Ic[a * 3 + b] + mass * cc[a * 3 + b];
                t->I[i][a * 6 + b + 3] = mass * cx[a * 3 + b];
                t->I[i][(a + 3) * 6 + b] = mass * cxT[a * 3 + b];
            }
        for (int a = 0; a < 3; a++) t->I[i][(a + 3) * 6 + a + 3] = mass;
    }
}

static void tree_setup(const xo_model *m, const real *q, tree_t *t) { tree_setup_base(m, q, t, 0, 0); }

/* Featherstone articulated-body algorithm (RBDA Table 7.1), fixed base, link coordinates.
 * Restates btMultiBody::computeAccelerationsArticulatedBodyAlgorithmMultiDof. */
static void aba_forward_dynamics(const xo_model *m, tree_t *t, const real *qd, const real *tau, real g, real *qdd) {
    int n = m->n_links;
    real v[XO_MAXL][6], c[XO_MAXL][6], pA[XO_MAXL][6], u[XO_MAXL], a[XO_MAXL][6];
    for (int i = 0; i < n; i++) {
        real vJ[6];
        real qdi = t->dof[i] >= 0 ? qd[t->dof[i]] : 0;
        for (int k = 0; k < 6; k++) vJ[k] = t->S[i][k] * qdi;
        int p = m->parent[i];
        if (p < 0) {
            memcpy(v[i], vJ, sizeof vJ);
            memset(c[i], 0, sizeof c[i]);
        } else {
            m6_vec(v[i], t->X[i], v[p]);
            for (int k = 0; k < 6; k++) v[i][k] += vJ[k];
            crm(c[i], v[i], vJ);
        }
        memcpy(t->IA[i], t->I[i], sizeof t->IA[i]);
        real Iv[6];
        m6_vec(Iv, t->I[i], v[i]);
        crf(pA[i], v[i], Iv);
    }
    for (int i = n - 1; i >= 0; i--) {
        real Ia[36], pa[6], Ic6[6];
        int p = m->parent[i];
        if (t->dof[i] >= 0) {
            m6_vec(t->U[i], t->IA[i], t->S[i]);
            t->D[i] = v6_dot(t->S[i], t->U[i]);
            u[i] = tau[t->dof[i]] - v6_dot(t->S[i], pA[i]);
            for (int r = 0; r < 6; r++)
                for (int s = 0; s < 6; s++) Ia[r * 6 + s] = t->IA[i][r * 6 + s] - t->U[i][r] * t->U[i][s] / t->D[i];
            m6_vec(Ic6, Ia, c[i]);
            for (int k = 0; k < 6; k++) pa[k] = pA[i][k] + Ic6[k] + t->U[i][k] * (u[i] / t->D[i]);
        } else {
            memcpy(Ia, t->IA[i], sizeof Ia);
            m6_vec(Ic6, Ia, c[i]);
            for (int k = 0; k < 6; k++) pa[k] = pA[i][k] + Ic6[k];
        }
        if (p >= 0) {
            real T[36], XtIX[36], Xtp[6];
            m6_mul(T, Ia, t->X[i]);
            m6_tmul(XtIX, t->X[i], T);
            for (int k = 0; k < 36; k++) t->IA[p][k] += XtIX[k];
            m6_tvec(Xtp, t->X[i], pa);
            for (int k = 0; k < 6; k++) pA[p][k] += Xtp[k];
        }
    }
    real a0[6] = {0, 0, 0, 0, 0, 0}, gw[3] = {0, 0, g}; /* base accelerates upward = gravity pulls down */
    m3_tvec(a0 + 3, t->Rbase, gw);                     /* expressed in base axes */
    for (int i = 0; i < n; i++) {
        int p = m->parent[i];
        real ap[6];
        m6_vec(ap, t->X[i], p < 0 ? a0 : a[p]);
        for (int k = 0; k < 6; k++) ap[k] += c[i][k];
        if (t->dof[i] >= 0) {
            real qddi = (u[i] - v6_dot(t->U[i], ap)) / t->D[i];
            qdd[t->dof[i]] = qddi;
            for (int k = 0; k < 6; k++) a[i][k] = ap[k] + t->S[i][k] * qddi;
        } else
            memcpy(a[i], ap, sizeof ap);
    }
}

/* unit-impulse response dqd = M^-1 imp using the cached IA/U/D
 * (restates btMultiBody::calcAccelerationDeltasMultiDof) */
static void aba_impulse_response(const xo_model *m, const tree_t *t, const real *imp, real *dqd) {
    int n = m->n_links;
    real pA[XO_MAXL][6], u[XO_MAXL], a[XO_MAXL][6];
    memset(pA, 0, sizeof pA);
    for (int i = n - 1; i >= 0; i--) {
        real pa[6];
        int p = m->parent[i];
        if (t->dof[i] >= 0) {
            u[i] = imp[t->dof[i]] - v6_dot(t->S[i], pA[i]);
            for (int k = 0; k < 6; k++) pa[k] = pA[i][k] + t->U[i][k] * (u[i] / t->D[i]);
        } else
            memcpy(pa, pA[i], sizeof pa);
        if (p >= 0) {
            real Xtp[6];
            m6_tvec(Xtp, t->X[i], pa);
            for (int k = 0; k < 6; k++) pA[p][k] += Xtp[k];
        }
    }
    real a0[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        int p = m->parent[i];
        real ap[6];
        m6_vec(ap, t->X[i], p < 0 ? a0 : a[p]);
        if (t->dof[i] >= 0) {
            real d = (u[i] - v6_dot(t->U[i], ap)) / t->D[i];
            dqd[t->dof[i]] = d;
            for (int k = 0; k < 6; k++) a[i][k] = ap[k] + t->S[i][k] * d;
        } else
            memcpy(a[i], ap, sizeof ap);
    }
}

/* is link `anc` an ancestor-or-self of link `l`? */
static int is_ancestor(const xo_model *m, int anc, int l) {
    while (l >= 0) {
        if (l == anc) return 1;
        l = m->parent[l];
    }
    return 0;
}
/* joint-space row of "direction d . velocity of the point p fixed to link l" */
static void point_jacobian_row(const xo_model *m, const tree_t *t, int l, const real *p, const real *d, real *J) {
    memset(J, 0, XO_MAXD * sizeof(real));
    for (int j = 0; j < m->n_links; j++) {
        if (t->dof[j] < 0 || !is_ancestor(m, j, l)) continue;
        if (m->jtype[j] == 1) {
            real r[3], c[3];
            v3_sub(r, p, t->o[j]);
            v3_cross(c, t->a[j], r);
            J[t->dof[j]] = v3_dot(d, c);
        } else
            J[t->dof[j]] = v3_dot(d, t->a[j]);
    }
}

/* ------------------------------------------------------------------ inverse kinematics
 * Damped-least-squares position+orientation IK for link `eef_link`, seeded from q, target
 * orientation quaternion (x,y,z,w) = (1,0,0,0) (xarm_pick_and_place.py:207,253).  Restates the
 * BussIK DLS step PyBullet uses for calculateInverseKinematics with an orientation target. */
static void rot_error_vec(const real *Rt, const real *Rc, real *e) {
    real RcT[9], Re[9];
    m3_transpose(RcT, Rc);
    m3_mul(Re, Rt, RcT);
    real vee[3] = {0.5 * (Re[7] - Re[5]), 0.5 * (Re[2] - Re[6]), 0.5 * (Re[3] - Re[1])};
    real s = v3_norm(vee), c = 0.5 * (Re[0] + Re[4] + Re[8] - 1);
    if (s > 1e-6) {
        real k = atan2(s, c) / s;
        v3_set(e, vee[0] * k, vee[1] * k, vee[2] * k);
    } else if (c > 0) {
        v3_copy(e, vee);
    } else { /* angle ~ pi: axis from the diagonal, largest component positive */
        real ax[3];
        for (int i = 0; i < 3; i++) {
            real d = 0.5 * (Re[i * 4] + 1);
            ax[i] = sqrt(d > 0 ? d : 0);
        }
        int k = 0;
        if (ax[1] > ax[k]) k = 1;
        if (ax[2] > ax[k]) k = 2;
        for (int i = 0; i < 3; i++)
            if (i != k && (Re[k * 3 + i] + Re[i * 3 + k]) < 0) ax[i] = -ax[i];
        real pi = 3.14159265358979323846;
        v3_set(e, pi * ax[0], pi * ax[1], pi * ax[2]);
    }
}

static void ik_solve_base(const xo_model *m, const real *q_in, const real *target, int max_iter, real *q_out,
                          const real *Rbase, const real *pbase) {
    real q[XO_MAXD];
    tree_t t;
    const real Rt[9] = {1, 0, 0, 0, -1, 0, 0, 0, -1}; /* quaternion (1,0,0,0), world frame */
    int e = m->eef_link;
    tree_setup_base(m, q_in, &t, Rbase, pbase);
    memcpy(q, q_in, t.nd * sizeof(real));
    for (int it = 0; it < max_iter; it++) {
        if (it > 0) tree_setup_base(m, q, &t, Rbase, pbase);
        real err[6];
        v3_sub(err, target, t.o[e]);
        if (v3_norm(err) < m->ik_residual) break;
        rot_error_vec(Rt, t.R[e], err + 3);
        /* Jacobian 6 x nd over the revolute ancestors of the eef link */
        real J[6][XO_MAXD];
        memset(J, 0, sizeof J);
        for (int j = 0; j < m->n_links; j++) {
            if (t.dof[j] < 0 || !is_ancestor(m, j, e)) continue;
            int dj = t.dof[j];
            if (m->jtype[j] == 1) {
                real r[3], c[3];
                v3_sub(r, t.o[e], t.o[j]);
                v3_cross(c, t.a[j], r);
                for (int k = 0; k < 3; k++) { J[k][dj] = c[k]; J[k + 3][dj] = t.a[j][k]; }
            } else
                for (int k = 0; k < 3; k++) J[k][dj] = t.a[j][k];
        }
        /* A = J J^T + lambda^2 I ; solve A x = err by Cholesky */
        real A[6][6], L[6][6], x[6], y[6];
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) {
                real s = 0;
                for (int k = 0; k < t.nd; k++) s += J[r][k] * J[c][k];
                A[r][c] = s + (r == c ? m->ik_lambda * m->ik_lambda : 0);
            }
        memset(L, 0, sizeof L);
        for (int r = 0; r < 6; r++)
            for (int c = 0; c <= r; c++) {
                real s = A[r][c];
                for (int k = 0; k < c; k++) s -= L[r][k] * L[c][k];
                L[r][c] = (r == c) ? sqrt(s) : s / L[c][c];
            }
        for (int r = 0; r < 6; r++) {
            real s = err[r];
            for (int k = 0; k < r; k++) s -= L[r][k] * y[k];
            y[r] = s / L[r][r];
        }
        for (int r = 5; r >= 0; r--) {
            real s = y[r];
            for (int k = r + 1; k < 6; k++) s -= L[k][r] * x[k];
            x[r] = s / L[r][r];
        }
        real dq[XO_MAXD], mx = 0;
        for (int k = 0; k < t.nd; k++) {
            real s = 0;
            for (int r = 0; r < 6; r++) s += J[r][k] * x[r];
            dq[k] = s;
            if (fabs(s) > mx) mx = fabs(s);
        }
        real sc = mx > m->ik_max_dtheta ? m->ik_max_dtheta / mx : 1.0;
        for (int k = 0; k < t.nd; k++) q[k] += sc * dq[k];
    }
    memcpy(q_out, q, t.nd * sizeof(real));
}
static void ik_solve(const xo_model *m, const real *q_in, const real *target, int max_iter, real *q_out) {
    ik_solve_base(m, q_in, target, max_iter, q_out, 0, 0);
}

static int dof_of_link(const xo_model *m, int link) {
    int nd = 0;
    for (int i = 0; i < link; i++) nd += (m->jtype[i] != 0);
    return nd;
}

/* ------------------------------------------------------------------ state row access */
enum { S_Q = 0, S_QD = 9, S_BP = 18, S_BQ = 21, S_BV = 25, S_BW = 28, S_GOAL = 31, S_LT = 34, S_LP = 42,
       S_TOUCH = 50, S_MUG = 51, S_STEPS = 52, S_EPISODE = 53 };

/* ------------------------------------------------------------------ solver rows */
typedef struct {
    int has_a, has_b;
    real Ja[XO_MAXD], Ba[XO_MAXD], Jb[6], Bb[6];
    real vt, cfm, inv_d, lo, hi, lam;
    int normal_row; /* index of the normal row whose impulse bounds this friction row, or -1 */
    real mu;
    int arm;        /* which arm the joint-space part belongs to (dual-arm envs), 0 otherwise */
} row_t;
#define XO_MAXROWS 96

/* btPlaneSpace1 */
static void plane_space(const real *n, real *p, real *q) {
    if (fabs(n[2]) > 0.7071067811865475244) {
        real a = n[1] * n[1] + n[2] * n[2], k = 1.0 / sqrt(a);
        v3_set(p, 0, -n[2] * k, n[1] * k);
        v3_set(q, a * k, -n[0] * p[2], n[0] * p[1]);
    } else {
        real a = n[0] * n[0] + n[1] * n[1], k = 1.0 / sqrt(a);
        v3_set(p, -n[1] * k, n[0] * k, 0);
        v3_set(q, -n[2] * p[1], n[2] * p[0], a * k);
    }
}

typedef struct {
    const xo_model *m;
    tree_t t, t2; /* t2: second arm of the dual-arm envs */
    real Rb[9], Iinv_w[9];
    row_t rows[XO_MAXROWS];
    int nrows;
} solver_t;

static void row_finish(solver_t *s, row_t *r) {
    real d = 0;
    if (r->has_a) {
        const tree_t *t = r->arm ? &s->t2 : &s->t;
        aba_impulse_response(s->m, t, r->Ja, r->Ba);
        for (int k = 0; k < t->nd; k++) d += r->Ja[k] * r->Ba[k];
    }
    if (r->has_b) {
        real w[3];
        for (int k = 0; k < 3; k++) r->Bb[k] = r->Jb[k] / s->m->obj_mass;
        m3_vec(w, s->Iinv_w, r->Jb + 3);
        v3_copy(r->Bb + 3, w);
        for (int k = 0; k < 6; k++) d += r->Jb[k] * r->Bb[k];
    }
    r->inv_d = 1.0 / (d + r->cfm);
}
static row_t *row_new(solver_t *s) {
    row_t *r = &s->rows[s->nrows++];
    memset(r, 0, sizeof *r);
    r->normal_row = -1;
    return r;
}
/* three rows (normal, 2 x friction) of one contact point.  The normal points from body B to
 * body A; link >= 0: A is that arm link and B the object; link < 0: A is the object, B static. */
static int add_contact_arm(solver_t *s, int arm, int link, const real *p, const real *n, real dist, real dt, real erp,
                           real cfm, real mu, real lam0, const real *cb) {
    real t1[3], t2[3], r[3];
    plane_space(n, t1, t2);
    v3_sub(r, p, cb);
    const real *dirs[3] = {n, t1, t2};
    int nrow = s->nrows;
    for (int k = 0; k < 3; k++) {
        row_t *row = row_new(s);
        real rxd[3];
        v3_cross(rxd, r, dirs[k]);
        real sgn = (link >= 0) ? -1.0 : 1.0;
        row->has_b = 1;
        for (int c = 0; c < 3; c++) { row->Jb[c] = sgn * dirs[k][c]; row->Jb[c + 3] = sgn * rxd[c]; }
        if (link >= 0) {
            row->has_a = 1;
            row->arm = arm;
            point_jacobian_row(s->m, arm ? &s->t2 : &s->t, link, p, dirs[k], row->Ja);
        }
        if (k == 0) {
            row->vt = dist < 0 ? -erp * dist / dt : -dist / dt;
            row->cfm = cfm;
            row->lo = 0; row->hi = 1e30;
            row->lam = lam0;
        } else {
            row->normal_row = nrow;
            row->mu = mu;
        }
        row_finish(s, row);
    }
    return nrow;
}

static int add_contact(solver_t *s, int link, const real *p, const real *n, real dist, real dt, real erp,
                       real cfm, real mu, real lam0, const real *cb) {
    return add_contact_arm(s, 0, link, p, n, dist, dt, erp, cfm, mu, lam0, cb);
}

static void apply_row_impulse(const row_t *r, real dl, real *qd, real *vb, int nd) {
    if (r->has_a)
        for (int k = 0; k < nd; k++) qd[k] += r->Ba[k] * dl;
    if (r->has_b)
        for (int k = 0; k < 6; k++) vb[k] += r->Bb[k] * dl;
}

/* sphere (centre c, radius rho) against the object box */
static int sphere_box(const real *c, real rho, const real *cb, const real *Rb, const real *h, real margin,
                      real *dist, real *n, real *p) {
    real d[3], cl[3], ql[3], dl[3], nl[3], pl[3];
    v3_sub(d, c, cb);
    m3_tvec(cl, Rb, d);
    for (int i = 0; i < 3; i++) ql[i] = cl[i] < -h[i] ? -h[i] : (cl[i] > h[i] ? h[i] : cl[i]);
    v3_sub(dl, cl, ql);
    real d2 = v3_dot(dl, dl);
    if (d2 > 1e-12) {
        real len = sqrt(d2);
        v3_set(nl, dl[0] / len, dl[1] / len, dl[2] / len);
        *dist = len - rho;
        v3_copy(pl, ql);
    } else {
        int k = 0;
        real best = h[0] - fabs(cl[0]);
        for (int i = 1; i < 3; i++) {
            real pen = h[i] - fabs(cl[i]);
            if (pen < best) { best = pen; k = i; }
        }
        real sg = cl[k] < 0 ? -1.0 : 1.0;
        v3_set(nl, 0, 0, 0);
        nl[k] = sg;
        *dist = -best - rho;
        v3_copy(pl, cl);
        pl[k] = sg * h[k];
    }
    m3_vec(n, Rb, nl);
    m3_vec(p, Rb, pl);
    v3_add(p, p, cb);
    return *dist < margin;
}

/* btRigidBody::computeGyroscopicImpulseImplicit_Body (Bullet's default for rigid bodies,
 * BT_ENABLE_GYROSCOPIC_FORCE_IMPLICIT_BODY): one Newton step of the implicit Euler equation
 *     I (w' - w) + dt * w' x (I w') = 0
 * in body axes, f = dt * w x (I w), J = I + dt * ([w]x I - [I w]x), w' = w - J^-1 f (Cramer, as solve33).
 * The explicit form w += dt * I^-1 ((I w) x w) gains energy every step and runs away once |w| dt ~ 1 (a box
 * knocked into a spin by the gripper reached 1e6 rad/s within an episode). */
static void gyro_implicit(const real *Rb, const real *Ib, real dt, real *w /* world, in/out */) {
    real wl[3], iw[3], f[3], J[9], x[3], wn[3];
    m3_tvec(wl, Rb, w);
    v3_set(iw, Ib[0] * wl[0], Ib[1] * wl[1], Ib[2] * wl[2]);
    v3_cross(f, wl, iw);
    for (int k = 0; k < 3; k++) f[k] *= dt;
    const real s0[9] = {0, -wl[2], wl[1], wl[2], 0, -wl[0], -wl[1], wl[0], 0};
    const real s1[9] = {0, -iw[2], iw[1], iw[2], 0, -iw[0], -iw[1], iw[0], 0};
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) J[r * 3 + c] = (r == c ? Ib[r] : 0.0) + dt * (s0[r * 3 + c] * Ib[c] - s1[r * 3 + c]);
    real c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    real det = J[0] * c00 + J[1] * c01 + J[2] * c02, id = 1.0 / det;
    x[0] = (f[0] * c00 + f[1] * (J[2] * J[7] - J[1] * J[8]) + f[2] * (J[1] * J[5] - J[2] * J[4])) * id;
    x[1] = (f[0] * c01 + f[1] * (J[0] * J[8] - J[2] * J[6]) + f[2] * (J[2] * J[3] - J[0] * J[5])) * id;
    x[2] = (f[0] * c02 + f[1] * (J[1] * J[6] - J[0] * J[7]) + f[2] * (J[0] * J[4] - J[1] * J[3])) * id;
    for (int k = 0; k < 3; k++) wl[k] -= x[k];
    m3_vec(wn, Rb, wl);
    v3_copy(w, wn);
}

/* ------------------------------------------------------------------ one internal substep
 * (btMultiBodyDynamicsWorld::internalSingleStepSimulation, dt = timeStep / numSubSteps) */
static void substep(const xo_model *m, real *st, const real *q_target, real dt) {
    static const real finger_sign[2] = {1.0, -1.0};
    solver_t s;
    s.m = m;
    s.nrows = 0;
    tree_setup(m, st + S_Q, &s.t);
    int nd = s.t.nd;
    real *q = st + S_Q, *qd = st + S_QD, *bp = st + S_BP, *bq = st + S_BQ;
    real vb[6] = {st[S_BV], st[S_BV + 1], st[S_BV + 2], st[S_BW], st[S_BW + 1], st[S_BW + 2]};

    /* --- object frame and world inverse inertia (box: I = m/12 (b^2+c^2)) */
    quat_to_m3(s.Rb, bq);
    const real *h = m->obj_half;
    real Ib[3] = {m->obj_mass / 3.0 * (h[1] * h[1] + h[2] * h[2]), m->obj_mass / 3.0 * (h[0] * h[0] + h[2] * h[2]),
                  m->obj_mass / 3.0 * (h[0] * h[0] + h[1] * h[1])};
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            real v = 0;
            for (int k = 0; k < 3; k++) v += s.Rb[r * 3 + k] * s.Rb[c * 3 + k] / Ib[k];
            s.Iinv_w[r * 3 + c] = v;
        }

    /* --- unconstrained motion: arm (ABA with joint damping torque), then object */
    real tau[XO_MAXD], qdd[XO_MAXD];
    for (int i = 0; i < m->n_links; i++)
        if (s.t.dof[i] >= 0) tau[s.t.dof[i]] = -m->damping[i] * qd[s.t.dof[i]];
    aba_forward_dynamics(m, &s.t, qd, tau, m->gravity, qdd);
    for (int k = 0; k < nd; k++) qd[k] += dt * qdd[k];
    {
        /* gyroscopic term (implicit, body axes), gravity, then Bullet's (1-damping)^dt velocity damping */
        gyro_implicit(s.Rb, Ib, dt, vb + 3);
        vb[2] -= dt * m->gravity;
        real dl = pow(1.0 - m->lin_damping, dt), da = pow(1.0 - m->ang_damping, dt);
        for (int k = 0; k < 3; k++) { vb[k] *= dl; vb[k + 3] *= da; }
    }

    /* --- rows, in the order the solver sweeps them */
    real *lam_t = st + S_LT, *lam_p = st + S_LP;
    int row_t_n[8], row_p_n[8];
    /* (T) object corners against the table top; the manifold keeps at most 4 points
     * (btPersistentManifold), here the first four active corners in index order */
    int n_table = 0;
    for (int i = 0; i < 8; i++) {
        real rl[3] = {(i & 1) ? h[0] : -h[0], (i & 2) ? h[1] : -h[1], (i & 4) ? h[2] : -h[2]}, r[3], p[3];
        m3_vec(r, s.Rb, rl);
        v3_add(p, bp, r);
        real dist = p[2] - m->table_top_z;
        int active = dist < m->solver_margin && fabs(p[0]) <= m->table_half_x && fabs(p[1]) <= m->table_half_y &&
                     n_table < 4;
        row_t_n[i] = -1;
        if (!active) { lam_t[i] = 0; continue; }
        n_table++;
        real n[3] = {0, 0, 1};
        row_t_n[i] = add_contact(&s, -1, p, n, dist, dt, m->contact_erp, 0.0, m->mu_object * m->mu_table,
                                 m->warmstart * lam_t[i], bp);
    }
    /* (M) velocity-level PD motors, btMultiBodyJointMotor: one row per dof */
    for (int i = 0; i < m->n_links; i++) {
        if (s.t.dof[i] < 0) continue;
        int d = s.t.dof[i];
        row_t *r = row_new(&s);
        r->has_a = 1;
        r->Ja[d] = 1;
        r->vt = m->motor_kp * (q_target[d] - q[d]) / dt + (1.0 - m->motor_kd) * qd[d];
        real force = (m->jtype[i] == 2) ? m->finger_motor_force : m->arm_motor_force;
        r->hi = force * m->time_step;
        r->lo = -r->hi;
        row_finish(&s, r);
    }
    /* (L) joint limits, btMultiBodyJointLimitConstraint: lower then upper per dof */
    for (int i = 0; i < m->n_links; i++) {
        if (s.t.dof[i] < 0) continue;
        int d = s.t.dof[i];
        for (int side = 0; side < 2; side++) {
            real gap = side == 0 ? q[d] - m->lower[i] : m->upper[i] - q[d];
            if (gap >= m->limit_window) continue;
            row_t *r = row_new(&s);
            r->has_a = 1;
            r->Ja[d] = side == 0 ? 1.0 : -1.0;
            r->vt = gap < 0 ? -m->global_erp * gap / dt : -gap / dt;
            r->lo = 0; r->hi = 1e30;
            row_finish(&s, r);
        }
    }
    /* (G) gear between the two finger joints, ratio -1 (xarm_pick_and_place.py:78-79) */
    {
        int d1 = s.t.dof[m->finger_link[0]], d2 = s.t.dof[m->finger_link[1]];
        row_t *r = row_new(&s);
        r->has_a = 1;
        r->Ja[d1] = 1.0;
        r->Ja[d2] = -1.0;
        r->vt = -m->gear_erp * m->global_erp * (q[d1] - q[d2]) / dt;
        r->hi = m->gear_max_force * m->time_step;
        r->lo = -r->hi;
        row_finish(&s, r);
    }
    /* (F) finger pad spheres against the object */
    int touch[2] = {0, 0};
    {
        real denom = dt * m->finger_contact_stiffness + m->finger_contact_damping + m->object_contact_damping;
        real cfm = (1.0 / denom) / dt, erp = dt * m->finger_contact_stiffness / denom;
        real mu = m->mu_object * (st[S_MUG] > 0.5 ? m->mu_finger_grasp : m->mu_finger);
        for (int f = 0; f < 2; f++) {
            int l = m->finger_link[f];
            for (int j = 0; j < XO_NPAD; j++) {
                real cl[3] = {m->pad_center_left[j][0], finger_sign[f] * m->pad_center_left[j][1], m->pad_center_left[j][2]};
                real c[3], dist, n[3], p[3];
                m3_vec(c, s.t.R[l], cl);
                v3_add(c, c, s.t.o[l]);
                int idx = f * XO_NPAD + j;
                row_p_n[idx] = -1;
                /* getContactPoints() reports points inside the 0.02 breaking margin (touch flag); only
                 * points inside solver_margin can receive an impulse within one substep, so only those
                 * become solver rows (a row with dist/dt above any reachable approach speed is inert) */
                if (sphere_box(c, m->pad_radius, bp, s.Rb, h, m->contact_margin, &dist, n, p)) touch[f] = 1;
                if (!(dist < m->solver_margin)) { lam_p[idx] = 0; continue; }
                row_p_n[idx] = add_contact(&s, l, p, n, dist, dt, erp, cfm, mu, m->warmstart * lam_p[idx], bp);
            }
        }
    }
    st[S_TOUCH] = (touch[0] && touch[1]) ? 1.0 : 0.0;

    /* --- warm start, then projected Gauss-Seidel */
    for (int k = 0; k < s.nrows; k++)
        if (s.rows[k].lam != 0) apply_row_impulse(&s.rows[k], s.rows[k].lam, qd, vb, nd);
    for (int it = 0; it < m->num_iterations; it++) {
        for (int k = 0; k < s.nrows; k++) {
            row_t *r = &s.rows[k];
            if (r->normal_row >= 0) {
                real lim = r->mu * s.rows[r->normal_row].lam;
                r->lo = -lim; r->hi = lim;
            }
            real jv = 0;
            if (r->has_a) for (int c = 0; c < nd; c++) jv += r->Ja[c] * qd[c];
            if (r->has_b) for (int c = 0; c < 6; c++) jv += r->Jb[c] * vb[c];
            real dl = (r->vt - r->cfm * r->lam - jv) * r->inv_d;
            real nl = r->lam + dl;
            if (nl < r->lo) nl = r->lo;
            if (nl > r->hi) nl = r->hi;
            dl = nl - r->lam;
            r->lam = nl;
            apply_row_impulse(r, dl, qd, vb, nd);
        }
    }
    for (int i = 0; i < 8; i++)
        if (row_t_n[i] >= 0) lam_t[i] = s.rows[row_t_n[i]].lam;
    for (int i = 0; i < 2 * XO_NPAD; i++)
        if (row_p_n[i] >= 0) lam_p[i] = s.rows[row_p_n[i]].lam;

    /* --- integrate positions (semi-implicit Euler; quaternion by the exponential map of
     * btTransformUtil::integrateTransform) */
    for (int k = 0; k < nd; k++) q[k] += dt * qd[k];
    for (int k = 0; k < 3; k++) bp[k] += dt * vb[k];
    {
        real w[3] = {vb[3], vb[4], vb[5]}, ang = v3_norm(w), ax[3];
        if (ang * dt > 0.7853981633974483) ang = 0.7853981633974483 / dt;
        real k = ang < 0.001 ? 0.5 * dt - dt * dt * dt * 0.020833333333 * ang * ang : sin(0.5 * ang * dt) / ang;
        v3_set(ax, w[0] * k, w[1] * k, w[2] * k);
        real cw = cos(ang * dt * 0.5);
        real x = bq[0], y = bq[1], z = bq[2], w0 = bq[3];
        /* dorn * q */
        real nx = cw * x + ax[0] * w0 + ax[1] * z - ax[2] * y;
        real ny = cw * y + ax[1] * w0 + ax[2] * x - ax[0] * z;
        real nz = cw * z + ax[2] * w0 + ax[0] * y - ax[1] * x;
        real nw = cw * w0 - ax[0] * x - ax[1] * y - ax[2] * z;
        real inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        bq[0] = nx * inv; bq[1] = ny * inv; bq[2] = nz * inv; bq[3] = nw * inv;
    }
    for (int k = 0; k < 3; k++) { st[S_BV + k] = vb[k]; st[S_BW + k] = vb[k + 3]; }
}

/* p.stepSimulation() with numSubSteps = n_substeps (xarm_pick_and_place.py:64,111) */
static void sim_tick(const xo_model *m, real *st, const real *q_target) {
    real dt = m->time_step / m->n_substeps;
    for (int k = 0; k < m->n_substeps; k++) substep(m, st, q_target, dt);
}

/* ------------------------------------------------------------------ observation (_get_obs :220-248) */
static void hand_com_state(const xo_model *m, const real *st, real *pos, real *vel) {
    tree_t t;
    tree_setup(m, st + S_Q, &t);
    int l = m->hand_link;
    real c[3];
    m3_vec(c, t.R[l], m->com[l]);
    v3_add(pos, t.o[l], c);
    for (int k = 0; k < 3; k++) {
        real d[3] = {0, 0, 0}, J[XO_MAXD];
        d[k] = 1;
        point_jacobian_row(m, &t, l, pos, d, J);
        real s = 0;
        for (int j = 0; j < t.nd; j++) s += J[j] * st[S_QD + j];
        vel[k] = s;
    }
}
static void get_obs(const xo_model *m, const real *st, real *obs, real *ag, real *dg) {
    real hp[3], hv[3];
    tree_t t;
    tree_setup(m, st + S_Q, &t);
    hand_com_state(m, st, hp, hv);
    int d1 = t.dof[m->finger_link[0]];
    for (int k = 0; k < 3; k++) { obs[k] = hp[k]; obs[3 + k] = hv[k]; }
    obs[6] = st[S_Q + d1];
    obs[7] = st[S_QD + d1];
    for (int k = 0; k < 3; k++) obs[8 + k] = st[S_BP + k];
    for (int k = 0; k < 4; k++) obs[11 + k] = st[S_BQ + k];
    for (int k = 0; k < 3; k++) obs[15 + k] = st[S_BV + k] - hv[k];
    for (int k = 0; k < 3; k++) obs[18 + k] = st[S_BW + k];
    for (int k = 0; k < 3; k++) obs[21 + k] = st[S_BP + k] - hp[k];
    for (int k = 0; k < 3; k++) { ag[k] = st[S_BP + k]; dg[k] = st[S_GOAL + k]; }
}

/* ------------------------------------------------------------------ sampling (counter RNG) */
static void sample_draws(const xo_pnp_cfg *cfg, int64_t env, int64_t episode, real *u /*8*/) {
    uint32_t o[4];
    uint64_t gid = (uint64_t)(cfg->env_id_offset + env);
    for (int b = 0; b < 2; b++) {
        xo_philox(cfg->seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, (uint32_t)b, o);
        for (int k = 0; k < 4; k++) u[b * 4 + k] = u01(o[k]);
    }
}
/* draws: 0 init-grasp coin, 1-2 object xy, 3-5 goal xyz, 6 goal-on-ground coin */
static void sample_object(const xo_model *m, const xo_pnp_cfg *cfg, const real *u, real *st) {
    if (u[0] < cfg->init_grasp_rate) {
        st[S_BP] = m->start_gripper_pos[0];
        st[S_BP + 1] = m->start_gripper_pos[1];
    } else {
        st[S_BP] = m->obj_low[0] + u[1] * (m->obj_high[0] - m->obj_low[0]);
        st[S_BP + 1] = m->obj_low[1] + u[2] * (m->obj_high[1] - m->obj_low[1]);
    }
    st[S_BP + 2] = m->height_offset;
    st[S_BQ] = st[S_BQ + 1] = st[S_BQ + 2] = 0; st[S_BQ + 3] = 1;
    for (int k = 0; k < 6; k++) st[S_BV + k] = 0;
    for (int k = 0; k < 16; k++) st[S_LT + k] = 0;
}
static void sample_goal(const xo_model *m, const xo_pnp_cfg *cfg, const real *u, real *st) {
    for (int k = 0; k < 3; k++) st[S_GOAL + k] = m->goal_low[k] + u[3 + k] * (m->goal_high[k] - m->goal_low[k]);
    if (cfg->goal_shape == 0) {
        if (u[6] < cfg->goal_ground_rate) st[S_GOAL + 2] = m->goal_low[2];
    } else
        st[S_GOAL + 2] = m->height_offset;
}

/* ------------------------------------------------------------------ public API */
int xo_state_dim(void) { return XO_STATE_DIM; }

int xo_pnp_init(const xo_model *m, const xo_pnp_cfg *cfg, int64_t E, double *state) {
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_STATE_DIM, u[8];
        memset(st, 0, XO_STATE_DIM * sizeof(real));
        sample_draws(cfg, e, 0, u);
        sample_object(m, cfg, u, st);
        sample_goal(m, cfg, u, st);
    }
    return 0;
}

static void reset_one(const xo_model *m, const xo_pnp_cfg *cfg, int64_t e, real *st) {
    real tgt[XO_MAXD], u[8];
    int64_t episode = (int64_t)st[S_EPISODE] + 1;
    for (int k = 0; k < m->reset_ticks; k++) {
        ik_solve(m, st + S_Q, m->start_gripper_pos, m->n_substeps, tgt);
        tgt[dof_of_link(m, m->finger_link[0])] = tgt[dof_of_link(m, m->finger_link[1])] = m->reset_finger_target;
        sim_tick(m, st, tgt);
    }
    sample_draws(cfg, e, episode, u);
    sample_object(m, cfg, u, st);
    sim_tick(m, st, tgt);
    sample_goal(m, cfg, u, st);
    st[S_STEPS] = 0;
    st[S_EPISODE] = (real)episode;
}

int xo_pnp_reset(const xo_model *m, const xo_pnp_cfg *cfg, int64_t E, double *state, const uint8_t *mask,
                 double *obs, double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        real *st = state + e * XO_STATE_DIM;
        reset_one(m, cfg, e, st);
        if (obs) get_obs(m, st, obs + e * XO_OBS_DIM, ag + e * XO_GOAL_DIM, dg + e * XO_GOAL_DIM);
    }
    return 0;
}

int xo_pnp_compute_reward(const xo_model *m, int reward_type, int64_t n, const double *ag, const double *g,
                          double *out) {
    for (int64_t i = 0; i < n; i++) {
        real d[3];
        v3_sub(d, ag + i * 3, g + i * 3);
        real dist = v3_norm(d);
        if (reward_type == 0) out[i] = dist < m->distance_threshold ? 1.0 : 0.0;
        else if (reward_type == 1) out[i] = -dist;
        else return -1;
    }
    return 0;
}

/* staged dense reward, xarm_pick_and_place.py:166-175; if_grasp = both fingers have contact points
 * with object 0 after the step, grip_pos = hand COM (getLinkState(9)[0]) - eef2grip_offset */
double xo_pnp_dense_reward(const xo_model *m, int if_grasp, const double *hand_com, const double *ag, const double *g) {
    real gp[3] = {hand_com[0] - m->eef2grip[0] - ag[0] + 0.06, hand_com[1] - m->eef2grip[1] - ag[1],
                  hand_com[2] - m->eef2grip[2] - ag[2]};
    real d[3];
    v3_sub(d, ag, g);
    if (!if_grasp) return 0.25 * (1 - tanh(1.0 * v3_norm(gp)));
    if (ag[2] > 0.05) return 1.0 + 0.25 * (1 - tanh(1.0 * v3_norm(d)));
    return 0.5;
}

int xo_pnp_step(const xo_model *m, const xo_pnp_cfg *cfg, int64_t E, double *state, const double *actions,
                double *obs, double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success) {
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_STATE_DIM;
        const real *act = actions + e * XO_ACT_DIM;
        real a[4], tgt[XO_MAXD], new_pos[3];
        tree_t t;
        st[S_STEPS] += 1;
        /* _set_action :199-218 */
        for (int k = 0; k < 4; k++) a[k] = act[k] < -1 ? -1 : (act[k] > 1 ? 1 : act[k]);
        tree_setup(m, st + S_Q, &t);
        for (int k = 0; k < 3; k++) {
            real v = t.o[m->eef_link][k] + a[k] * m->max_vel * m->action_dt;
            new_pos[k] = v < m->pos_low[k] ? m->pos_low[k] : (v > m->pos_high[k] ? m->pos_high[k] : v);
        }
        int d1 = t.dof[m->finger_link[0]], d2 = t.dof[m->finger_link[1]];
        real g = st[S_Q + d1] + a[3] * m->action_dt * m->max_gripper_vel;
        g = g < m->gripper_low ? m->gripper_low : (g > m->gripper_high ? m->gripper_high : g);
        ik_solve(m, st + S_Q, new_pos, m->n_substeps, tgt);
        tgt[d1] = tgt[d2] = g;
        st[S_MUG] = st[S_TOUCH]; /* friction toggle from LAST step's contacts :212-218 */
        sim_tick(m, st, tgt);
        /* obs / info / reward / done :112-119 */
        get_obs(m, st, obs + e * XO_OBS_DIM, ag + e * XO_GOAL_DIM, dg + e * XO_GOAL_DIM);
        real d[3];
        v3_sub(d, ag + e * 3, dg + e * 3);
        real dist = v3_norm(d);
        int succ = dist < m->distance_threshold;
        success[e] = (uint8_t)succ;
        if (cfg->reward_type == 2)
            reward[e] = xo_pnp_dense_reward(m, st[S_TOUCH] > 0.5, obs + e * XO_OBS_DIM, ag + e * 3, dg + e * 3);
        else
            xo_pnp_compute_reward(m, cfg->reward_type, 1, ag + e * 3, dg + e * 3, reward + e);
        done[e] = (uint8_t)(succ || ((int)st[S_STEPS] == m->max_episode_steps));
    }
    return 0;
}

/* test hook, counterpart of xarm_debug_substeps: n internal substeps (dt = time_step / n_substeps) toward the joint
 * targets q_target [E, 9]; no action / IK / observation logic.  Used for the literal replay of
 * XarmPickAndPlace._run_demo (xarm_pick_and_place.py:310-349), which drives the motors below env.step. */
int xo_pnp_substeps(const xo_model *m, int64_t E, double *state, const double *q_target, int32_t n) {
    real dt = m->time_step / m->n_substeps;
    for (int64_t e = 0; e < E; e++)
        for (int k = 0; k < n; k++) substep(m, state + e * XO_STATE_DIM, q_target + e * 9, dt);
    return 0;
}
int xo_fk(const xo_model *m, const double *q, double *link_pos, double *link_rot) {
    tree_t t;
    tree_setup(m, q, &t);
    for (int i = 0; i < m->n_links; i++) {
        memcpy(link_pos + i * 3, t.o[i], 3 * sizeof(real));
        memcpy(link_rot + i * 9, t.R[i], 9 * sizeof(real));
    }
    return 0;
}
int xo_ik(const xo_model *m, const double *q, const double *target, int max_iter, double *q_out) {
    ik_solve(m, q, target, max_iter, q_out);
    return 0;
}
int xo_forward_dynamics(const xo_model *m, const double *q, const double *qd, const double *tau, double *qdd) {
    tree_t t;
    tree_setup(m, q, &t);
    aba_forward_dynamics(m, &t, qd, tau, m->gravity, qdd);
    return 0;
}
int xo_mass_matrix_inv(const xo_model *m, const double *q, double *minv) {
    tree_t t;
    real qd[XO_MAXD] = {0}, tau[XO_MAXD] = {0}, qdd[XO_MAXD];
    tree_setup(m, q, &t);
    aba_forward_dynamics(m, &t, qd, tau, 0.0, qdd);
    for (int k = 0; k < t.nd; k++) {
        real imp[XO_MAXD] = {0}, col[XO_MAXD];
        imp[k] = 1;
        aba_impulse_response(m, &t, imp, col);
        for (int r = 0; r < t.nd; r++) minv[r * t.nd + k] = col[r];
    }
    return 0;
}

/* ==================================================================================== XarmReach-v0
 * /root/reference/gym_xarm/envs/xarm_reach.py: step :81-94, _set_action :131-142, _get_obs :144-161,
 * reset/_reset_sim/_sample_goal :96-102,163-173, compute_reward :107-116, _is_success :175-177.
 * Contact-free: the solver rows are the 13 POSITION_CONTROL motors (force 5*240, :140-142) and the
 * joint limits; 20 substeps of 1/4800 s per env step (:16-17,51). */
enum { R_Q = 0, R_QD = 13, R_QT = 26, R_GOAL = 39, R_DOLD = 42, R_STEPS = 43, R_EPISODE = 44 };

static void reach_substep(const xo_model *m, const xo_reach_cfg *cfg, real *st, real dt) {
    solver_t s;
    s.m = m;
    s.nrows = 0;
    tree_setup(m, st + R_Q, &s.t);
    int nd = s.t.nd;
    real *q = st + R_Q, *qd = st + R_QD, *qt = st + R_QT;
    real tau[XO_MAXD] = {0}, qdd[XO_MAXD], vb[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < m->n_links; i++)
        if (s.t.dof[i] >= 0) tau[s.t.dof[i]] = -m->damping[i] * qd[s.t.dof[i]];
    aba_forward_dynamics(m, &s.t, qd, tau, m->gravity, qdd);
    for (int k = 0; k < nd; k++) qd[k] += dt * qdd[k];
    for (int i = 0; i < m->n_links; i++) {
        if (s.t.dof[i] < 0) continue;
        int d = s.t.dof[i];
        row_t *r = row_new(&s);
        r->has_a = 1;
        r->Ja[d] = 1;
        r->vt = m->motor_kp * (qt[d] - q[d]) / dt + (1.0 - m->motor_kd) * qd[d];
        r->hi = cfg->motor_force * cfg->time_step;
        r->lo = -r->hi;
        row_finish(&s, r);
    }
    for (int i = 0; i < m->n_links; i++) {
        if (s.t.dof[i] < 0) continue;
        int d = s.t.dof[i];
        for (int side = 0; side < 2; side++) {
            real gap = side == 0 ? q[d] - m->lower[i] : m->upper[i] - q[d];
            if (gap >= m->limit_window) continue;
            row_t *r = row_new(&s);
            r->has_a = 1;
            r->Ja[d] = side == 0 ? 1.0 : -1.0;
            r->vt = gap < 0 ? -m->global_erp * gap / dt : -gap / dt;
            r->lo = 0; r->hi = 1e30;
            row_finish(&s, r);
        }
    }
    for (int it = 0; it < m->num_iterations; it++)
        for (int k = 0; k < s.nrows; k++) {
            row_t *r = &s.rows[k];
            real jv = 0;
            for (int c = 0; c < nd; c++) jv += r->Ja[c] * qd[c];
            real dl = (r->vt - jv) * r->inv_d, nl = r->lam + dl;
            if (nl < r->lo) nl = r->lo;
            if (nl > r->hi) nl = r->hi;
            dl = nl - r->lam;
            r->lam = nl;
            apply_row_impulse(r, dl, qd, vb, nd);
        }
    for (int k = 0; k < nd; k++) q[k] += dt * qd[k];
}
static void reach_tick(const xo_model *m, const xo_reach_cfg *cfg, real *st) {
    real dt = cfg->time_step / cfg->n_substeps;
    for (int k = 0; k < cfg->n_substeps; k++) reach_substep(m, cfg, st, dt);
}
static void reach_obs(const xo_model *m, const xo_reach_cfg *cfg, const real *st, real *obs, real *ag, real *dg) {
    tree_t t;
    tree_setup(m, st + R_Q, &t);
    int l = m->hand_link, dd = t.dof[cfg->driver_link];
    real c[3], hp[3];
    m3_vec(c, t.R[l], m->com[l]);
    v3_add(hp, t.o[l], c);
    for (int k = 0; k < 3; k++) {
        real d[3] = {0, 0, 0}, J[XO_MAXD], s = 0;
        d[k] = 1;
        point_jacobian_row(m, &t, l, hp, d, J);
        for (int j = 0; j < t.nd; j++) s += J[j] * st[R_QD + j];
        obs[k] = hp[k];
        obs[3 + k] = s;
        ag[k] = hp[k];
        dg[k] = st[R_GOAL + k];
    }
    obs[6] = st[R_Q + dd];
    obs[7] = st[R_QD + dd];
}
static void reach_sample_goal(const xo_reach_cfg *cfg, int64_t env, int64_t episode, real *st) {
    uint32_t o[4];
    uint64_t gid = (uint64_t)(cfg->env_id_offset + env);
    xo_philox(cfg->seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, 0u, o);
    for (int k = 0; k < 3; k++) st[R_GOAL + k] = cfg->goal_low[k] + u01(o[k]) * (cfg->goal_high[k] - cfg->goal_low[k]);
}
int xo_reach_init(const xo_model *m, const xo_reach_cfg *cfg, int64_t E, double *state) {
    (void)m;
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_REACH_STATE_DIM;
        memset(st, 0, XO_REACH_STATE_DIM * sizeof(real));
        for (int k = 0; k < XO_MAXD; k++) st[R_Q + k] = st[R_QT + k] = cfg->joint_init_pos[k];
        reach_sample_goal(cfg, e, 0, st);
    }
    return 0;
}
int xo_reach_reset(const xo_model *m, const xo_reach_cfg *cfg, int64_t E, double *state, const uint8_t *mask,
                   double *obs, double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        real *st = state + e * XO_REACH_STATE_DIM;
        int64_t episode = (int64_t)st[R_EPISODE] + 1;
        /* _reset_sim :163-168: resetJointState(all joints) then one stepSimulation; the motors keep the
         * targets of the last _set_action */
        for (int k = 0; k < XO_MAXD; k++) { st[R_Q + k] = cfg->joint_init_pos[k]; st[R_QD + k] = 0; }
        reach_tick(m, cfg, st);
        reach_sample_goal(cfg, e, episode, st);
        real o8[8], a3[3], g3[3], d[3];
        reach_obs(m, cfg, st, o8, a3, g3);
        v3_sub(d, a3, g3);
        st[R_DOLD] = v3_norm(d); /* :100 */
        st[R_STEPS] = 0;
        st[R_EPISODE] = (real)episode;
        if (obs) {
            memcpy(obs + e * 8, o8, sizeof o8);
            memcpy(ag + e * 3, a3, sizeof a3);
            memcpy(dg + e * 3, g3, sizeof g3);
        }
    }
    return 0;
}
int xo_reach_compute_reward(const xo_reach_cfg *cfg, int reward_type, int64_t n, const double *ag, const double *g,
                            double *out) {
    for (int64_t i = 0; i < n; i++) {
        real d[3];
        v3_sub(d, ag + i * 3, g + i * 3);
        real dist = v3_norm(d);
        if (reward_type == 0) out[i] = dist < cfg->distance_threshold ? 1.0 : 0.0;
        else if (reward_type == 1) out[i] = -dist;
        else return -1; /* dense_diff is stateful (self.d_old) */
    }
    return 0;
}
int xo_reach_step(const xo_model *m, const xo_reach_cfg *cfg, int64_t E, double *state, const double *actions,
                  double *obs, double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success,
                  int32_t *future_length) {
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_REACH_STATE_DIM;
        const real *act = actions + e * 4;
        real a[4], tgt[XO_MAXD], new_pos[3];
        tree_t t;
        st[R_STEPS] += 1;
        for (int k = 0; k < 4; k++) a[k] = act[k] < -1 ? -1 : (act[k] > 1 ? 1 : act[k]); /* :83 */
        tree_setup(m, st + R_Q, &t);
        for (int k = 0; k < 3; k++) {
            real v = t.o[m->eef_link][k] + a[k] * cfg->max_vel * cfg->action_dt;
            new_pos[k] = v < cfg->pos_low[k] ? cfg->pos_low[k] : (v > cfg->pos_high[k] ? cfg->pos_high[k] : v);
        }
        int dd = t.dof[cfg->driver_link];
        real g = st[R_Q + dd] + a[3] * cfg->action_dt * cfg->max_gripper_vel; /* no clip, :137 */
        ik_solve(m, st + R_Q, new_pos, cfg->n_substeps, tgt);
        for (int k = 0; k < t.nd; k++) st[R_QT + k] = k < 7 ? tgt[k] : g; /* joints 10..16 all get new_gripper_pos :141-142 */
        reach_tick(m, cfg, st);
        reach_obs(m, cfg, st, obs + e * 8, ag + e * 3, dg + e * 3);
        real d[3];
        v3_sub(d, ag + e * 3, dg + e * 3);
        real dist = v3_norm(d);
        success[e] = (uint8_t)(dist < cfg->distance_threshold);
        if (cfg->reward_type == 0) reward[e] = success[e] ? 1.0 : 0.0;
        else if (cfg->reward_type == 1) reward[e] = -dist;
        else { reward[e] = st[R_DOLD] - dist; st[R_DOLD] = dist; } /* :113-116 */
        done[e] = (uint8_t)((int)st[R_STEPS] == cfg->max_episode_steps); /* :93 */
        if (future_length) future_length[e] = cfg->max_episode_steps - (int)st[R_STEPS]; /* :90 */
    }
    return 0;
}

/* ==================================================================================== XarmHandover-v0
 * /root/reference/gym_xarm/envs/xarm_handover.py: step :128-139, _set_action :244-297, _get_obs :299-336,
 * reset/_reset_sim/_sample_goal :141-145,338-393, compute_reward (sparse) :164-183, _is_success :395-402.
 * Two xarm7_pd arms (bases at x = -+0.6, the second yawed by pi, :50-53,95-96), one 0.15 x 0.05 x 0.05 stick
 * (:89-92), two tables with a 0.2 m gap over a ground plane at z = -0.625 (:79-83); timeStep 1/240 with the
 * default single substep, 15 stepSimulation calls per env step (:28-29,131-132).  num_obj = 1, use_stand False. */
enum { H_Q = 0, H_QD = 18, H_FT = 36, H_BP = 38, H_BQ = 41, H_BV = 45, H_BW = 48, H_GOAL = 51, H_LT = 54, H_LP = 62,
       H_TOUCH = 70, H_MUG = 72, H_STEPS = 74, H_EPISODE = 75 };

static void ho_base(const xo_ho_cfg *c, int arm, real *R, real *p) {
    real rpy[3] = {0, 0, c->base_yaw[arm]};
    m3_from_rpy(R, rpy);
    v3_copy(p, c->base_pos[arm]);
}

/* the static stand under goal g (config['use_stand'], xarm_handover.py:391-392): the candidates of a stick's support
 * manifold against the stand's top rectangle - the overlap of the footprint of the stick's most downward face with the
 * rectangle (both axis-aligned), on the face's plane.  Returns 0 (no overlap) or 4; dist[v] = gap to the stand's top. */
static int ho_stand_points(const xo_ho_cfg *c, const real *Rb, const real *bp, const real *h, const real *g, real pts[4][3], real dist[4]) {
    real top = g[2] - c->stand_below_goal + c->stand_half[2];
    int kf = 0;
    real sgn = 1, bestz = 1e30;
    for (int k = 0; k < 3; k++)
        for (int sg = -1; sg <= 1; sg += 2) {
            real nz = sg * Rb[2 * 3 + k];            /* z component of the face normal sg * axis k */
            if (nz < bestz) { bestz = nz; kf = k; sgn = sg; }
        }
    int k1 = (kf + 1) % 3, k2 = (kf + 2) % 3;
    real nf[3] = {sgn * Rb[0 * 3 + kf], sgn * Rb[1 * 3 + kf], sgn * Rb[2 * 3 + kf]}, fc[3];
    for (int r = 0; r < 3; r++) fc[r] = bp[r] + nf[r] * h[kf];
    real xlo = 1e30, xhi = -1e30, ylo = 1e30, yhi = -1e30;
    for (int v = 0; v < 4; v++) {
        real s1 = (v & 1) ? h[k1] : -h[k1], s2 = (v & 2) ? h[k2] : -h[k2];
        real x = fc[0] + s1 * Rb[0 * 3 + k1] + s2 * Rb[0 * 3 + k2], y = fc[1] + s1 * Rb[1 * 3 + k1] + s2 * Rb[1 * 3 + k2];
        xlo = x < xlo ? x : xlo; xhi = x > xhi ? x : xhi; ylo = y < ylo ? y : ylo; yhi = y > yhi ? y : yhi;
    }
    real ox0 = xlo > g[0] - c->stand_half[0] ? xlo : g[0] - c->stand_half[0], ox1 = xhi < g[0] + c->stand_half[0] ? xhi : g[0] + c->stand_half[0];
    real oy0 = ylo > g[1] - c->stand_half[1] ? ylo : g[1] - c->stand_half[1], oy1 = yhi < g[1] + c->stand_half[1] ? yhi : g[1] + c->stand_half[1];
    if (!(ox0 < ox1 && oy0 < oy1)) return 0;
    for (int v = 0; v < 4; v++) {
        real x = (v & 1) ? ox1 : ox0, y = (v & 2) ? oy1 : oy0;
        real z = fc[2] - (nf[0] * (x - fc[0]) + nf[1] * (y - fc[1])) / nf[2];      /* on the face's plane; nf[2] <= -1/sqrt(3) */
        pts[v][0] = x; pts[v][1] = y; pts[v][2] = z;
        dist[v] = z - top;
    }
    return 4;
}

static void ho_substep(const xo_model *m, const xo_ho_cfg *c, real *st, const real qt[2][XO_MAXD], real dt) {
    static const real finger_sign[2] = {1.0, -1.0};
    solver_t s;
    s.m = m;
    s.nrows = 0;
    tree_t *tr[2] = {&s.t, &s.t2};
    for (int a = 0; a < 2; a++) {
        real Rb[9], pb[3];
        ho_base(c, a, Rb, pb);
        tree_setup_base(m, st + H_Q + 9 * a, tr[a], Rb, pb);
    }
    real *bp = st + H_BP, *bq = st + H_BQ;
    real vb[6] = {st[H_BV], st[H_BV + 1], st[H_BV + 2], st[H_BW], st[H_BW + 1], st[H_BW + 2]};
    quat_to_m3(s.Rb, bq);
    const real *h = c->obj_half;
    real Ib[3] = {m->obj_mass / 3.0 * (h[1] * h[1] + h[2] * h[2]), m->obj_mass / 3.0 * (h[0] * h[0] + h[2] * h[2]),
                  m->obj_mass / 3.0 * (h[0] * h[0] + h[1] * h[1])};
    for (int r = 0; r < 3; r++)
        for (int cc = 0; cc < 3; cc++) {
            real v = 0;
            for (int k = 0; k < 3; k++) v += s.Rb[r * 3 + k] * s.Rb[cc * 3 + k] / Ib[k];
            s.Iinv_w[r * 3 + cc] = v;
        }
    /* unconstrained motion */
    for (int a = 0; a < 2; a++) {
        real tau[XO_MAXD] = {0}, qdd[XO_MAXD], *qd = st + H_QD + 9 * a;
        for (int i = 0; i < m->n_links; i++)
            if (tr[a]->dof[i] >= 0) tau[tr[a]->dof[i]] = -m->damping[i] * qd[tr[a]->dof[i]];
        aba_forward_dynamics(m, tr[a], qd, tau, m->gravity, qdd);
        for (int k = 0; k < 9; k++) qd[k] += dt * qdd[k];
    }
    {
        gyro_implicit(s.Rb, Ib, dt, vb + 3);
        vb[2] -= dt * m->gravity;
        real dl = pow(1.0 - m->lin_damping, dt), da = pow(1.0 - m->ang_damping, dt);
        for (int k = 0; k < 3; k++) { vb[k] *= dl; vb[k + 3] *= da; }
    }
    real *lam_t = st + H_LT, *lam_p = st + H_LP;
    int row_t_n[8], row_p_n[8], n_table = 0;
    /* (T) stick corners against the table tops (z = 0) or, over the gap / beside the tables, the ground plane */
    for (int i = 0; i < 8; i++) {
        real rl[3] = {(i & 1) ? h[0] : -h[0], (i & 2) ? h[1] : -h[1], (i & 4) ? h[2] : -h[2]}, r[3], p[3];
        m3_vec(r, s.Rb, rl);
        v3_add(p, bp, r);
        int on_table = fabs(p[0]) >= c->table_x_min && fabs(p[0]) <= c->table_x_max && fabs(p[1]) <= c->table_half_y;
        real dist = p[2] - (on_table ? m->table_top_z : c->ground_z);
        int active = dist < m->solver_margin && n_table < 4;
        row_t_n[i] = -1;
        if (!active) { lam_t[i] = 0; continue; }
        n_table++;
        real n[3] = {0, 0, 1};
        row_t_n[i] = add_contact_arm(&s, 0, -1, p, n, dist, dt, m->contact_erp, 0.0, m->mu_object * m->mu_table,
                                     m->warmstart * lam_t[i], bp);
    }
    /* (S) the static stand under the goal (config['use_stand'], :391-392): the stick's most downward face against the
     * stand's top rectangle.  The overlap of the face's footprint with the rectangle (both taken axis-aligned: the env
     * zeroes the stick's roll and yaw every step, :282-297) gives up to four support points on the face's plane with
     * the stand's normal +z; they join the <= 4 point object/support manifold after the corners, no warm start. */
    if (c->use_stand) {
        real pts[4][3], dd[4];
        int np = ho_stand_points(c, s.Rb, bp, h, st + H_GOAL, pts, dd);
        for (int v = 0; v < np; v++) {
            real n[3] = {0, 0, 1};
            int active = dd[v] < m->solver_margin && dd[v] > -(2 * c->stand_half[2] + 0.01) && n_table < 4;
            if (!active) continue;
            n_table++;
            add_contact_arm(&s, 0, -1, pts[v], n, dd[v], dt, m->contact_erp, 0.0, m->mu_object * m->mu_table, 0.0, bp);
        }
    }
    /* (M)(L)(G) per arm */
    for (int a = 0; a < 2; a++) {
        real *q = st + H_Q + 9 * a, *qd = st + H_QD + 9 * a;
        for (int i = 0; i < m->n_links; i++) {
            if (tr[a]->dof[i] < 0) continue;
            int d = tr[a]->dof[i];
            row_t *r = row_new(&s);
            r->has_a = 1; r->arm = a;
            r->Ja[d] = 1;
            r->vt = m->motor_kp * (qt[a][d] - q[d]) / dt + (1.0 - m->motor_kd) * qd[d];
            real force = (m->jtype[i] == 2) ? c->finger_motor_force : m->arm_motor_force;
            r->hi = force * c->time_step;
            r->lo = -r->hi;
            row_finish(&s, r);
        }
        for (int i = 0; i < m->n_links; i++) {
            if (tr[a]->dof[i] < 0) continue;
            int d = tr[a]->dof[i];
            for (int side = 0; side < 2; side++) {
                real gap = side == 0 ? q[d] - m->lower[i] : m->upper[i] - q[d];
                if (gap >= m->limit_window) continue;
                row_t *r = row_new(&s);
                r->has_a = 1; r->arm = a;
                r->Ja[d] = side == 0 ? 1.0 : -1.0;
                r->vt = gap < 0 ? -m->global_erp * gap / dt : -gap / dt;
                r->lo = 0; r->hi = 1e30;
                row_finish(&s, r);
            }
        }
        {
            int d1 = tr[a]->dof[m->finger_link[0]], d2 = tr[a]->dof[m->finger_link[1]];
            row_t *r = row_new(&s);
            r->has_a = 1; r->arm = a;
            r->Ja[d1] = 1.0;
            r->Ja[d2] = -1.0;
            r->vt = -m->gear_erp * m->global_erp * (q[d1] - q[d2]) / dt;
            r->hi = m->gear_max_force * c->time_step;
            r->lo = -r->hi;
            row_finish(&s, r);
        }
    }
    /* (F) pads of arm 0, then of arm 1, against the stick */
    {
        real denom = dt * m->finger_contact_stiffness + m->finger_contact_damping + m->object_contact_damping;
        real cfm = (1.0 / denom) / dt, erp = dt * m->finger_contact_stiffness / denom;
        for (int a = 0; a < 2; a++) {
            int touch[2] = {0, 0};
            real mu = m->mu_object * (st[H_MUG + a] > 0.5 ? m->mu_finger_grasp : m->mu_finger);
            for (int f = 0; f < 2; f++) {
                int l = m->finger_link[f];
                for (int j = 0; j < XO_NPAD; j++) {
                    real cl[3] = {m->pad_center_left[j][0], finger_sign[f] * m->pad_center_left[j][1], m->pad_center_left[j][2]};
                    real cw[3], dist, n[3], p[3];
                    m3_vec(cw, tr[a]->R[l], cl);
                    v3_add(cw, cw, tr[a]->o[l]);
                    int idx = a * 4 + f * XO_NPAD + j;
                    row_p_n[idx] = -1;
                    if (sphere_box(cw, m->pad_radius, bp, s.Rb, h, m->contact_margin, &dist, n, p)) touch[f] = 1;
                    if (!(dist < m->solver_margin)) { lam_p[idx] = 0; continue; }
                    row_p_n[idx] = add_contact_arm(&s, a, l, p, n, dist, dt, erp, cfm, mu, m->warmstart * lam_p[idx], bp);
                }
            }
            st[H_TOUCH + a] = (touch[0] && touch[1]) ? 1.0 : 0.0;
        }
    }
    /* warm start + PGS */
    for (int k = 0; k < s.nrows; k++)
        if (s.rows[k].lam != 0) apply_row_impulse(&s.rows[k], s.rows[k].lam, st + H_QD + 9 * s.rows[k].arm, vb, 9);
    for (int it = 0; it < m->num_iterations; it++)
        for (int k = 0; k < s.nrows; k++) {
            row_t *r = &s.rows[k];
            real *qd = st + H_QD + 9 * r->arm;
            if (r->normal_row >= 0) {
                real lim = r->mu * s.rows[r->normal_row].lam;
                r->lo = -lim; r->hi = lim;
            }
            real jv = 0;
            if (r->has_a) for (int cc = 0; cc < 9; cc++) jv += r->Ja[cc] * qd[cc];
            if (r->has_b) for (int cc = 0; cc < 6; cc++) jv += r->Jb[cc] * vb[cc];
            real dl = (r->vt - r->cfm * r->lam - jv) * r->inv_d, nl = r->lam + dl;
            if (nl < r->lo) nl = r->lo;
            if (nl > r->hi) nl = r->hi;
            dl = nl - r->lam;
            r->lam = nl;
            apply_row_impulse(r, dl, qd, vb, 9);
        }
    for (int i = 0; i < 8; i++) {
        if (row_t_n[i] >= 0) lam_t[i] = s.rows[row_t_n[i]].lam;
        if (row_p_n[i] >= 0) lam_p[i] = s.rows[row_p_n[i]].lam;
    }
    for (int k = 0; k < 18; k++) st[H_Q + k] += dt * st[H_QD + k];
    for (int k = 0; k < 3; k++) bp[k] += dt * vb[k];
    {
        real w[3] = {vb[3], vb[4], vb[5]}, ang = v3_norm(w), ax[3];
        if (ang * dt > 0.7853981633974483) ang = 0.7853981633974483 / dt;
        real k = ang < 0.001 ? 0.5 * dt - dt * dt * dt * 0.020833333333 * ang * ang : sin(0.5 * ang * dt) / ang;
        v3_set(ax, w[0] * k, w[1] * k, w[2] * k);
        real cw = cos(ang * dt * 0.5), x = bq[0], y = bq[1], z = bq[2], w0 = bq[3];
        real nx = cw * x + ax[0] * w0 + ax[1] * z - ax[2] * y, ny = cw * y + ax[1] * w0 + ax[2] * x - ax[0] * z;
        real nz = cw * z + ax[2] * w0 + ax[0] * y - ax[1] * x, nw = cw * w0 - ax[0] * x - ax[1] * y - ax[2] * z;
        real inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        bq[0] = nx * inv; bq[1] = ny * inv; bq[2] = nz * inv; bq[3] = nw * inv;
    }
    for (int k = 0; k < 3; k++) { st[H_BV + k] = vb[k]; st[H_BW + k] = vb[k + 3]; }
}

/* IK of one arm in the world frame (the arm's base pose enters through its forward kinematics) */
static void ho_ik(const xo_model *m, const xo_ho_cfg *c, int arm, const real *q9, const real *target, real *q_out) {
    /* world-frame DLS (the DLS step is invariant under a rotation of the task frame); target orientation
     * quaternion (1,0,0,0) is given in the world frame (:257-258) */
    real Rb[9], pb[3], qin[XO_MAXD] = {0};
    ho_base(c, arm, Rb, pb);
    memcpy(qin, q9, 9 * sizeof(real));
    ik_solve_base(m, qin, target, c->n_ticks, q_out, Rb, pb);
}
static void ho_eef(const xo_model *m, const xo_ho_cfg *c, int arm, const real *q9, real *pos) {
    tree_t t;
    real Rb[9], pb[3], qin[XO_MAXD] = {0};
    ho_base(c, arm, Rb, pb);
    memcpy(qin, q9, 9 * sizeof(real));
    tree_setup_base(m, qin, &t, Rb, pb);
    v3_copy(pos, t.o[m->eef_link]);
}
static void ho_obs(const xo_model *m, const xo_ho_cfg *c, const real *st, real *obs, real *ag, real *dg) {
    for (int k = 0; k < 3; k++) obs[k] = st[H_BP + k];
    for (int k = 0; k < 4; k++) obs[3 + k] = st[H_BQ + k];
    for (int k = 0; k < 3; k++) { obs[7 + k] = st[H_BV + k]; obs[10 + k] = st[H_BW + k]; }
    for (int a = 0; a < 2; a++) {
        tree_t t;
        real Rb[9], pb[3], qin[XO_MAXD] = {0}, cm[3], hp[3];
        ho_base(c, a, Rb, pb);
        memcpy(qin, st + H_Q + 9 * a, 9 * sizeof(real));
        tree_setup_base(m, qin, &t, Rb, pb);
        int l = m->hand_link, d1 = t.dof[m->finger_link[0]];
        m3_vec(cm, t.R[l], m->com[l]);
        v3_add(hp, t.o[l], cm);
        for (int k = 0; k < 3; k++) {
            real dd[3] = {0, 0, 0}, J[XO_MAXD], sum = 0;
            dd[k] = 1;
            point_jacobian_row(m, &t, l, hp, dd, J);
            for (int j = 0; j < 9; j++) sum += J[j] * st[H_QD + 9 * a + j];
            obs[13 + 8 * a + k] = hp[k] - c->eef2grip[k];   /* :310-311 */
            obs[13 + 8 * a + 3 + k] = sum;
        }
        obs[13 + 8 * a + 6] = st[H_Q + 9 * a + d1];
        obs[13 + 8 * a + 7] = st[H_QD + 9 * a + d1];
    }
    for (int k = 0; k < 3; k++) { ag[k] = st[H_BP + k]; dg[k] = st[H_GOAL + k]; }
}
/* draws: 0-1 object xy, 2 mirror coin, 3-5 goal xyz, 6 same-side coin */
static void ho_draws(const xo_ho_cfg *c, int64_t env, int64_t episode, real *u) {
    uint32_t o[4];
    uint64_t gid = (uint64_t)(c->env_id_offset + env);
    for (int b = 0; b < 2; b++) {
        xo_philox(c->seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, (uint32_t)b, o);
        for (int k = 0; k < 4; k++) u[b * 4 + k] = u01(o[k]);
    }
}
static void ho_sample_object(const xo_ho_cfg *c, const real *u, real *st) {
    real x = c->obj_low[0] + u[0] * (c->obj_high[0] - c->obj_low[0]);
    st[H_BP] = u[2] < 0.5 ? -x : x;                                        /* :361-362 */
    st[H_BP + 1] = c->obj_low[1] + u[1] * (c->obj_high[1] - c->obj_low[1]);
    st[H_BP + 2] = c->height_offset;
    st[H_BQ] = st[H_BQ + 1] = st[H_BQ + 2] = 0; st[H_BQ + 3] = 1;
    for (int k = 0; k < 6; k++) st[H_BV + k] = 0;
    for (int k = 0; k < 16; k++) st[H_LT + k] = 0;
}
static void ho_sample_goal(const xo_ho_cfg *c, const real *u, real *st) {
    for (int k = 0; k < 3; k++) st[H_GOAL + k] = c->goal_low[k] + u[3 + k] * (c->goal_high[k] - c->goal_low[k]);
    int same = u[6] < c->same_side_rate;
    if ((st[H_BP] > 0) != same) st[H_GOAL] = -st[H_GOAL];                   /* (obj_x > 0) XOR same_side, :380-382 */
    if (c->goal_shape == 1) st[H_GOAL + 2] = c->height_offset;
}
int xo_ho_init(const xo_model *m, const xo_ho_cfg *c, int64_t E, double *state) {
    (void)m;
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_HO_STATE_DIM, u[8];
        memset(st, 0, XO_HO_STATE_DIM * sizeof(real));
        for (int a = 0; a < 2; a++) {
            for (int k = 0; k < 9; k++) st[H_Q + 9 * a + k] = c->joint_init_pos[k];
            st[H_FT + a] = c->joint_init_pos[7];
        }
        ho_draws(c, e, 0, u);
        ho_sample_object(c, u, st);
        ho_sample_goal(c, u, st);
    }
    return 0;
}
int xo_ho_reset(const xo_model *m, const xo_ho_cfg *c, int64_t E, double *state, const uint8_t *mask, double *obs,
                double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        real *st = state + e * XO_HO_STATE_DIM, u[8], qt[2][XO_MAXD];
        int64_t episode = (int64_t)st[H_EPISODE] + 1;
        for (int k = 0; k <= c->reset_ticks; k++) {
            if (k < c->reset_ticks) {
                for (int a = 0; a < 2; a++) {
                    ho_ik(m, c, a, st + H_Q + 9 * a, c->eff_init_pos[a], qt[a]);
                    qt[a][7] = qt[a][8] = st[H_FT + a];     /* the finger motors keep their last targets */
                }
            } else {
                ho_draws(c, e, episode, u);
                ho_sample_object(c, u, st);
            }
            ho_substep(m, c, st, qt, c->time_step);          /* one stepSimulation, :353,365 */
        }
        ho_sample_goal(c, u, st);
        st[H_STEPS] = 0;
        st[H_EPISODE] = (real)episode;
        if (obs) ho_obs(m, c, st, obs + e * XO_HO_OBS_DIM, ag + e * 3, dg + e * 3);
    }
    return 0;
}
double xo_ho_dense_reward(const double *grip1, const double *grip2, int if1, int if2, const double *ag, const double *g) {
    /* xarm_handover.py:185-199 */
    real d1v[3] = {grip1[0] - ag[0] + 0.06, grip1[1] - ag[1], grip1[2] - ag[2]};
    real d2v[3] = {grip2[0] - ag[0] - 0.06, grip2[1] - ag[1], grip2[2] - ag[2]};
    real d1 = v3_norm(d1v), d2 = v3_norm(d2v);
    if (!if1 && !if2) return 0.25 * (1 - tanh(1.0 * d1)) / 2.25;
    if (if1 && !if2) return ag[2] > 0.05 ? (1.0 + 0.25 * (1 - tanh(1.0 * d2))) / 2.25 : 0.5 / 2.25;
    if (if1 && if2) return 1.5 / 2.25;
    real dg[3];
    v3_sub(dg, ag, g);
    return (2.0 + 0.25 * (1 - tanh(1.0 * v3_norm(dg)))) / 2.25;   /* `d` undefined in the reference (:199): object-to-goal */
}
int xo_ho_compute_reward(const xo_ho_cfg *c, int64_t n, const double *ag, const double *g, double *out) {
    for (int64_t i = 0; i < n; i++) {
        real d[3];
        v3_sub(d, ag + i * 3, g + i * 3);
        out[i] = -(v3_norm(d) > c->distance_threshold ? 1.0 : 0.0);
    }
    return 0;
}
int xo_ho_step(const xo_model *m, const xo_ho_cfg *c, int64_t E, double *state, const double *actions, double *obs,
               double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success) {
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_HO_STATE_DIM, qt[2][XO_MAXD];
        const real *act = actions + e * XO_HO_ACT_DIM;
        st[H_STEPS] += 1;
        for (int a = 0; a < 2; a++) {
            real av[4], cur[3], tgt[3];
            for (int k = 0; k < 4; k++) { real v = act[a * 4 + k]; av[k] = v < -1 ? -1 : (v > 1 ? 1 : v); }   /* :129 */
            ho_eef(m, c, a, st + H_Q + 9 * a, cur);
            for (int k = 0; k < 3; k++) {
                real v = cur[k] + av[k] * c->max_vel * c->action_dt;
                tgt[k] = v < c->pos_low[a][k] ? c->pos_low[a][k] : (v > c->pos_high[a][k] ? c->pos_high[a][k] : v);
            }
            real g = st[H_Q + 9 * a + 7] + av[3] * c->action_dt * c->max_gripper_vel;
            g = g < c->gripper_low ? c->gripper_low : (g > c->gripper_high ? c->gripper_high : g);
            ho_ik(m, c, a, st + H_Q + 9 * a, tgt, qt[a]);
            qt[a][7] = qt[a][8] = g;
            st[H_FT + a] = g;
            st[H_MUG + a] = st[H_TOUCH + a];     /* friction toggle from the current contact points, :269-280 */
        }
        /* clamp the stick into the play field, keep only its pitch, zero its velocity (:282-297) */
        {
            real x = st[H_BQ], y = st[H_BQ + 1], z = st[H_BQ + 2], w = st[H_BQ + 3];
            real sarg = 2 * (w * y - x * z), pitch;
            if (sarg <= -0.99999) pitch = -0.5 * 3.14159265358979323846;
            else if (sarg >= 0.99999) pitch = 0.5 * 3.14159265358979323846;
            else pitch = asin(sarg);
            st[H_BQ] = 0; st[H_BQ + 1] = sin(0.5 * pitch); st[H_BQ + 2] = 0; st[H_BQ + 3] = cos(0.5 * pitch);
            for (int k = 0; k < 2; k++) {
                real v = st[H_BP + k], hi = c->obj_high[k];
                st[H_BP + k] = v < -hi ? -hi : (v > hi ? hi : v);
            }
            for (int k = 0; k < 6; k++) st[H_BV + k] = 0;
        }
        for (int k = 0; k < c->n_ticks; k++) ho_substep(m, c, st, qt, c->time_step);
        ho_obs(m, c, st, obs + e * XO_HO_OBS_DIM, ag + e * 3, dg + e * 3);
        real d[3];
        v3_sub(d, ag + e * 3, dg + e * 3);
        real dist = v3_norm(d);
        success[e] = (uint8_t)(dist < c->distance_threshold);
        if (c->reward_type == 1) {
            const real *o = obs + e * XO_HO_OBS_DIM;   /* hand COM - eef2grip of both arms: obs[13:16], obs[21:24] */
            reward[e] = xo_ho_dense_reward(o + 13, o + 21, st[H_MUG] > 0.5, st[H_MUG + 1] > 0.5, ag + e * 3, dg + e * 3);
        } else
            reward[e] = -(dist > c->distance_threshold ? 1.0 : 0.0);
        done[e] = (uint8_t)(success[e] || ((int)st[H_STEPS] == c->max_episode_steps));
    }
    return 0;
}

#include "xarm_oracle_stack.inc.c"
#include "xarm_oracle_handover2.inc.c"
