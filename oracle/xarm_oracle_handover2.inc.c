/* XarmHandover-v0 with num_obj = 2 - CPU restatement, TEST INFRASTRUCTURE ONLY.
 * Included at the end of xarm_oracle.c after xarm_oracle_stack.inc.c (it reuses that file's generic contact rows and
 * its box/box manifold, and xarm_oracle.c's tree / ABA / IK helpers).
 *
 * This is the configuration of the reference's only test (/root/reference/test.py:9-15: num_obj 2, goal_shape 'any',
 * same_side_rate 0.5, use_stand False) and of the env file's own __main__ (xarm_handover.py:448-455).  Follows
 * /root/reference/gym_xarm/envs/xarm_handover.py with config['num_obj'] = 2:
 *   step :128-139, _set_action :244-297 (grasp flags and friction toggle read contacts with legos[0] only, :263-280; the
 *   clamp loop runs over every stick, :282-297), _get_obs :299-336 (obs = pos 6, quat 8, v 6, w 6, then 8 per arm = 42;
 *   achieved_goal = the two stick positions), _reset_sim :338-368 (stick i > 0 resampled while its y is within 0.05 of an
 *   earlier stick's y, :357-360), _sample_goal :370-393 (goal i > 0 resampled while its y is within 0.08 of an earlier
 *   goal's y or its xy within 0.08 of any stick's xy, :375-379), compute_reward sparse = -sum_i [d_i > 0.05] :177-183,
 *   _is_success = AND_i (d_i < 0.05) :395-402.
 *
 * PARITY UNPINNED for the physics (PyBullet is absent).  Contact model (model JSON handover._num_obj_2_cite):
 * per stick the N = 1 support rows (corners against table tops / ground, first 4 active), stick/stick = the StackTower
 * box/box manifold with the sticks' half extents (friction mu_object^2, no warm start), every pad sphere against its
 * nearest stick.  Row order: T stick 0, T stick 1, BB, (M L G) arm 0, (M L G) arm 1, F arm 0, F arm 1.
 * Sampling: the reference loops on an unseeded global RNG without bound; here attempt k is a Philox block and the loop
 * is cut at sample_max_tries (probability < 1e-9 per reset), see ho2_sample_object / ho2_sample_goal.
 * The reward / success arithmetic IS pinned by tests/golden/handover_reward_reference.npz (reference NumPy code, N = 2). */

#define H2_NOBJ 2
enum { G_Q = 0, G_QD = 18, G_FT = 36, G_BP = 38, G_BQ = 44, G_BV = 52, G_BW = 58, G_GOAL = 64, G_LT = 70, G_LP = 86,
       G_TOUCH = 94, G_MUG = 96, G_STEPS = 98, G_EPISODE = 99 };

static void ho2_apply(const srow_t *r, real dl, real *st, real vb[H2_NOBJ][6]) {
    if (r->arm >= 0)
        for (int k = 0; k < 9; k++) st[G_QD + 9 * r->arm + k] += r->Ba[k] * dl;
    if (r->bp >= 0)
        for (int k = 0; k < 6; k++) vb[r->bp][k] += r->Bp[k] * dl;
    if (r->bn >= 0)
        for (int k = 0; k < 6; k++) vb[r->bn][k] += r->Bn[k] * dl;
}

/* one p.stepSimulation() at timeStep 1/240 (no internal substeps, :28-29,131-132) */
static void ho2_substep(const xo_model *m, const xo_ho_cfg *c, real *st, const real qt[2][XO_MAXD], real dt) {
    static const real finger_sign[2] = {1.0, -1.0};
    ssolver_t s;
    s.m = m; s.c = 0; s.nrows = 0;
    for (int a = 0; a < 2; a++) {
        real Rb[9], pb[3];
        ho_base(c, a, Rb, pb);
        tree_setup_base(m, st + G_Q + 9 * a, &s.t[a], Rb, pb);
    }
    const real *h = c->obj_half;
    const real Ib[3] = {m->obj_mass / 3.0 * (h[1] * h[1] + h[2] * h[2]), m->obj_mass / 3.0 * (h[0] * h[0] + h[2] * h[2]),
                        m->obj_mass / 3.0 * (h[0] * h[0] + h[1] * h[1])};
    real vb[H2_NOBJ][6], Rc[H2_NOBJ][9], Iw[H2_NOBJ][9];
    for (int o = 0; o < H2_NOBJ; o++) {
        for (int k = 0; k < 3; k++) { vb[o][k] = st[G_BV + 3 * o + k]; vb[o][3 + k] = st[G_BW + 3 * o + k]; }
        quat_to_m3(Rc[o], st + G_BQ + 4 * o);
        for (int r = 0; r < 3; r++)
            for (int cc = 0; cc < 3; cc++) {
                real v = 0;
                for (int k = 0; k < 3; k++) v += Rc[o][r * 3 + k] * Rc[o][cc * 3 + k] / Ib[k];
                Iw[o][r * 3 + cc] = v;
            }
    }
    s.imass = 1.0 / m->obj_mass;
    s.iinertia = 0;
    s.Iinv_w = &Iw[0][0];
    /* unconstrained motion */
    for (int a = 0; a < 2; a++) {
        real tau[XO_MAXD] = {0}, qdd[XO_MAXD], *qd = st + G_QD + 9 * a;
        for (int i = 0; i < m->n_links; i++)
            if (s.t[a].dof[i] >= 0) tau[s.t[a].dof[i]] = -m->damping[i] * qd[s.t[a].dof[i]];
        aba_forward_dynamics(m, &s.t[a], qd, tau, m->gravity, qdd);
        for (int k = 0; k < 9; k++) qd[k] += dt * qdd[k];
    }
    {
        real dl = pow(1.0 - m->lin_damping, dt), da = pow(1.0 - m->ang_damping, dt);
        for (int o = 0; o < H2_NOBJ; o++) {
            gyro_implicit(Rc[o], Ib, dt, vb[o] + 3);
            vb[o][2] -= dt * m->gravity;
            for (int k = 0; k < 3; k++) { vb[o][k] *= dl; vb[o][k + 3] *= da; }
        }
    }
    int row_t_n[H2_NOBJ * 8], row_p_n[8];
    /* (T) stick corners against the table tops (z = 0) or, over the gap / beside the tables, the ground plane */
    for (int o = 0; o < H2_NOBJ; o++) {
        const real *bp = st + G_BP + 3 * o;
        int cnt = 0;
        for (int i = 0; i < 8; i++) {
            real rl[3] = {(i & 1) ? h[0] : -h[0], (i & 2) ? h[1] : -h[1], (i & 4) ? h[2] : -h[2]}, r[3], p[3];
            m3_vec(r, Rc[o], rl);
            v3_add(p, bp, r);
            int on_table = fabs(p[0]) >= c->table_x_min && fabs(p[0]) <= c->table_x_max && fabs(p[1]) <= c->table_half_y;
            real dist = p[2] - (on_table ? m->table_top_z : c->ground_z);
            int active = dist < m->solver_margin && cnt < 4;
            row_t_n[o * 8 + i] = -1;
            if (!active) { st[G_LT + o * 8 + i] = 0; continue; }
            cnt++;
            real n[3] = {0, 0, 1};
            row_t_n[o * 8 + i] = st_add_contact(&s, -1, -1, o, -1, bp, 0, p, n, dist, dt, m->contact_erp, 0.0,
                                                m->mu_object * m->mu_table, m->warmstart * st[G_LT + o * 8 + i]);
        }
        /* (S) config['use_stand'] (:391-392): one stand per goal; the stick's most downward face against the top of either
         * stand - the one-stick model (ho_stand_points) per (stick, stand) pair, stand 0 before stand 1, after the corners
         * in the same <= 4 point manifold, no warm start */
        if (c->use_stand)
            for (int j = 0; j < H2_NOBJ; j++) {
                real pts[4][3], dd[4];
                int np = ho_stand_points(c, Rc[o], bp, h, st + G_GOAL + 3 * j, pts, dd);
                for (int v = 0; v < np; v++) {
                    real n[3] = {0, 0, 1};
                    int active = dd[v] < m->solver_margin && dd[v] > -(2 * c->stand_half[2] + 0.01) && cnt < 4;
                    if (!active) continue;
                    cnt++;
                    st_add_contact(&s, -1, -1, o, -1, bp, 0, pts[v], n, dd[v], dt, m->contact_erp, 0.0, m->mu_object * m->mu_table, 0.0);
                }
            }
    }
    /* (BB) stick 0 / stick 1 */
    {
        real pts[4][3], n[3], dist[4];
        int np = box_box(st + G_BP, Rc[0], h, st + G_BP + 3, Rc[1], h, m->solver_margin, pts, n, dist);
        for (int q = 0; q < np; q++)
            st_add_contact(&s, -1, -1, 0, 1, st + G_BP, st + G_BP + 3, pts[q], n, dist[q], dt, m->contact_erp, 0.0,
                           m->mu_object * m->mu_object, 0.0);
    }
    /* (M)(L)(G) per arm */
    for (int a = 0; a < 2; a++) {
        real *q = st + G_Q + 9 * a, *qd = st + G_QD + 9 * a;
        tree_t *tr = &s.t[a];
        for (int i = 0; i < m->n_links; i++) {
            if (tr->dof[i] < 0) continue;
            int d = tr->dof[i];
            srow_t *r = srow_new(&s);
            r->arm = a;
            r->Ja[d] = 1;
            r->vt = m->motor_kp * (qt[a][d] - q[d]) / dt + (1.0 - m->motor_kd) * qd[d];
            real force = (m->jtype[i] == 2) ? c->finger_motor_force : m->arm_motor_force;
            r->hi = force * c->time_step;
            r->lo = -r->hi;
            srow_finish(&s, r);
        }
        for (int i = 0; i < m->n_links; i++) {
            if (tr->dof[i] < 0) continue;
            int d = tr->dof[i];
            for (int side = 0; side < 2; side++) {
                real gap = side == 0 ? q[d] - m->lower[i] : m->upper[i] - q[d];
                if (gap >= m->limit_window) continue;
                srow_t *r = srow_new(&s);
                r->arm = a;
                r->Ja[d] = side == 0 ? 1.0 : -1.0;
                r->vt = gap < 0 ? -m->global_erp * gap / dt : -gap / dt;
                r->lo = 0; r->hi = 1e30;
                srow_finish(&s, r);
            }
        }
        {
            int d1 = tr->dof[m->finger_link[0]], d2 = tr->dof[m->finger_link[1]];
            srow_t *r = srow_new(&s);
            r->arm = a;
            r->Ja[d1] = 1.0;
            r->Ja[d2] = -1.0;
            r->vt = -m->gear_erp * m->global_erp * (q[d1] - q[d2]) / dt;
            r->hi = m->gear_max_force * c->time_step;
            r->lo = -r->hi;
            srow_finish(&s, r);
        }
    }
    /* (F) pads of arm 0, then of arm 1, each against its nearest stick; the grasp flag of an arm = both of its fingers
     * within the contact margin of stick 0 (getContactPoints(xarm, self.legos[0], finger), :263-264) */
    {
        real denom = dt * m->finger_contact_stiffness + m->finger_contact_damping + m->object_contact_damping;
        real cfm = (1.0 / denom) / dt, erp = dt * m->finger_contact_stiffness / denom;
        for (int a = 0; a < 2; a++) {
            int touch[2] = {0, 0};
            real mu = m->mu_object * (st[G_MUG + a] > 0.5 ? m->mu_finger_grasp : m->mu_finger);
            for (int f = 0; f < 2; f++) {
                int l = m->finger_link[f];
                for (int j = 0; j < XO_NPAD; j++) {
                    real cl[3] = {m->pad_center_left[j][0], finger_sign[f] * m->pad_center_left[j][1], m->pad_center_left[j][2]};
                    real cw[3], bd = 1e30, bn[3] = {0, 0, 1}, bpnt[3] = {0, 0, 0};
                    int bo = 0;
                    m3_vec(cw, s.t[a].R[l], cl);
                    v3_add(cw, cw, s.t[a].o[l]);
                    for (int o = 0; o < H2_NOBJ; o++) {
                        real dist, n[3], p[3];
                        int near = sphere_box(cw, m->pad_radius, st + G_BP + 3 * o, Rc[o], h, m->contact_margin, &dist, n, p);
                        if (o == 0 && near) touch[f] = 1;
                        if (dist < bd) { bd = dist; bo = o; v3_copy(bn, n); v3_copy(bpnt, p); }
                    }
                    int idx = a * 4 + f * XO_NPAD + j;
                    row_p_n[idx] = -1;
                    if (!(bd < m->solver_margin)) { st[G_LP + idx] = 0; continue; }
                    row_p_n[idx] = st_add_contact(&s, a, l, -1, bo, 0, st + G_BP + 3 * bo, bpnt, bn, bd, dt, erp, cfm, mu,
                                                  m->warmstart * st[G_LP + idx]);
                }
            }
            st[G_TOUCH + a] = (touch[0] && touch[1]) ? 1.0 : 0.0;
        }
    }
    /* warm start + PGS */
    for (int k = 0; k < s.nrows; k++)
        if (s.rows[k].lam != 0) ho2_apply(&s.rows[k], s.rows[k].lam, st, vb);
    for (int it = 0; it < m->num_iterations; it++)
        for (int k = 0; k < s.nrows; k++) {
            srow_t *r = &s.rows[k];
            if (r->normal_row >= 0) {
                real lim = r->mu * s.rows[r->normal_row].lam;
                r->lo = -lim; r->hi = lim;
            }
            real jv = 0;
            if (r->arm >= 0) for (int cc = 0; cc < 9; cc++) jv += r->Ja[cc] * st[G_QD + 9 * r->arm + cc];
            if (r->bp >= 0) for (int cc = 0; cc < 6; cc++) jv += r->Jp[cc] * vb[r->bp][cc];
            if (r->bn >= 0) for (int cc = 0; cc < 6; cc++) jv += r->Jn[cc] * vb[r->bn][cc];
            real dl = (r->vt - r->cfm * r->lam - jv) * r->inv_d, nl = r->lam + dl;
            if (nl < r->lo) nl = r->lo;
            if (nl > r->hi) nl = r->hi;
            dl = nl - r->lam;
            r->lam = nl;
            ho2_apply(r, dl, st, vb);
        }
    for (int i = 0; i < H2_NOBJ * 8; i++)
        if (row_t_n[i] >= 0) st[G_LT + i] = s.rows[row_t_n[i]].lam;
    for (int i = 0; i < 8; i++)
        if (row_p_n[i] >= 0) st[G_LP + i] = s.rows[row_p_n[i]].lam;
    /* integrate */
    for (int k = 0; k < 18; k++) st[G_Q + k] += dt * st[G_QD + k];
    for (int o = 0; o < H2_NOBJ; o++) {
        real *bp = st + G_BP + 3 * o, *bq = st + G_BQ + 4 * o;
        for (int k = 0; k < 3; k++) bp[k] += dt * vb[o][k];
        real w[3] = {vb[o][3], vb[o][4], vb[o][5]}, ang = v3_norm(w), ax[3];
        if (ang * dt > 0.7853981633974483) ang = 0.7853981633974483 / dt;
        real k = ang < 0.001 ? 0.5 * dt - dt * dt * dt * 0.020833333333 * ang * ang : sin(0.5 * ang * dt) / ang;
        v3_set(ax, w[0] * k, w[1] * k, w[2] * k);
        real cw = cos(ang * dt * 0.5), x = bq[0], y = bq[1], z = bq[2], w0 = bq[3];
        real nx = cw * x + ax[0] * w0 + ax[1] * z - ax[2] * y, ny = cw * y + ax[1] * w0 + ax[2] * x - ax[0] * z;
        real nz = cw * z + ax[2] * w0 + ax[0] * y - ax[1] * x, nw = cw * w0 - ax[0] * x - ax[1] * y - ax[2] * z;
        real inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz + nw * nw);
        bq[0] = nx * inv; bq[1] = ny * inv; bq[2] = nz * inv; bq[3] = nw * inv;
        for (int c2 = 0; c2 < 3; c2++) { st[G_BV + 3 * o + c2] = vb[o][c2]; st[G_BW + 3 * o + c2] = vb[o][c2 + 3]; }
    }
}

static void ho2_obs(const xo_model *m, const xo_ho_cfg *c, const real *st, real *obs, real *ag, real *dg) {
    /* :314-329: obj_pos 3N, obj_rot 4N, obj_velp 3N, obj_velr 3N, then per arm grip_pos 3, grip_velp 3, finger q, qd */
    for (int k = 0; k < 6; k++) { obs[k] = st[G_BP + k]; obs[14 + k] = st[G_BV + k]; obs[20 + k] = st[G_BW + k]; }
    for (int k = 0; k < 8; k++) obs[6 + k] = st[G_BQ + k];
    for (int a = 0; a < 2; a++) {
        tree_t t;
        real Rb[9], pb[3], qin[XO_MAXD] = {0}, cm[3], hp[3];
        ho_base(c, a, Rb, pb);
        memcpy(qin, st + G_Q + 9 * a, 9 * sizeof(real));
        tree_setup_base(m, qin, &t, Rb, pb);
        int l = m->hand_link, d1 = t.dof[m->finger_link[0]];
        m3_vec(cm, t.R[l], m->com[l]);
        v3_add(hp, t.o[l], cm);
        for (int k = 0; k < 3; k++) {
            real dd[3] = {0, 0, 0}, J[XO_MAXD], sum = 0;
            dd[k] = 1;
            point_jacobian_row(m, &t, l, hp, dd, J);
            for (int j = 0; j < 9; j++) sum += J[j] * st[G_QD + 9 * a + j];
            obs[26 + 8 * a + k] = hp[k] - c->eef2grip[k];   /* :310-311 */
            obs[26 + 8 * a + 3 + k] = sum;
        }
        obs[26 + 8 * a + 6] = st[G_Q + 9 * a + d1];
        obs[26 + 8 * a + 7] = st[G_QD + 9 * a + d1];
    }
    for (int k = 0; k < 6; k++) { ag[k] = st[G_BP + k]; dg[k] = st[G_GOAL + k]; }
}

static void ho2_block(const xo_ho_cfg *c, int64_t env, int64_t episode, int b, real *u) {
    uint32_t o[4];
    uint64_t gid = (uint64_t)(c->env_id_offset + env);
    xo_philox(c->seed, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)episode, (uint32_t)b, o);
    for (int k = 0; k < 4; k++) u[k] = u01(o[k]);
}
/* _reset_sim's spawn (:354-363).  Stick 0: the N = 1 draws (block 0 words 0-2).  Stick 1: attempt k = block 2 + k words
 * 0-1, the first whose y is at least spawn_min_dy away from stick 0's y (:357-360 compares y only); its mirror coin is
 * word 3 of block 1.  The y test is made on the uniforms themselves (|u_y1 - u_y0| against a float-rounded threshold),
 * so the float32 kernels and this float64 code accept exactly the same attempts. */
static void ho2_sample_object(const xo_ho_cfg *c, int64_t env, int64_t episode, real *st) {
    real u0[4], u1[4], ua[4] = {0, 0, 0, 0};
    ho2_block(c, env, episode, 0, u0);
    ho2_block(c, env, episode, 1, u1);
    const real wx = c->obj_high[0] - c->obj_low[0], wy = c->obj_high[1] - c->obj_low[1];
    const real thr = (real)(float)(c->spawn_min_dy / wy);
    real x0 = c->obj_low[0] + u0[0] * wx;
    st[G_BP] = u0[2] < 0.5 ? -x0 : x0;                                     /* :361-362 */
    st[G_BP + 1] = c->obj_low[1] + u0[1] * wy;
    int ok = 0;
    for (int k = 0; k < c->sample_max_tries && !ok; k++) {
        ho2_block(c, env, episode, 2 + k, ua);
        ok = !(fabs(ua[1] - u0[1]) < thr);
    }
    real x1 = c->obj_low[0] + ua[0] * wx, y1 = c->obj_low[1] + ua[1] * wy;
    if (!ok) {   /* every attempt rejected: shift to the nearest admissible y inside obj_space */
        real y0 = st[G_BP + 1];
        y1 = y0 + c->spawn_min_dy <= c->obj_high[1] ? y0 + c->spawn_min_dy : y0 - c->spawn_min_dy;
    }
    st[G_BP + 3] = u1[3] < 0.5 ? -x1 : x1;
    st[G_BP + 4] = y1;
    for (int o = 0; o < H2_NOBJ; o++) {
        st[G_BP + 3 * o + 2] = c->height_offset;
        st[G_BQ + 4 * o] = st[G_BQ + 4 * o + 1] = st[G_BQ + 4 * o + 2] = 0; st[G_BQ + 4 * o + 3] = 1;
    }
    for (int k = 0; k < 12; k++) st[G_BV + k] = 0;      /* G_BV and G_BW are adjacent */
    for (int k = 0; k < 24; k++) st[G_LT + k] = 0;      /* G_LT and G_LP are adjacent */
}
/* _sample_goal (:370-393).  Goal 0: the N = 1 draws (block 0 word 3, block 1 words 0-2: x y z same-side coin), no
 * rejection (:374).  Goal 1: attempt k = block 2 + sample_max_tries + k (x y z coin), the first whose y is at least
 * goal_min_dy away from goal 0's y and whose xy is at least goal_min_obj_dist away from the xy of EVERY stick - its own
 * included, with the goal's x still positive (:375-379 test before the side flip of :380-382). */
static void ho2_sample_goal(const xo_ho_cfg *c, int64_t env, int64_t episode, real *st) {
    real u0[4], u1[4], ua[4] = {0, 0, 0, 0}, g[3] = {0, 0, 0};
    ho2_block(c, env, episode, 0, u0);
    ho2_block(c, env, episode, 1, u1);
    const real w[3] = {c->goal_high[0] - c->goal_low[0], c->goal_high[1] - c->goal_low[1], c->goal_high[2] - c->goal_low[2]};
    const real thr = (real)(float)(c->goal_min_dy / w[1]);
    const real ug0[3] = {u0[3], u1[0], u1[1]};
    for (int k = 0; k < 3; k++) st[G_GOAL + k] = c->goal_low[k] + ug0[k] * w[k];
    if ((st[G_BP] > 0) != (u1[2] < c->same_side_rate)) st[G_GOAL] = -st[G_GOAL];    /* (obj_x > 0) XOR same_side, :380-382 */
    int ok = 0;
    for (int k = 0; k < c->sample_max_tries && !ok; k++) {
        ho2_block(c, env, episode, 2 + c->sample_max_tries + k, ua);
        for (int j = 0; j < 3; j++) g[j] = c->goal_low[j] + ua[j] * w[j];
        ok = !(fabs(ua[1] - ug0[1]) < thr);
        for (int o = 0; o < H2_NOBJ; o++) {
            real dx = g[0] - st[G_BP + 3 * o], dy = g[1] - st[G_BP + 3 * o + 1];
            if (sqrt(dx * dx + dy * dy) < c->goal_min_obj_dist) ok = 0;
        }
    }
    for (int j = 0; j < 3; j++) st[G_GOAL + 3 + j] = g[j];
    if ((st[G_BP + 3] > 0) != (ua[3] < c->same_side_rate)) st[G_GOAL + 3] = -st[G_GOAL + 3];
    if (c->goal_shape == 1) st[G_GOAL + 2] = st[G_GOAL + 5] = c->height_offset;   /* :387-388 */
}
int xo_ho2_init(const xo_model *m, const xo_ho_cfg *c, int64_t E, double *state) {
    (void)m;
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_HO2_STATE_DIM;
        memset(st, 0, XO_HO2_STATE_DIM * sizeof(real));
        for (int a = 0; a < 2; a++) {
            for (int k = 0; k < 9; k++) st[G_Q + 9 * a + k] = c->joint_init_pos[k];
            st[G_FT + a] = c->joint_init_pos[7];
        }
        ho2_sample_object(c, e, 0, st);
        ho2_sample_goal(c, e, 0, st);
    }
    return 0;
}
int xo_ho2_reset(const xo_model *m, const xo_ho_cfg *c, int64_t E, double *state, const uint8_t *mask, double *obs,
                 double *ag, double *dg) {
    for (int64_t e = 0; e < E; e++) {
        if (mask && !mask[e]) continue;
        real *st = state + e * XO_HO2_STATE_DIM, qt[2][XO_MAXD];
        int64_t episode = (int64_t)st[G_EPISODE] + 1;
        for (int k = 0; k <= c->reset_ticks; k++) {
            if (k < c->reset_ticks) {
                for (int a = 0; a < 2; a++) {
                    ho_ik(m, c, a, st + G_Q + 9 * a, c->eff_init_pos[a], qt[a]);
                    qt[a][7] = qt[a][8] = st[G_FT + a];     /* the finger motors keep their last targets */
                }
            } else
                ho2_sample_object(c, e, episode, st);
            ho2_substep(m, c, st, qt, c->time_step);         /* one stepSimulation, :353,365 */
        }
        ho2_sample_goal(c, e, episode, st);
        st[G_STEPS] = 0;
        st[G_EPISODE] = (real)episode;
        if (obs) ho2_obs(m, c, st, obs + e * XO_HO2_OBS_DIM, ag + e * 6, dg + e * 6);
    }
    return 0;
}
/* sparse reward of xarm_handover.py:177-183 for N = 2 over n rows of 6: -sum_i [|ag_i - g_i| > thr] */
int xo_ho2_compute_reward(const xo_ho_cfg *c, int64_t n, const double *ag, const double *g, double *out) {
    for (int64_t i = 0; i < n; i++) {
        real r = 0;
        for (int o = 0; o < H2_NOBJ; o++) {
            real d[3];
            v3_sub(d, ag + i * 6 + 3 * o, g + i * 6 + 3 * o);
            r += v3_norm(d) > c->distance_threshold ? 1.0 : 0.0;
        }
        out[i] = -r;
    }
    return 0;
}
int xo_ho2_step(const xo_model *m, const xo_ho_cfg *c, int64_t E, double *state, const double *actions, double *obs,
                double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success) {
    for (int64_t e = 0; e < E; e++) {
        real *st = state + e * XO_HO2_STATE_DIM, qt[2][XO_MAXD];
        const real *act = actions + e * XO_HO_ACT_DIM;
        st[G_STEPS] += 1;
        for (int a = 0; a < 2; a++) {
            real av[4], cur[3], tgt[3];
            for (int k = 0; k < 4; k++) { real v = act[a * 4 + k]; av[k] = v < -1 ? -1 : (v > 1 ? 1 : v); }   /* :129 */
            ho_eef(m, c, a, st + G_Q + 9 * a, cur);
            for (int k = 0; k < 3; k++) {
                real v = cur[k] + av[k] * c->max_vel * c->action_dt;
                tgt[k] = v < c->pos_low[a][k] ? c->pos_low[a][k] : (v > c->pos_high[a][k] ? c->pos_high[a][k] : v);
            }
            real g = st[G_Q + 9 * a + 7] + av[3] * c->action_dt * c->max_gripper_vel;
            g = g < c->gripper_low ? c->gripper_low : (g > c->gripper_high ? c->gripper_high : g);
            ho_ik(m, c, a, st + G_Q + 9 * a, tgt, qt[a]);
            qt[a][7] = qt[a][8] = g;
            st[G_FT + a] = g;
            st[G_MUG + a] = st[G_TOUCH + a];     /* friction toggle from the current contact points with stick 0, :269-280 */
        }
        /* clamp every stick into the play field, keep only its pitch, zero its velocity (:282-297) */
        for (int o = 0; o < H2_NOBJ; o++) {
            real *bq = st + G_BQ + 4 * o, *bp = st + G_BP + 3 * o;
            real x = bq[0], y = bq[1], z = bq[2], w = bq[3];
            real sarg = 2 * (w * y - x * z), pitch;
            if (sarg <= -0.99999) pitch = -0.5 * 3.14159265358979323846;
            else if (sarg >= 0.99999) pitch = 0.5 * 3.14159265358979323846;
            else pitch = asin(sarg);
            bq[0] = 0; bq[1] = sin(0.5 * pitch); bq[2] = 0; bq[3] = cos(0.5 * pitch);
            for (int k = 0; k < 2; k++) {
                real v = bp[k], hi = c->obj_high[k];
                bp[k] = v < -hi ? -hi : (v > hi ? hi : v);
            }
            for (int k = 0; k < 3; k++) st[G_BV + 3 * o + k] = st[G_BW + 3 * o + k] = 0;
        }
        for (int k = 0; k < c->n_ticks; k++) ho2_substep(m, c, st, qt, c->time_step);
        ho2_obs(m, c, st, obs + e * XO_HO2_OBS_DIM, ag + e * 6, dg + e * 6);
        int all = 1;
        real rew = 0;
        for (int o = 0; o < H2_NOBJ; o++) {
            real d[3];
            v3_sub(d, ag + e * 6 + 3 * o, dg + e * 6 + 3 * o);
            real dist = v3_norm(d);
            all = all && dist < c->distance_threshold;              /* :395-402 */
            rew += dist > c->distance_threshold ? 1.0 : 0.0;        /* :177-181 */
        }
        success[e] = (uint8_t)all;
        reward[e] = -rew;
        done[e] = (uint8_t)(success[e] || ((int)st[G_STEPS] == c->max_episode_steps));
    }
    return 0;
}
