/*
 * xarm_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C restatement of the reference hot path `XarmPickAndPlace.step / reset /
 * compute_reward` (/root/reference/gym_xarm/envs/xarm_pick_and_place.py:107-127,155-291)
 * including the physics that the reference delegates to PyBullet
 * (calculateInverseKinematics, setJointMotorControl2, createConstraint(JOINT_GEAR),
 * getContactPoints, changeDynamics, stepSimulation — SURVEY.md §8a rows a3-a8).
 *
 * PARITY STATUS
 *   - glue / observation layout / rewards / done: pinned by golden vectors generated from
 *     the reference's own NumPy code (tests/golden/, tools/gen_golden.py) and by the URDF
 *     known-answer FK values of SURVEY.md §8c.
 *   - physics (everything PyBullet does): PARITY UNPINNED. pybullet 3.x (unpinned in the
 *     reference's setup.py:17) is not installed and cannot be fetched; the algorithm below
 *     restates Bullet's published multibody pipeline (Featherstone ABA in link coordinates,
 *     per-row Jacobian + unit-impulse response, projected Gauss-Seidel with 50 iterations,
 *     speculative contacts with ERP, velocity-level PD motors as solver rows, gear row,
 *     joint-limit rows, semi-implicit Euler) with every constant listed in
 *     gym_xarm_amd/model/xarm7_pd.json["solver"].
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (gym_xarm_amd) never links, imports or executes anything in oracle/.
 */
#ifndef XARM_ORACLE_H
#define XARM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XO_MAXL 16 /* links  */
#define XO_MAXD 13 /* joint dofs */
#define XO_NPAD 2  /* pad spheres per finger */

/* state / observation widths for PickAndPlace with one object (row-major [E, width]) */
#define XO_STATE_DIM 54
#define XO_OBS_DIM 24
#define XO_GOAL_DIM 3
#define XO_ACT_DIM 4

typedef struct {
    int32_t n_links;
    int32_t parent[XO_MAXL];
    int32_t jtype[XO_MAXL]; /* 0 fixed, 1 revolute, 2 prismatic */
    int32_t eef_link, hand_link, finger_link[2];
    double org_p[XO_MAXL][3];
    double org_rpy[XO_MAXL][3];
    double axis[XO_MAXL][3];
    double lower[XO_MAXL], upper[XO_MAXL], damping[XO_MAXL];
    double mass[XO_MAXL];
    double com[XO_MAXL][3];
    double inertia[XO_MAXL][6]; /* ixx ixy ixz iyy iyz izz about COM */
    double pad_radius;
    double pad_center_left[XO_NPAD][3];
    /* solver */
    double gravity, contact_erp, contact_margin, solver_margin, warmstart, motor_kp, motor_kd, arm_motor_force;
    double gear_erp, gear_max_force, global_erp;
    double finger_contact_stiffness, finger_contact_damping, object_contact_damping;
    double lin_damping, ang_damping, ik_lambda, ik_residual, ik_max_dtheta, limit_window;
    double mu_object, mu_table, mu_finger, mu_finger_grasp;
    int32_t num_iterations;
    int32_t _pad0;
    /* table */
    double table_half_x, table_half_y, table_top_z;
    /* pick and place */
    double time_step, action_dt, max_vel, max_gripper_vel;
    double pos_low[3], pos_high[3], goal_low[3], goal_high[3], obj_low[2], obj_high[2];
    double gripper_low, gripper_high, height_offset, start_gripper_pos[3], reset_finger_target;
    double finger_motor_force, distance_threshold;
    double obj_half[3], obj_mass, eef2grip[3];
    int32_t n_substeps, reset_ticks, max_episode_steps, _pad1;
} xo_model;

typedef struct {
    uint64_t seed;
    int64_t env_id_offset;     /* global id of env 0 of this shard */
    double init_grasp_rate;
    double goal_ground_rate;
    int32_t goal_shape;        /* 0 = 'air', 1 = 'ground' */
    int32_t reward_type;       /* 0 = sparse, 1 = dense_o2g, 2 = dense (stateful, :166-175) */
} xo_pnp_cfg;

/* state row layout (doubles): q[9] qd[9] box_pos[3] box_quat_xyzw[4] box_v[3] box_w[3] goal[3]
 * lam_table[8] lam_pad[8] (slots finger*XO_NPAD+pad, the rest unused) touch mu_grasp num_steps episode */

int xo_state_dim(void);
/* zero-pose arm, box at its episode-0 spawn, goal sampled; call xo_pnp_reset afterwards */
int xo_pnp_init(const xo_model *m, const xo_pnp_cfg *cfg, int64_t E, double *state);
/* reset envs with mask[e] != 0 (mask NULL = all); writes fresh obs rows for those envs */
int xo_pnp_reset(const xo_model *m, const xo_pnp_cfg *cfg, int64_t E, double *state,
                 const uint8_t *mask, double *obs, double *ag, double *dg);
/* one env step for every env, no auto-reset */
int xo_pnp_step(const xo_model *m, const xo_pnp_cfg *cfg, int64_t E, double *state,
                const double *actions, double *obs, double *ag, double *dg, double *reward,
                uint8_t *done, uint8_t *success);
/* test hook (counterpart of xarm_debug_substeps): n internal substeps toward the joint targets q_target [E, 9] */
int xo_pnp_substeps(const xo_model *m, int64_t E, double *state, const double *q_target, int32_t n);
/* batched reward restatement: xarm_pick_and_place.py:155-177 (sparse, dense_o2g) */
int xo_pnp_compute_reward(const xo_model *m, int reward_type, int64_t n, const double *ag,
                          const double *g, double *out);
double xo_pnp_dense_reward(const xo_model *m, int if_grasp, const double *hand_com, const double *ag,
                           const double *g);
/* ---- XarmReach-v0 (xarm_reach.py), model = gym_xarm_amd/model/xarm7_reach.json ---- */
#define XO_REACH_STATE_DIM 45 /* q[13] qd[13] motor_target[13] goal[3] d_old num_steps episode */
#define XO_REACH_OBS_DIM 8
typedef struct {
    uint64_t seed;
    int64_t env_id_offset;
    int32_t reward_type; /* 0 sparse, 1 dense, 2 dense_diff (xarm_reach.py:107-116) */
    int32_t driver_link; /* link whose joint is gripper_driver_index 10 */
    double time_step, action_dt, max_vel, max_gripper_vel;
    double pos_low[3], pos_high[3], goal_low[3], goal_high[3];
    double motor_force, distance_threshold;
    double joint_init_pos[XO_MAXD];
    int32_t n_substeps, max_episode_steps;
} xo_reach_cfg;
int xo_reach_init(const xo_model *m, const xo_reach_cfg *cfg, int64_t E, double *state);
int xo_reach_reset(const xo_model *m, const xo_reach_cfg *cfg, int64_t E, double *state, const uint8_t *mask,
                   double *obs, double *ag, double *dg);
int xo_reach_step(const xo_model *m, const xo_reach_cfg *cfg, int64_t E, double *state, const double *actions,
                  double *obs, double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success,
                  int32_t *future_length);
int xo_reach_compute_reward(const xo_reach_cfg *cfg, int reward_type, int64_t n, const double *ag, const double *g,
                            double *out);
/* ---- XarmHandover-v0 / XarmPDHandover-v0 (xarm_handover.py), two xarm7_pd arms + one stick ---- */
#define XO_HO_STATE_DIM 76 /* q[2][9] qd[2][9] finger_target[2] obj_pos[3] obj_quat[4] obj_v[3] obj_w[3] goal[3]
                              lam_table[8] lam_pad[2][4] touch[2] mu_grasp[2] num_steps episode */
#define XO_HO_OBS_DIM 29
#define XO_HO_ACT_DIM 8
typedef struct {
    uint64_t seed;
    int64_t env_id_offset;
    double same_side_rate;      /* config['same_side_rate'] */
    int32_t goal_shape;         /* 1 = 'ground' (goal z = 0.025), anything else keeps the sampled z (:387-388) */
    int32_t n_ticks;            /* python loop of stepSimulation per env step (:131) */
    double time_step, action_dt, max_vel, max_gripper_vel;
    double pos_low[2][3], pos_high[2][3], goal_low[3], goal_high[3], obj_low[2], obj_high[2];
    double gripper_low, gripper_high, height_offset, eff_init_pos[2][3], joint_init_pos[9];
    double base_pos[2][3], base_yaw[2];
    double finger_motor_force, distance_threshold, obj_half[3], eef2grip[3];
    double table_x_min, table_x_max, table_half_y, ground_z; /* tables cover table_x_min <= |x| <= table_x_max */
    int32_t reset_ticks, max_episode_steps;
    int32_t reward_type;        /* 0 sparse (hard-wired in the reference, :40), 1 the staged dense reward (:184-199) */
    int32_t use_stand;          /* config['use_stand'] (:391-392): a static box under the goal */
    double stand_half[3], stand_below_goal; /* my_stand.urdf box 0.07 x 0.06 x 0.01; centre = goal - (0,0,stand_below_goal) */
    /* num_obj = 2 (xo_ho2_*): rejection sampling of the second stick / goal (:357-360, :375-379) */
    double spawn_min_dy, goal_min_dy, goal_min_obj_dist;
    int32_t sample_max_tries, _pad2;
} xo_ho_cfg;
int xo_ho_init(const xo_model *m, const xo_ho_cfg *cfg, int64_t E, double *state);
int xo_ho_reset(const xo_model *m, const xo_ho_cfg *cfg, int64_t E, double *state, const uint8_t *mask, double *obs,
                double *ag, double *dg);
int xo_ho_step(const xo_model *m, const xo_ho_cfg *cfg, int64_t E, double *state, const double *actions, double *obs,
               double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success);
/* staged dense reward of xarm_handover.py:184-199 (N = 1): grip_k = hand COM of arm k - eef2grip_offset, if_k = the grasp
 * flags _set_action set before the step's simulation (:263-264).  The reference's last branch (only arm 2 holds the
 * object) reads an undefined `d` and raises; here d = |achieved_goal - goal|, the distance the docstring's stage 7 means. */
double xo_ho_dense_reward(const double *grip1, const double *grip2, int if1, int if2, const double *ag, const double *g);
/* sparse reward of xarm_handover.py:177-183 for N = 1 over n rows */
int xo_ho_compute_reward(const xo_ho_cfg *cfg, int64_t n, const double *ag, const double *g, double *out);
/* ---- XarmHandover-v0 with config['num_obj'] = 2 (test.py:9-15; xarm_oracle_handover2.inc.c) ---- */
#define XO_HO2_STATE_DIM 100 /* q[2][9] qd[2][9] finger_target[2] obj_pos[2][3] obj_quat[2][4] obj_v[2][3] obj_w[2][3]
                                goal[2][3] lam_table[2][8] lam_pad[2][4] touch[2] mu_grasp[2] num_steps episode */
#define XO_HO2_OBS_DIM 42    /* 13 N + 16 (:325-329) */
#define XO_HO2_GOAL_DIM 6
int xo_ho2_init(const xo_model *m, const xo_ho_cfg *cfg, int64_t E, double *state);
int xo_ho2_reset(const xo_model *m, const xo_ho_cfg *cfg, int64_t E, double *state, const uint8_t *mask, double *obs,
                 double *ag, double *dg);
int xo_ho2_step(const xo_model *m, const xo_ho_cfg *cfg, int64_t E, double *state, const double *actions, double *obs,
                double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success);
/* sparse reward -sum_i [|ag_i - g_i| > thr] over n rows of 6 (:177-183) */
int xo_ho2_compute_reward(const xo_ho_cfg *cfg, int64_t n, const double *ag, const double *g, double *out);
/* ---- XarmPDStackTower-v0 (xarm_stack_tower.py), two xarm7_pd arms + three cubes ---- */
#define XO_ST_STATE_DIM 136 /* q[2][9] qd[2][9] motor_target[2][9] cube_pos[3][3] cube_quat[3][4] cube_v[3][3]
                               cube_w[3][3] goal[3][3] lam_table[3][8] lam_pad[2][4] num_steps episode */
#define XO_ST_OBS_DIM 55
#define XO_ST_ACT_DIM 8
#define XO_ST_GOAL_DIM 9
typedef struct {
    uint64_t seed;
    int64_t env_id_offset;
    int32_t reward_type;        /* 0 sparse -(d > thr), else -d (:124-129) */
    int32_t n_substeps, max_episode_steps, reserved;
    double time_step, action_dt, max_vel, max_gripper_vel;
    double pos_low[2][3], pos_high[2][3], goal_low[2], goal_high[2], obj_low[2], obj_high[2];
    double gripper_low, gripper_high, height_offset, joint_init_pos[9];
    double base_pos[2][3], base_yaw[2];
    double finger_motor_force, distance_threshold, cube_half, cube_mass;
} xo_st_cfg;
int xo_st_init(const xo_model *m, const xo_st_cfg *cfg, int64_t E, double *state);
int xo_st_reset(const xo_model *m, const xo_st_cfg *cfg, int64_t E, double *state, const uint8_t *mask, double *obs,
                double *ag, double *dg);
int xo_st_step(const xo_model *m, const xo_st_cfg *cfg, int64_t E, double *state, const double *actions, double *obs,
               double *ag, double *dg, double *reward, uint8_t *done, uint8_t *success);
int xo_st_compute_reward(const xo_st_cfg *cfg, int reward_type, int64_t n, const double *ag, const double *g, double *out);
/* box/box manifold used by the cube/cube rows: <= 4 points, normal from B to A, signed distances */
int xo_box_box(const double *pA, const double *RA, const double *hA, const double *pB, const double *RB,
               const double *hB, double margin, double *pts /*[4][3]*/, double *nrm, double *dist);
/* diagnostics used by tests */
int xo_fk(const xo_model *m, const double *q, double *link_pos /*[n_links*3]*/,
          double *link_rot /*[n_links*9]*/);
int xo_ik(const xo_model *m, const double *q, const double *target, int max_iter, double *q_out);
int xo_mass_matrix_inv(const xo_model *m, const double *q, double *minv /*[81]*/);
int xo_forward_dynamics(const xo_model *m, const double *q, const double *qd, const double *tau,
                        double *qdd);
void xo_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
