/*
 * xarm_hip.h - C ABI of libxarm_hip.so, the MI355X (gfx950) batched Xarm7 manipulation
 * environment.  This is the drop-in boundary for the hot path of jc-bao/gym-xarm:
 *
 *   reference interface (Python, /root/reference/gym_xarm/envs/xarm_pick_and_place.py)
 *     XarmPickAndPlace.__init__(config)            :17-103   -> xarm_create
 *     XarmPickAndPlace.reset()                     :121-127  -> xarm_reset
 *     XarmPickAndPlace.step(action)                :107-119  -> xarm_step
 *     XarmPickAndPlace.compute_reward(ag, g, info) :155-177  -> xarm_compute_reward
 *     XarmPickAndPlace.close / p.disconnect                  -> xarm_destroy
 *   and the PyBullet C-API calls those methods make per step, which this
 *   library replaces wholesale:  calculateInverseKinematics :207, setJointMotorControl2
 *   :208-211, getContactPoints :212, changeDynamics :213-218, stepSimulation :111,
 *   getLinkState/getJointStates/getBasePositionAndOrientation/getBaseVelocity :222-236.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 on success or a negative
 *     XARM_E_* code and never throws; xarm_last_error() gives the message.
 *   - all *_dev pointers are DEVICE pointers borrowed from the caller (e.g. torch tensors'
 *     data_ptr()), row-major [num_envs, dim], float32 unless stated; they must stay valid
 *     until the stream has executed the call.
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls are
 *     asynchronous, the caller synchronises.
 *   - one handle per GPU per process; a handle is not thread-safe.
 */
#ifndef XARM_HIP_H
#define XARM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XARM_OK 0
#define XARM_E_INVALID (-1)   /* bad argument / unsupported configuration */
#define XARM_E_HIP (-2)       /* a HIP runtime call failed */
#define XARM_E_NODEVICE (-3)  /* no HIP device available */

#define XARM_ENV_PICK_AND_PLACE 0 /* XarmPickAndPlace-v1 / XarmPDPickAndPlace-v0 (xarm_pick_and_place.py) */
#define XARM_ENV_REACH 1          /* XarmReach-v0 (xarm_reach.py): obs 8, 25 steps, 20 substeps of 1/4800 s;
                                     xarm_config.num_obj / goal_shape / *_rate are ignored */
#define XARM_ENV_HANDOVER 2       /* XarmHandover-v0 / XarmPDHandover-v0 (xarm_handover.py): two arms, one stick,
                                     obs 29, action 8, sparse reward -1/0, 100 steps; uses same_side_rate and
                                     goal_shape (XARM_GOAL_GROUND = 'ground', else the sampled height) */
#define XARM_ENV_STACK_TOWER 3    /* XarmPDStackTower-v0 (xarm_stack_tower.py): two arms, three cubes (num_obj = 3),
                                     obs 55, action 8, goal 9, reward_type 0 = -(d > 0.09) / 1 = -d (:124-129),
                                     50 steps (:43); step() itself never reports done (:111) */

/* auto_reset = XARM_AUTO_RESET_LAZY (PickAndPlace only; NOT the reference's semantics, opt-in for throughput): an env that
 * finishes an episode runs the reference's six reset ticks (xarm_pick_and_place.py:250-266) one per call in its next six
 * xarm_step calls instead of inside the call in which it finished.  The tick sequence, and so the state the new episode
 * starts from, is the same.  done[e] then reports the phase: 0 ordinary step, 1 the episode ended in this call (obs =
 * terminal observation), 2 reset tick (action ignored, reward 0, to be masked by the learner; the sixth such call returns
 * the first observation of the new episode).  No terminal_obs buffer is written. */
#define XARM_AUTO_RESET_LAZY 2

#define XARM_REWARD_SPARSE 0    /* (|ag-g| < 0.05) -> 1/0            :163-165 */
#define XARM_REWARD_DENSE_O2G 1 /* -|ag-g|                           :176-177 */
#define XARM_REWARD_DENSE 2     /* staged reach/grasp/lift reward    :166-175; uses the simulator's contact
                                   state, so xarm_compute_reward (relabelling) rejects it */

/* XarmReach-v0 reward types, xarm_reach.py:107-116 */
#define XARM_REACH_REWARD_SPARSE 0     /* (|ag-g| < 0.05) -> 1/0 */
#define XARM_REACH_REWARD_DENSE 1      /* -|ag-g| */
#define XARM_REACH_REWARD_DENSE_DIFF 2 /* d_old - d, stateful: xarm_compute_reward rejects it */

#define XARM_GOAL_AIR 0    /* goal_space.sample(), z -> ground w.p. goal_ground_rate  :272-280 */
#define XARM_GOAL_GROUND 1 /* shared xy, z = 0.025 (2i+1)                             :282-286 */

typedef struct xarm_config {
    int64_t num_envs;       /* environments owned by this handle (this GPU's shard) */
    int64_t env_id_offset;  /* global id of env 0 of the shard; the RNG is keyed by global id */
    uint64_t seed;
    int32_t env_kind;       /* XARM_ENV_* */
    int32_t num_obj;        /* config['num_obj']: 1; XarmHandover also 2 (the reference's test.py:9-15: obs 42, goals 6); StackTower 3 */
    int32_t reward_type;    /* XARM_REWARD_* (config['reward_type']) */
    int32_t goal_shape;     /* XARM_GOAL_*   (config['goal_shape']) */
    float init_grasp_rate;  /* config['init_grasp_rate'] */
    float goal_ground_rate; /* config['goal_ground_rate'] */
    int32_t auto_reset;     /* 1: envs that finish an episode in xarm_step are reset in the same call (the reference's
                               VecEnv semantics); XARM_AUTO_RESET_LAZY: see below */
    int32_t device;         /* HIP device ordinal */
    float same_side_rate;   /* config['same_side_rate'] (Handover, xarm_handover.py:380) */
    int32_t reset_coop_limit; /* PickAndPlace, Reach, Handover (one stick): resets of at most this many envs per call run on the cooperative
                               (16 lanes per env) kernel; 0 = default (XARM_RESET_COOP_LIMIT_DEFAULT), < 0 = never */
    int32_t step_coop_limit;  /* PickAndPlace, Reach, Handover (one stick): a handle of at most this many envs also STEPS on the cooperative kernel
                                 (the one-env-per-lane launch would leave most SIMDs without a wavefront);
                                 0 = default (XARM_STEP_COOP_LIMIT_DEFAULT), < 0 = never.  Larger PickAndPlace handles step on
                                 the pad-free fast kernel and hand the envs with an active finger-pad row to the cooperative
                                 kernel (XARM_EJECT_COOP_CAP); < 0 also pins those to the plain one-env-per-lane k_step */
    int32_t use_stand;        /* XarmHandover config['use_stand'] (xarm_handover.py:391-392): a static 0.07 x 0.06 x 0.01 box
                                 whose top sits 25 mm under the goal; 0 = parked away (the BASELINE configuration) */
} xarm_config;                /* 72 bytes */
#define XARM_RESET_COOP_LIMIT_DEFAULT 8192
/* PickAndPlace batches above step_coop_limit step on the pad-free fast kernel; the envs with an active finger-pad row
 * (~2 % in the steady state, ~16 % in the first steps after a bulk reset) are handed off to the cooperative kernel -
 * hand-offs of more than this many envs to the one-env-per-lane kernel.  The break-even is ~8 192 envs; the cap is set
 * well above it (the first steps after a bulk reset of 65 536 envs hand off ~17 000) so that the choice - which, like the
 * reset family, depends on a COUNT and therefore on the shard - is only ever crossed by configurations that start most
 * episodes in the gripper (init_grasp_rate) */
#define XARM_EJECT_COOP_CAP 32768
/* A pipelined step with auto-reset resets the episodes that ended in the fast kernel on a side stream owned by the handle,
 * in parallel with the hand-off, and joins it (event wait) before its work on the caller's stream ends: the caller sees one
 * stream-ordered call.  XARM_RESET_OVERLAP=0 in the environment at xarm_create keeps everything on the caller's stream. */
/* PickAndPlace handles of at most this many envs step on the cooperative kernel as well (env XARM_STEP_COOP_LIMIT) */
#define XARM_STEP_COOP_LIMIT_DEFAULT 8192
/* XarmHandover with one stick steps the same way at every batch size (unless step_coop_limit < 0 or XARM_STEP_PIPELINE=0 select
 * the plain lane-pair k_ho_step): the pad-free fast lane-pair kernel, then the envs with an active finger-pad row on either arm
 * (~4 %) on the cooperative rows - TWO 16-lane rows per env, one per arm (csrc/xarm_handover_coop_core.h).  Hand-offs of more
 * than XARM_HO_EJECT_COOP_CAP envs go to k_ho_step instead (2 048 cooperative wavefronts per round of ~0.5 ms against 2.3 ms);
 * resets of at most XARM_HO_RESET_COOP_LIMIT_DEFAULT envs (reset_coop_limit / XARM_RESET_COOP_LIMIT override) run on the
 * cooperative rows too. */
#define XARM_HO_EJECT_COOP_CAP 8192
#define XARM_HO_RESET_COOP_LIMIT_DEFAULT 4096
/* ... and a Handover handle of at most XARM_HO_STEP_COOP_LIMIT_DEFAULT envs (step_coop_limit / XARM_STEP_COOP_LIMIT override, as
 * for PickAndPlace) steps on the cooperative rows altogether: one launch, no fast pass */
#define XARM_HO_STEP_COOP_LIMIT_DEFAULT 2048
/* The Handover pipeline is STAGED: the fast kernel runs the 15 ticks of a step in XARM_HO_STAGES_DEFAULT launches (env
 * XARM_HO_STAGES = 1 .. 5 at xarm_create; 1 = one fast launch and one hand-off).  An env whose pads come alive in stage c keeps the
 * state it had before that stage and re-runs the ticks from the stage's first one on the cooperative rows, on a side stream
 * owned by the handle, beside the next fast stage; every side stream is joined (event wait) before the call's work on the
 * caller's stream ends.  Which kernel runs which tick of an env is a function of that env's own state and of the handle's
 * configuration, never of its neighbours. */
#define XARM_HO_STAGES_DEFAULT 3
/* The pipelined PickAndPlace step (batches above step_coop_limit) is staged the same way over its 15 substeps: XARM_PNP_STAGES_DEFAULT
 * fast launches (env XARM_PNP_STAGES = 1 .. 5 at xarm_create; 1 = one fast launch and one hand-off, round 3's pipeline). */
#define XARM_PNP_STAGES_DEFAULT 3
/* test hook: XARM_HO_FORCE_COUPLED=1 at xarm_create sends every substep of the cooperative Handover step and reset through the
 * coupled (both-arms) sweep - same bits by construction (tests/test_handover_coop.py) */

typedef struct xarm_dims_t {
    int32_t obs_dim, goal_dim, act_dim, state_dim, max_episode_steps, n_substeps;
} xarm_dims_t;

typedef struct xarm_handle xarm_handle;

int xarm_create(const xarm_config *cfg, xarm_handle **out);
int xarm_destroy(xarm_handle *h);
int xarm_dims(const xarm_handle *h, xarm_dims_t *out);

/* reset(): mask_dev (uint8 [E], nullable = all envs) selects the environments to reset; the fresh
 * observation rows of those envs are written, other rows are left untouched. */
int xarm_reset(xarm_handle *h, const uint8_t *mask_dev, float *obs_dev, float *achieved_goal_dev,
               float *desired_goal_dev, void *stream);

/* step(actions): actions [E,4]; outputs obs [E,24], achieved/desired goal [E,3], reward [E],
 * done / is_success uint8 [E].  With auto_reset, rows of finished envs hold the first observation
 * of the next episode and terminal_obs_dev (nullable, [E,24]) receives their last observation. */
int xarm_step(xarm_handle *h, const float *actions_dev, float *obs_dev, float *achieved_goal_dev,
              float *desired_goal_dev, float *reward_dev, uint8_t *done_dev, uint8_t *success_dev,
              float *terminal_obs_dev, void *stream);

/* compute_reward(achieved_goal, goal, info) over n rows of goal_dim floats (HER relabelling) */
int xarm_compute_reward(xarm_handle *h, const float *achieved_goal_dev, const float *goal_dev, int64_t n,
                        float *out_dev, void *stream);

/* full simulator state, row-major [E, state_dim] (layout: gym_xarm_amd/csrc/xarm_core.h S_*);
 * used for parity injection and snapshots */
int xarm_get_state(xarm_handle *h, float *state_dev, void *stream);
int xarm_set_state(xarm_handle *h, const float *state_dev, void *stream);

/* steps taken in the current episode, int32 [E]; XarmReachEnv's info['future_length'] (xarm_reach.py:90) is
 * max_episode_steps - steps */
int xarm_episode_steps(xarm_handle *h, int32_t *steps_dev, void *stream);

/* test hook: advance every env by n internal substeps (dt = 1/900 s) toward the joint targets
 * qtarget_dev [E,9]; no action / IK / observation logic.  Used by the substep-level parity tests. */
int xarm_debug_substeps(xarm_handle *h, const float *qtarget_dev, int32_t n, void *stream);

/* optional kernel timing: HIP events recorded around the step kernel on the caller's stream */
int xarm_timing_enable(xarm_handle *h, int32_t enable);
int xarm_timing_read(xarm_handle *h, double *step_kernel_ms_total, int64_t *launches);
/* same for what follows the step kernel(s) on the caller's stream inside xarm_step (auto_reset): the reset kernels - and,
 * for a pipelined PickAndPlace call, only what is LEFT of them after the hand-off (the first reset launch runs on the
 * handle's side stream beside the hand-off; the join is inside this bracket).  Total ms over `launches` calls */
int xarm_timing_read_reset(xarm_handle *h, double *reset_kernels_ms_total, int64_t *launches);

/* the limits in force for this handle.  Precedence: an explicit xarm_config value (> 0, or < 0 = never) wins; with the
 * field at 0 the XARM_RESET_COOP_LIMIT / XARM_STEP_COOP_LIMIT environment variable replaces the built-in default: batches / reset lists of at most that many envs run on the cooperative kernels (0: never) */
int xarm_kernel_limits(const xarm_handle *h, int32_t *reset_coop_limit, int32_t *step_coop_limit);

/* which launches an xarm_step call of this handle is made of (what bench.py names the kernels from, instead of re-deriving it
 * from environment variables) and the solver constants the library was BUILT with:
 *   fast_pipeline    1: the pad-free fast step + the hand-off of the envs with an active finger-pad row to the cooperative
 *                    kernel (PickAndPlace batches above step_coop_limit, XarmHandover with one stick at every batch size);
 *                    0: one step kernel (the cooperative one for batches of at most step_coop_limit envs).  Only
 *                    step_coop_limit < 0 or XARM_STEP_PIPELINE=0 turn the pipeline off - XARM_STEP_COOP_LIMIT=0 turns off
 *                    the cooperative STEP kernel of small batches, not the pipeline of large ones
 *   reset_overlap    1: PickAndPlace pipeline with the first reset launch on the handle's side stream (XARM_RESET_OVERLAP)
 *   eject_coop_cap   hand-off lists of at most this many envs step on the cooperative kernel, longer ones on the
 *                    one-env-per-lane kernel; INT32_MAX when step_coop_limit == 1 (the pin of
 *                    gym_xarm_amd.distributed.reproducible_limits('fast'): the choice is then a function of the config alone)
 *   solver_iterations  Gauss-Seidel sweeps per substep compiled into the kernels (50 = Bullet's numSolverIterations; a
 *                    timing variant built with -DXC_SWEEP_ITERS / -DXK_SWEEP_ITERS reports its own value and
 *                    xarm_version() says "TIMING VARIANT") */
int xarm_pipeline_info(const xarm_handle *h, int32_t *fast_pipeline, int32_t *reset_overlap, int32_t *eject_coop_cap,
                       int32_t *solver_iterations);
/* The stages of the staged Handover step (XARM_HO_STAGES_DEFAULT above): *stages = their number (1 for a handle without the
 * staged pipeline), ticks[0 .. *stages] = the first tick of each stage and, last, the tick count of a step; ticks has room for
 * XARM_HO_MAX_STAGES + 1 entries */
#define XARM_HO_MAX_STAGES 5
int xarm_stage_info(const xarm_handle *h, int32_t *stages, int32_t *ticks);
/* development / test hook: the device-side counters of the LAST xarm_step call, after synchronising `stream` - episodes that
 * ended in the step kernels (a pipelined PickAndPlace call counts the ones that ended in the hand-off apart: not included) and
 * envs handed off by the fast kernel to the cooperative one (0 for a handle without the pipeline; the staged Handover step: all stages) */
int xarm_debug_counts(xarm_handle *h, int32_t *finished, int32_t *handed_off, void *stream);

/* StackTower: the row-set class each env's last substep fell into, uint8 [E] (bits 0-2 cube pairs (0,1) (0,2) (1,2) in
 * contact, bit 3 / 4 a finger pad of arm 0 / 1 active).  The step
 * kernel visits the envs grouped by this key so that a wavefront sweeps one class's rows, not the union of 32 unrelated
 * envs' (csrc/xarm_stack_core.h "class-homogeneous wavefronts"); an env's result does not depend on the order.
 * XARM_ST_CLASS_ORDER=0 in the environment at xarm_create keeps the arrival order (then, and for the other env kinds,
 * this call fails with XARM_E_INVALID).  Introspection only - nothing in the reference corresponds to it. */
int xarm_class_keys(xarm_handle *h, uint8_t *keys_dev, void *stream);

const char *xarm_last_error(const xarm_handle *h);
const char *xarm_version(void);

#ifdef __cplusplus
}
#endif
#endif
