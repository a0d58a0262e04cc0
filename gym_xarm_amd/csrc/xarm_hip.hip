// xarm_hip.hip - the C ABI (include/xarm_hip.h) of the batched Xarm environments: handles, kernel selection, launches.
// The gfx950 kernels live in xarm_k_*.hip, one translation unit per kernel family (prototypes: xarm_dev.h).
//
// PickAndPlace launches (xarm_k_pnp.hip, xarm_k_pnp_coop.hip; cores: xarm_core.h, xarm_coop_core.h):
//   k_step        one thread per environment, one 64-lane wavefront per workgroup.  The fused step keeps an env's whole
//                 working set on chip for all 15 substeps x 50 solver sweeps: ~450 VGPRs of per-env state / solver blocks
//                 (hence 1 wave per SIMD, __launch_bounds__(64)) plus 149 floats per env of LDS (hand Jacobian S, T =
//                 M^-1 S^T, A_hh, table slots: lane-private columns, 38 KB per workgroup).  65 536 envs = 1024 workgroups =
//                 4 per CU; workgroups never communicate, so no XCD-aware remap is needed.
//   k_step_fast + k_step_coop_list   the default for batches above 8 192 envs: the pad-free fast step, then the envs with an
//                 active finger-pad row on the cooperative core; XarmHandover the same way at every batch
//                 size (xarm_k_handover_coop.hip: k_ho_step_fast + k_ho_step_coop_list, two 16-lane rows per env).
//   k_step_coop / k_reset_coop   one environment per DPP row of 16 lanes (4 per wavefront), impulse-space sweep spread
//                 over the row: the latency-optimal form for small batches and for the resets that follow a step.
//   k_reset       the one-env-per-lane reset, for bulk resets (> coop_limit finished envs in one call).
// HBM is touched once per env step: 54 state floats in, 54 out (structure-of-arrays, lane = env, every load/store a fully
// coalesced 256-B wave access), 4 action floats in and the 24+3+3+1 output floats + 2 flag bytes out (row-major at the
// API edge, 16-B vector stores).  Episodes that end are compacted into a list (one atomic per finished env) and
// re-initialised by the reset kernel inside the same xarm_step call.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "xarm_dev.h"

using namespace xd;


// ------------------------------------------------------------------------------------- C ABI
struct xarm_handle {
    xarm_config cfg;
    KParams kp;
    int *done_list;   // [E]
    int *counters;    // [3 + 2 * NCLS], zeroed by ONE memset at the start of every step call (no host-side state: a captured
                      // step can be replayed): done_count = +0, eject_count = +1 (2), class_hist = +3
    int *done_count;  // [1] episodes that ended in this call (list A)
    int *mask_count;  // [1]
    int coop_step_limit; // PickAndPlace: batches of at most this many envs step on k_step_coop
    int fast_pipeline;   // PickAndPlace, larger batches: k_step_fast + hand-off of the envs with finger-pad rows (1) or k_step (0)
    int *eject_list;     // [E] envs handed off by k_step_fast
    int *eject_count;    // [2]: hand-off count, count of episodes that ended in the hand-off kernels
    // the two reset launches of a pipelined step (xarm_step): episodes that ended in k_step_fast are reset on `side`
    // while the hand-off still runs on the caller's stream, the few that end in the hand-off after it
    int *done_list_b;    // [E] episodes that ended in the hand-off kernels
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    int reset_overlap;
    // staged Handover step (xarm_step): the fast lane-pair kernel runs the step's 15 ticks in ho_stages launches; the envs a stage
    // hands off re-run only the ticks from that stage's first one on the cooperative rows, on a side stream beside the next stage
    static constexpr int MAX_ST = XARM_HO_MAX_STAGES;
    int ho_stages;        // 1 = one fast launch, one hand-off (round 4's first pipeline)
    int ho_tick[MAX_ST + 1]; // stage c runs the ticks [ho_tick[c], ho_tick[c + 1])
    float *ho_qt;         // [18][stride] joint targets of the step the first stage opened
    uint8_t *ho_flag;     // [stride] handed off in an earlier stage of this call
    hipStream_t st_side[MAX_ST];
    hipEvent_t st_fork[MAX_ST], st_join[MAX_ST];
    int ho_force_coupled; // test hook (XARM_HO_FORCE_COUPLED=1): every substep of the cooperative Handover step through the coupled sweep
    // StackTower: class-homogeneous wavefronts (xarm_stack_core.h class_layout); null when XARM_ST_CLASS_ORDER=0
    uint8_t *class_key;  // [E] row-set class of each env's last substep
    int *class_hist;     // [2 * NCLS] histogram, then the per-class arrival counters
    int *class_order;    // [E] slot -> env
    char err[512];
    // timing
    int timing;
    static constexpr int NEV = 1024;
    hipEvent_t ev0[NEV], ev1[NEV], ev2[NEV];   // before the step kernel, after it, after the reset kernels
    int ev_n;
    double ev_ms, ev_reset_ms;
    int64_t ev_launches;
    bool ev_created;
};

static char g_err[512] = "";

static int fail(xarm_handle *h, int code, const char *fmt, const char *detail) {
    char *dst = h ? h->err : g_err;
    snprintf(dst, 512, fmt, detail);
    return code;
}
#define HIPCHK(h, call)                                                          \
    do {                                                                         \
        hipError_t _e = (call);                                                  \
        if (_e != hipSuccess) return fail(h, XARM_E_HIP, #call ": %s", hipGetErrorString(_e)); \
    } while (0)

// grid of a cooperative Handover launch over at most `cap` envs: two envs per wavefront, grid stride beyond 2 048 workgroups
static unsigned ho_coop_grid(int64_t cap) {
    const int64_t g = (cap + xhc::ROW_ENVS - 1) / xhc::ROW_ENVS;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
// XarmHandover.step of the envs list[0 .. *count) (null: all) on the cooperative rows, from tick stage.tick0 on
static void launch_ho_coop_step(xarm_handle *h, unsigned g_, const float *actions_dev, float *obs_dev, float *ag_dev, float *dg_dev, float *reward_dev,
                                uint8_t *done_dev, uint8_t *success_dev, float *terminal_obs_dev, int *done_list, int *done_count,
                                const int *list, const int *count, HoStage stage, hipStream_t st) {
    if (h->kp.hcfg.use_stand) k_ho_step_coop_list<xh::HandoverStandScene, false><<<dim3(g_), dim3(WG), 0, st>>>(
        h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, done_list, done_count, list, count, stage);
    else if (h->ho_force_coupled) k_ho_step_coop_list<xh::HandoverScene, true><<<dim3(g_), dim3(WG), 0, st>>>(
        h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, done_list, done_count, list, count, stage);
    else k_ho_step_coop_list<xh::HandoverScene, false><<<dim3(g_), dim3(WG), 0, st>>>(
        h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, done_list, done_count, list, count, stage);
}
// Handover reset of the envs in list[0 .. *count) (null: all): the cooperative rows take counts up to kp.coop_limit, the lane-pair
// kernel the rest; both are launched, the one out of its range exits at once (the count lives on the device)
static void launch_ho_reset(xarm_handle *h, unsigned grid2, const int *list, const int *count, float *obs_dev, float *ag_dev, float *dg_dev,
                            hipStream_t st) {
    if (h->cfg.num_obj == 2) {
        if (h->kp.hcfg.use_stand) k_ho2_reset<xh::HandoverStandScene><<<dim3(grid2), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
        else k_ho2_reset<xh::HandoverScene><<<dim3(grid2), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
        return;
    }
    const int64_t cap = h->kp.num_envs < (int64_t)h->kp.coop_limit ? h->kp.num_envs : (int64_t)h->kp.coop_limit;
    if (cap > 0) {
        if (h->kp.hcfg.use_stand) k_ho_reset_coop<xh::HandoverStandScene, false><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
        else if (h->ho_force_coupled) k_ho_reset_coop<xh::HandoverScene, true><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
        else k_ho_reset_coop<xh::HandoverScene, false><<<dim3(ho_coop_grid(cap)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    }
    if (h->kp.num_envs > cap) {
        if (h->kp.hcfg.use_stand) k_ho_reset<xh::HandoverStandScene><<<dim3(grid2), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
        else k_ho_reset<xh::HandoverScene><<<dim3(grid2), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    }
}

static void launch_reach_reset(xarm_handle *h, const int *list, const int *count, float *obs_dev, float *ag_dev, float *dg_dev, hipStream_t st) {
    const int64_t cap = h->kp.num_envs < (int64_t)h->kp.coop_limit ? h->kp.num_envs : (int64_t)h->kp.coop_limit;
    if (cap > 0)
        k_reach_reset_coop<<<dim3((unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    if (h->kp.num_envs > cap)
        k_reach_reset<<<dim3((unsigned)(h->kp.stride / WG)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
}

// PickAndPlace reset of the envs in list[0 .. *count) (null: all): the cooperative kernel takes counts up to
// kp.coop_limit, the one-env-per-lane kernel the rest; both are launched, the one out of its range exits at once.
static void launch_pnp_reset(xarm_handle *h, const int *list, const int *count, float *obs_dev, float *ag_dev, float *dg_dev,
                             hipStream_t st) {
    const int64_t cap = h->kp.num_envs < (int64_t)h->kp.coop_limit ? h->kp.num_envs : (int64_t)h->kp.coop_limit;
    if (cap > 0)
        k_reset_coop<<<dim3((unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
    if (h->kp.num_envs > cap)
        k_reset<<<dim3((unsigned)(h->kp.stride / WG)), dim3(WG), 0, st>>>(h->kp, list, count, obs_dev, ag_dev, dg_dev);
}

// The handle's kernels, events and buffers live on cfg.device.  Every entry point makes that device current for its
// own duration and restores the caller's (torch's) current device on return, so a handle can be created and used
// while another device is current, and several handles on different GPUs can share a process.
struct DeviceGuard {
    int prev;
    bool switched;
    explicit DeviceGuard(int dev) : prev(-1), switched(false) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) hipSetDevice(prev); }
};
#define DEVGUARD(h) DeviceGuard _guard((h)->cfg.device)

static void timing_flush(xarm_handle *h) {
    for (int i = 0; i < h->ev_n; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(h->ev2[i]) == hipSuccess && hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]) == hipSuccess) {
            h->ev_ms += ms;
            h->ev_launches++;
            if (hipEventElapsedTime(&ms, h->ev1[i], h->ev2[i]) == hipSuccess) h->ev_reset_ms += ms;
        }
    }
    h->ev_n = 0;
}

extern "C" {

// a timing variant built with -DXC_SWEEP_ITERS=n (tools/coop_split.sh) runs fewer solver sweeps: it must never pass for the product
#define XARM_STR2(x) #x
#define XARM_STR(x) XARM_STR2(x)
#ifdef XARM_SWEEP_VARIANT
const char *xarm_version(void) { return "xarm_hip 0.1 (gfx950) TIMING VARIANT sweeps=" XARM_STR(XARM_SWEEP_VARIANT); }
#else
const char *xarm_version(void) { return "xarm_hip 0.1 (gfx950)"; }
#endif

const char *xarm_last_error(const xarm_handle *h) { return h ? h->err : g_err; }

int xarm_create(const xarm_config *cfg, xarm_handle **out) {
    if (!cfg || !out) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: null argument");
    const bool reach = cfg->env_kind == XARM_ENV_REACH, handover = cfg->env_kind == XARM_ENV_HANDOVER, stack = cfg->env_kind == XARM_ENV_STACK_TOWER;
    if (cfg->env_kind != XARM_ENV_PICK_AND_PLACE && !reach && !handover && !stack) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: unsupported env_kind");
    if (stack && cfg->num_obj != 3) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmStackTower has num_obj == 3 (xarm_stack_tower.py:19)");
    if (stack && cfg->reward_type > 1) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmStackTower reward_type is 0 (sparse) or 1 (-d)");
    if (handover && cfg->num_obj != 1 && cfg->num_obj != 2) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmHandover supports num_obj 1 or 2");
    if (handover && cfg->num_obj == 2 && cfg->reward_type != 0)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmHandover with num_obj == 2 takes the sparse reward (the reference's dense branch raises a broadcast error there, xarm_handover.py:187-188)");
    if (!reach && !stack && !handover && cfg->num_obj != 1)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmPickAndPlace supports num_obj == 1 (with more the reference's own step raises, xarm_pick_and_place.py:289-291)");
    if (handover && cfg->reward_type != 0 && cfg->reward_type != XARM_REWARD_DENSE)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: XarmHandover reward_type is sparse (hard-wired in the reference, xarm_handover.py:40) or dense (:184-199)");
    if (cfg->use_stand && !handover) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: use_stand belongs to XarmHandover (xarm_handover.py:391-392)");
    if (cfg->auto_reset < 0 || cfg->auto_reset > XARM_AUTO_RESET_LAZY) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: auto_reset must be 0, 1 or XARM_AUTO_RESET_LAZY");
    if (cfg->auto_reset == XARM_AUTO_RESET_LAZY && cfg->env_kind != XARM_ENV_PICK_AND_PLACE)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: lazy auto-reset is implemented for XarmPickAndPlace only");
    if (cfg->num_envs <= 0 || cfg->num_envs > (int64_t)1 << 30) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: num_envs out of range");
    if (cfg->reward_type < 0 || cfg->reward_type > 2) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: unsupported reward_type");
    if (!reach && !stack && cfg->goal_shape != XARM_GOAL_AIR && cfg->goal_shape != XARM_GOAL_GROUND)
        return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: unsupported goal_shape");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, XARM_E_NODEVICE, "%s", "xarm_create: no HIP device");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: bad device ordinal");
    DeviceGuard _guard(cfg->device);
    xarm_handle *h = new (std::nothrow) xarm_handle();
    if (!h) return fail(nullptr, XARM_E_INVALID, "%s", "xarm_create: out of host memory");
    memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    const int64_t E = cfg->num_envs, stride = (E + WG - 1) / WG * WG;
    h->kp.stride = stride;
    h->kp.num_envs = E;
    h->kp.cfg.seed = cfg->seed;
    h->kp.cfg.env_id_offset = cfg->env_id_offset;
    h->kp.cfg.init_grasp_rate = cfg->init_grasp_rate;
    h->kp.cfg.goal_ground_rate = cfg->goal_ground_rate;
    h->kp.cfg.goal_shape = cfg->goal_shape;
    h->kp.cfg.reward_type = cfg->reward_type;
    h->kp.auto_reset = cfg->auto_reset;
    // cooperative reset kernel (PickAndPlace): default cross-over measured on MI355X; 0 disables it
    h->kp.coop_limit = 0;
    if (cfg->env_kind == XARM_ENV_PICK_AND_PLACE || cfg->env_kind == XARM_ENV_REACH) {
        h->kp.coop_limit = cfg->reset_coop_limit > 0 ? cfg->reset_coop_limit : (cfg->reset_coop_limit < 0 ? 0 : XARM_RESET_COOP_LIMIT_DEFAULT);
        // the environment variable replaces the DEFAULT only: an explicit xarm_config value (incl. "< 0 = never") wins
        const char *ev = getenv("XARM_RESET_COOP_LIMIT");
        if (ev && *ev && cfg->reset_coop_limit == 0) h->kp.coop_limit = atoi(ev) > 0 ? atoi(ev) : 0;
    }
    // cooperative step kernel: pays while the one-env-per-lane launch would leave SIMDs empty (measured cross-over,
    // DESIGN.md 5); XARM_STEP_COOP_LIMIT overrides, 0 disables
    h->coop_step_limit = (cfg->env_kind != XARM_ENV_PICK_AND_PLACE && cfg->env_kind != XARM_ENV_REACH) || cfg->step_coop_limit < 0 ? 0 :
                         (cfg->step_coop_limit > 0 ? cfg->step_coop_limit : XARM_STEP_COOP_LIMIT_DEFAULT);
    {
        const char *ev = getenv("XARM_STEP_COOP_LIMIT");
        if (ev && *ev && cfg->step_coop_limit == 0 && (cfg->env_kind == XARM_ENV_PICK_AND_PLACE || cfg->env_kind == XARM_ENV_REACH))
            h->coop_step_limit = atoi(ev) > 0 ? atoi(ev) : 0;
    }
    const bool handover2 = handover && cfg->num_obj == 2;
    const bool handover1 = handover && !handover2;
    // Handover (one stick): small batches step on the cooperative rows altogether - 2 048 envs are one round of 1 024 wavefronts
    // and one kernel of ~0.6 ms, where the fast lane-pair kernel (0.72 ms whatever the batch, a latency) plus the hand-off take 1.4
    if (handover1 && cfg->step_coop_limit >= 0) {
        h->coop_step_limit = cfg->step_coop_limit > 0 ? cfg->step_coop_limit : XARM_HO_STEP_COOP_LIMIT_DEFAULT;
        const char *ev = getenv("XARM_STEP_COOP_LIMIT");
        if (ev && *ev && cfg->step_coop_limit == 0) h->coop_step_limit = atoi(ev) > 0 ? atoi(ev) : 0;
    }
    // cooperative reset of Handover (one stick): two rows per env, measured cross-over against the lane-pair reset
    if (handover1) {
        h->kp.coop_limit = cfg->reset_coop_limit > 0 ? cfg->reset_coop_limit : (cfg->reset_coop_limit < 0 ? 0 : XARM_HO_RESET_COOP_LIMIT_DEFAULT);
        const char *ev = getenv("XARM_RESET_COOP_LIMIT");
        if (ev && *ev && cfg->reset_coop_limit == 0) h->kp.coop_limit = atoi(ev) > 0 ? atoi(ev) : 0;
    }
    // fast-step pipeline (PickAndPlace batches above the cooperative limit, Handover with one stick): on unless the caller
    // pinned the one-env-per-lane family (step_coop_limit < 0: gym_xarm_amd.distributed.reproducible_limits('lane')) or
    // XARM_STEP_PIPELINE=0 asks for the plain k_step / k_ho_step
    h->fast_pipeline = ((cfg->env_kind == XARM_ENV_PICK_AND_PLACE || handover1) && cfg->step_coop_limit >= 0 && cfg->auto_reset != XARM_AUTO_RESET_LAZY) ? 1 : 0;
    {
        const char *ev = getenv("XARM_STEP_PIPELINE");
        if (ev && *ev) h->fast_pipeline = h->fast_pipeline && atoi(ev) != 0;
        ev = getenv("XARM_HO_FORCE_COUPLED");
        h->ho_force_coupled = ev && *ev && atoi(ev) != 0;
        // staged Handover step: XARM_HO_STAGES=1 is the unstaged pipeline (one fast launch, one hand-off)
        h->ho_stages = handover1 && h->fast_pipeline ? XARM_HO_STAGES_DEFAULT : 1;
        ev = getenv("XARM_HO_STAGES");
        if (ev && *ev && handover1 && h->fast_pipeline) h->ho_stages = atoi(ev) < 1 ? 1 : (atoi(ev) > xarm_handle::MAX_ST ? xarm_handle::MAX_ST : atoi(ev));
        // PickAndPlace: the same staging of the 15 substeps (XARM_PNP_STAGES; 1 = unstaged)
        const bool pnp_pipe = cfg->env_kind == XARM_ENV_PICK_AND_PLACE && h->fast_pipeline;
        if (pnp_pipe) h->ho_stages = XARM_PNP_STAGES_DEFAULT;
        ev = getenv("XARM_PNP_STAGES");
        if (ev && *ev && pnp_pipe) h->ho_stages = atoi(ev) < 1 ? 1 : (atoi(ev) > xarm_handle::MAX_ST ? xarm_handle::MAX_ST : atoi(ev));
        static_assert(xm::HO_N_TICKS == xm::PNP_N_SUBSTEPS, "one stage table for both");
        for (int c = 0; c <= h->ho_stages; c++) h->ho_tick[c] = c * xm::HO_N_TICKS / h->ho_stages;
        // measurement hook: XARM_HO_STAGE_TICKS="3,9" = the interior stage boundaries (increasing, inside 1 .. 14)
        ev = getenv("XARM_HO_STAGE_TICKS");
        if (ev && *ev && h->ho_stages > 1) {
            int c = 1, prev = 0;
            const char *q = ev;
            while (*q && c < h->ho_stages) {
                const int v = atoi(q);
                if (v <= prev || v >= xm::HO_N_TICKS) break;
                h->ho_tick[c++] = prev = v;
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
            if (c != h->ho_stages) for (int k = 0; k <= h->ho_stages; k++) h->ho_tick[k] = k * xm::HO_N_TICKS / h->ho_stages;   // malformed: the default
        }
    }
    h->kp.eject_coop_cap = handover1 ? XARM_HO_EJECT_COOP_CAP : XARM_EJECT_COOP_CAP;
    // step_coop_limit == 1 is the pin of reproducible_limits('fast'): every hand-off list steps on the cooperative kernel (it
    // walks the list with a grid stride), so that which kernel steps an env is a function of the handle's config alone
    if (cfg->step_coop_limit == 1) h->kp.eject_coop_cap = 0x7fffffff;
    h->kp.state_dim = reach ? xr::STATE_DIM : (handover2 ? xh2::STATE_DIM : (handover ? xh::STATE_DIM : (stack ? xs::STATE_DIM : xk::STATE_DIM)));
    h->kp.hcfg.seed = cfg->seed;
    h->kp.hcfg.env_id_offset = cfg->env_id_offset;
    h->kp.hcfg.same_side_rate = cfg->same_side_rate;
    h->kp.hcfg.goal_shape = cfg->goal_shape;
    h->kp.hcfg.use_stand = handover && cfg->use_stand ? 1 : 0;
    h->kp.hcfg.reward_type = cfg->reward_type == XARM_REWARD_DENSE ? 1 : 0;
    h->kp.rcfg.seed = cfg->seed;
    h->kp.rcfg.env_id_offset = cfg->env_id_offset;
    h->kp.rcfg.reward_type = cfg->reward_type;
    hipError_t e1 = hipMalloc(&h->kp.state, sizeof(float) * h->kp.state_dim * stride);
    hipError_t e2 = hipMalloc(&h->done_list, sizeof(int) * stride);
    hipError_t e3 = hipMalloc(&h->counters, sizeof(int) * (3 + 2 * xs::NCLS));   // (Handover: +3 + c = hand-off count of stage c >= 1)
    h->done_count = h->counters; h->eject_count = h->counters + 1; h->class_hist = h->counters + 3;
    hipError_t e4 = hipMalloc(&h->mask_count, sizeof(int));
    if (e4 == hipSuccess && h->fast_pipeline) {
        e4 = hipMalloc(&h->eject_list, sizeof(int) * stride * h->ho_stages);   // one list per stage
        if (e4 == hipSuccess) e4 = hipMalloc(&h->done_list_b, sizeof(int) * stride);
        const char *ev = getenv("XARM_RESET_OVERLAP");
        // (PickAndPlace only: Handover's reset is six single-substep ticks, 0.28 ms whether it runs beside the hand-off or after
        // it, and the two launches slow each other down when they overlap - 1.78 against 1.76 ms per call)
        if (e4 == hipSuccess && cfg->auto_reset && !handover1 && !(ev && *ev && atoi(ev) == 0)) {
            e4 = hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking);
            if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
            if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
            h->reset_overlap = e4 == hipSuccess;
        }
    }
    if (e4 == hipSuccess && h->fast_pipeline && h->ho_stages > 1) {
        e4 = hipMalloc(&h->ho_qt, sizeof(float) * 18 * stride);
        if (e4 == hipSuccess) e4 = hipMalloc(&h->ho_flag, stride);
        if (e4 == hipSuccess) e4 = hipMemset(h->ho_flag, 0, stride);
        for (int c = 0; c + 1 < h->ho_stages && e4 == hipSuccess; c++) {
            const char *pv = getenv("XARM_HO_SIDE_PRIO");
            int lo = 0, hi = 0;
            hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = the LEAST urgent (numerically greatest)
            if (pv && *pv && atoi(pv) != 0) e4 = hipStreamCreateWithPriority(&h->st_side[c], hipStreamNonBlocking, lo);
            else e4 = hipStreamCreateWithFlags(&h->st_side[c], hipStreamNonBlocking);
            if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&h->st_fork[c], hipEventDisableTiming);
            if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&h->st_join[c], hipEventDisableTiming);
        }
    }
    if (e4 == hipSuccess && stack) {
        const char *ev = getenv("XARM_ST_CLASS_ORDER");
        if (!(ev && *ev && atoi(ev) == 0)) {
            e4 = hipMalloc(&h->class_key, stride);
            if (e4 == hipSuccess) e4 = hipMalloc(&h->class_order, sizeof(int) * stride);
            if (e4 == hipSuccess) e4 = hipMemset(h->class_key, 0, stride);
        }
    }
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
        fail(nullptr, XARM_E_HIP, "xarm_create: hipMalloc failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : (e3 != hipSuccess ? e3 : e4))));
        xarm_destroy(h);
        return XARM_E_HIP;
    }
    hipMemset(h->kp.state, 0, sizeof(float) * h->kp.state_dim * stride);
    hipMemset(h->counters, 0, sizeof(int) * (3 + 2 * xs::NCLS));
    hipMemset(h->mask_count, 0, sizeof(int));
    if (reach) k_reach_init<<<dim3((unsigned)(stride / WG)), dim3(WG)>>>(h->kp);
    else if (handover2) k_ho2_init<<<dim3((unsigned)(2 * stride / WG)), dim3(WG)>>>(h->kp);
    else if (handover) k_ho_init<<<dim3((unsigned)(2 * stride / WG)), dim3(WG)>>>(h->kp);
    else if (stack) k_st_init<<<dim3((unsigned)(2 * stride / WG)), dim3(WG)>>>(h->kp);
    else k_init<<<dim3((unsigned)(stride / WG)), dim3(WG)>>>(h->kp);
    hipError_t e5 = hipDeviceSynchronize();
    if (e5 != hipSuccess) {
        fail(nullptr, XARM_E_HIP, "xarm_create: k_init: %s", hipGetErrorString(e5));
        xarm_destroy(h);
        return XARM_E_HIP;
    }
    *out = h;
    return XARM_OK;
}

int xarm_destroy(xarm_handle *h) {
    if (!h) return XARM_OK;
    DEVGUARD(h);
    hipDeviceSynchronize();
    if (h->ev_created)
        for (int i = 0; i < xarm_handle::NEV; i++) { hipEventDestroy(h->ev0[i]); hipEventDestroy(h->ev1[i]); hipEventDestroy(h->ev2[i]); }
    if (h->kp.state) hipFree(h->kp.state);
    if (h->done_list) hipFree(h->done_list);
    if (h->counters) hipFree(h->counters);
    if (h->mask_count) hipFree(h->mask_count);
    if (h->eject_list) hipFree(h->eject_list);
    if (h->done_list_b) hipFree(h->done_list_b);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->side) hipStreamDestroy(h->side);
    if (h->class_key) hipFree(h->class_key);
    if (h->class_order) hipFree(h->class_order);
    if (h->ho_qt) hipFree(h->ho_qt);
    if (h->ho_flag) hipFree(h->ho_flag);
    for (int c = 0; c < xarm_handle::MAX_ST; c++) {
        if (h->st_fork[c]) hipEventDestroy(h->st_fork[c]);
        if (h->st_join[c]) hipEventDestroy(h->st_join[c]);
        if (h->st_side[c]) hipStreamDestroy(h->st_side[c]);
    }
    delete h;
    return XARM_OK;
}

int xarm_dims(const xarm_handle *h, xarm_dims_t *out) {
    if (!out) return XARM_E_INVALID;
    const bool reach = h && h->cfg.env_kind == XARM_ENV_REACH, handover = h && h->cfg.env_kind == XARM_ENV_HANDOVER;
    const bool stack = h && h->cfg.env_kind == XARM_ENV_STACK_TOWER;
    const bool handover2 = handover && h->cfg.num_obj == 2;
    out->obs_dim = reach ? xr::OBS_DIM : (handover2 ? xh2::OBS_DIM : (handover ? xh::OBS_DIM : (stack ? xs::OBS_DIM : xk::OBS_DIM)));
    out->goal_dim = stack ? xs::GOAL_DIM : (handover2 ? xh2::GOAL_DIM : xk::GOAL_DIM);
    out->act_dim = handover ? xh::ACT_DIM : (stack ? xs::ACT_DIM : xk::ACT_DIM);
    out->state_dim = reach ? xr::STATE_DIM : (handover2 ? xh2::STATE_DIM : (handover ? xh::STATE_DIM : (stack ? xs::STATE_DIM : xk::STATE_DIM)));
    out->max_episode_steps = reach ? xmr::MAX_EPISODE_STEPS : (handover ? xm::HO_MAX_EPISODE_STEPS : (stack ? xm::ST_MAX_EPISODE_STEPS : xm::PNP_MAX_EPISODE_STEPS));
    out->n_substeps = reach ? xmr::N_SUBSTEPS : (handover ? xm::HO_N_TICKS : (stack ? xm::ST_N_SUBSTEPS : xm::PNP_N_SUBSTEPS));
    return XARM_OK;
}

int xarm_reset(xarm_handle *h, const uint8_t *mask_dev, float *obs_dev, float *ag_dev, float *dg_dev, void *stream) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (obs_dev && (!ag_dev || !dg_dev)) return fail(h, XARM_E_INVALID, "%s", "xarm_reset: goal buffers required with obs");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(h->kp.stride / WG);
    if (mask_dev) {
        HIPCHK(h, hipMemsetAsync(h->mask_count, 0, sizeof(int), st));
        k_compact_mask<<<dim3((unsigned)((h->kp.num_envs + 255) / 256)), dim3(256), 0, st>>>(mask_dev, h->kp.num_envs, h->done_list, h->mask_count);
        if (h->cfg.env_kind == XARM_ENV_REACH) launch_reach_reset(h, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_HANDOVER) launch_ho_reset(h, 2 * grid, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_STACK_TOWER) k_st_reset<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, h->class_key);
        else launch_pnp_reset(h, h->done_list, h->mask_count, obs_dev, ag_dev, dg_dev, st);
    } else {
        if (h->cfg.env_kind == XARM_ENV_REACH) launch_reach_reset(h, nullptr, nullptr, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_HANDOVER) launch_ho_reset(h, 2 * grid, nullptr, nullptr, obs_dev, ag_dev, dg_dev, st);
        else if (h->cfg.env_kind == XARM_ENV_STACK_TOWER) k_st_reset<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, nullptr, nullptr, obs_dev, ag_dev, dg_dev, h->class_key);
        else launch_pnp_reset(h, nullptr, nullptr, obs_dev, ag_dev, dg_dev, st);
    }
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_step(xarm_handle *h, const float *actions_dev, float *obs_dev, float *ag_dev, float *dg_dev, float *reward_dev,
              uint8_t *done_dev, uint8_t *success_dev, float *terminal_obs_dev, void *stream) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (!actions_dev || !obs_dev || !ag_dev || !dg_dev || !reward_dev || !done_dev || !success_dev)
        return fail(h, XARM_E_INVALID, "%s", "xarm_step: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)(h->kp.stride / WG);
    int *cnt = h->done_count;
    const bool timed = h->timing && h->ev_created;
    bool pipelined = false;
    const bool overlap = h->reset_overlap && h->kp.auto_reset;
    if (timed && h->ev_n == xarm_handle::NEV) timing_flush(h);
    if (timed) HIPCHK(h, hipEventRecord(h->ev0[h->ev_n], st));
    const bool reach = h->cfg.env_kind == XARM_ENV_REACH, handover = h->cfg.env_kind == XARM_ENV_HANDOVER;
    const bool stack = h->cfg.env_kind == XARM_ENV_STACK_TOWER;
    const HoStage whole{0, xm::HO_N_TICKS, nullptr, nullptr};   // Handover: an unstaged step
    if (h->kp.auto_reset == XARM_AUTO_RESET_LAZY) {
        k_step_lazy<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev);
        if (timed) { HIPCHK(h, hipEventRecord(h->ev1[h->ev_n], st)); HIPCHK(h, hipEventRecord(h->ev2[h->ev_n], st)); h->ev_n++; }
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    // the call's device-side counters (ended episodes, hand-offs, class histogram): zeroed here, in stream order - the
    // handle keeps no host-side per-step state, so a captured step call replays correctly
    HIPCHK(h, hipMemsetAsync(h->counters, 0, sizeof(int) * ((stack && h->class_key) ? 3 + 2 * xs::NCLS : (h->fast_pipeline ? (h->ho_stages > 1 ? 4 + xarm_handle::MAX_ST : 3) : 1)), st));
    if (stack) {
        if (h->class_key) {
            const unsigned cg = (unsigned)((h->kp.num_envs + 255) / 256);
            k_class_hist<<<dim3(cg), dim3(256), 0, st>>>(h->class_key, h->kp.num_envs, h->class_hist);
            k_class_place<<<dim3(cg), dim3(256), 0, st>>>(h->class_key, h->kp.num_envs, h->class_hist, h->class_hist + xs::NCLS, h->class_order, WG / 2);
        }
        k_st_step<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                       terminal_obs_dev, h->done_list, cnt, h->class_order, h->class_key);
    }
    else if (handover && h->cfg.num_obj == 2) {
        if (h->kp.hcfg.use_stand) k_ho2_step<xh::HandoverStandScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                        terminal_obs_dev, h->done_list, cnt);
        else k_ho2_step<xh::HandoverScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                        terminal_obs_dev, h->done_list, cnt);
    }
    else if (handover && h->cfg.num_obj == 1 && h->kp.num_envs <= (int64_t)h->coop_step_limit) {
        // small batch: every env on the cooperative rows, one launch (list == null: all envs; finished episodes -> done_list)
        launch_ho_coop_step(h, ho_coop_grid(h->kp.num_envs), actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev,
                            h->done_list, cnt, nullptr, nullptr, whole, st);
    }
    else if (handover && h->fast_pipeline) {
        // as for PickAndPlace below: every env on the pad-free fast lane-pair step, the ones with an active finger-pad row
        // handed off, untouched, to the cooperative rows (lists of at most eject_coop_cap envs) or to k_ho_step (longer).
        // STAGED: the fast kernel runs the 15 ticks in ho_stages launches.  An env whose pads come alive in stage c keeps the
        // state it had before that stage and re-runs the ticks from the stage's first one on the cooperative rows - on a side
        // stream, beside the next fast stage (16 384 envs are 512 of the 1 024 SIMDs); only the last stage's hand-off, a third
        // of a step long, is on the critical path: fast 0.73 + hand-off 0.69 ms became 0.77 + 0.27.
        pipelined = true;
        const bool stand = h->kp.hcfg.use_stand != 0;
        const int nst = h->ho_stages;
        const int64_t cap = h->kp.num_envs < (int64_t)h->kp.eject_coop_cap ? h->kp.num_envs : (int64_t)h->kp.eject_coop_cap;
        for (int c = 0; c < nst; c++) {
            const HoStage sg{h->ho_tick[c], h->ho_tick[c + 1], h->ho_qt, h->ho_flag};
            int *elist = h->eject_list + (int64_t)c * h->kp.stride, *ecnt = c == 0 ? h->eject_count : h->counters + 3 + c;
            if (stand) k_ho_step_fast<xh::HandoverStandScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                                 terminal_obs_dev, h->done_list, cnt, elist, ecnt, sg);
            else k_ho_step_fast<xh::HandoverScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                      terminal_obs_dev, h->done_list, cnt, elist, ecnt, sg);
            hipStream_t hs = st;
            if (c + 1 < nst) {
                hs = h->st_side[c];
                HIPCHK(h, hipEventRecord(h->st_fork[c], st));
                HIPCHK(h, hipStreamWaitEvent(hs, h->st_fork[c], 0));
            }
            // one done list for every kernel of the call (atomic appends), one reset launch after the last hand-off
            const HoStage rest{sg.tick0, xm::HO_N_TICKS, h->ho_qt, h->ho_flag};
            launch_ho_coop_step(h, ho_coop_grid(cap), actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev,
                                h->done_list, cnt, elist, ecnt, rest, hs);
            if (h->kp.num_envs > cap) {
                if (stand) k_ho_step<xh::HandoverStandScene><<<dim3(2 * grid), dim3(WG), 0, hs>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                                terminal_obs_dev, h->done_list, cnt, elist, ecnt, rest);
                else k_ho_step<xh::HandoverScene><<<dim3(2 * grid), dim3(WG), 0, hs>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                                     terminal_obs_dev, h->done_list, cnt, elist, ecnt, rest);
            }
            if (c + 1 < nst) HIPCHK(h, hipEventRecord(h->st_join[c], hs));
        }
        for (int c = 0; c + 1 < nst; c++) HIPCHK(h, hipStreamWaitEvent(st, h->st_join[c], 0));
    }
    else if (handover && h->kp.hcfg.use_stand)
        k_ho_step<xh::HandoverStandScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev,
                                                                           success_dev, terminal_obs_dev, h->done_list, cnt, nullptr, nullptr, whole);
    else if (handover)
        k_ho_step<xh::HandoverScene><<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev,
                                                                      success_dev, terminal_obs_dev, h->done_list, cnt, nullptr, nullptr, whole);
    else if (reach && h->kp.num_envs <= (int64_t)h->coop_step_limit)
        k_reach_step_coop<<<dim3((unsigned)((h->kp.num_envs + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(
            h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, h->done_list, cnt);
    else if (reach)
        k_reach_step<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                      terminal_obs_dev, h->done_list, cnt);
    else if (h->kp.num_envs <= (int64_t)h->coop_step_limit)
        k_step_coop<<<dim3((unsigned)((h->kp.num_envs + COOP_ENVS - 1) / COOP_ENVS)), dim3(WG), 0, st>>>(
            h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev, terminal_obs_dev, h->done_list, cnt);
    else if (h->fast_pipeline && h->ho_stages > 1) {
        // STAGED PickAndPlace step (XARM_PNP_STAGES): as the staged Handover step above - fast stages on the caller's stream, each
        // stage's hand-off on a side stream beside the next stage, the last one's on the caller's stream; with the reset overlap the
        // episodes that ended on the fast path are reset on `side` after the last fast stage, those that ended in a hand-off after all of them
        pipelined = true;
        const int nst = h->ho_stages;
        const int64_t cap = h->kp.num_envs < (int64_t)h->kp.eject_coop_cap ? h->kp.num_envs : (int64_t)h->kp.eject_coop_cap;
        const unsigned cgrid = (unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS) < 1024u ? (unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS) : 1024u;
        int *list_b = overlap ? h->done_list_b : h->done_list, *cnt_b = overlap ? h->eject_count + 1 : cnt;
        for (int c = 0; c < nst; c++) {
            const HoStage sg{h->ho_tick[c], h->ho_tick[c + 1], h->ho_qt, h->ho_flag};
            int *elist = h->eject_list + (int64_t)c * h->kp.stride, *ecnt = c == 0 ? h->eject_count : h->counters + 3 + c;
            k_step_fast_stage<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                               terminal_obs_dev, h->done_list, cnt, elist, ecnt, sg);
            hipStream_t hs = st;
            if (c + 1 < nst) {
                hs = h->st_side[c];
                HIPCHK(h, hipEventRecord(h->st_fork[c], st));
                HIPCHK(h, hipStreamWaitEvent(hs, h->st_fork[c], 0));
            } else if (overlap) {
                HIPCHK(h, hipEventRecord(h->ev_fork, st));
                HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
                launch_pnp_reset(h, h->done_list, cnt, obs_dev, ag_dev, dg_dev, h->side);
                HIPCHK(h, hipEventRecord(h->ev_join, h->side));
            }
            const HoStage rest{sg.tick0, xm::PNP_N_SUBSTEPS, h->ho_qt, h->ho_flag};
            k_step_coop_list_stage<<<dim3(cgrid), dim3(WG), 0, hs>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                     terminal_obs_dev, list_b, cnt_b, elist, ecnt, rest);
            if (h->kp.num_envs > cap)
                k_step_from_stage<<<dim3(grid), dim3(WG), 0, hs>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                                   terminal_obs_dev, list_b, cnt_b, elist, ecnt, rest);
            if (c + 1 < nst) HIPCHK(h, hipEventRecord(h->st_join[c], hs));
        }
        for (int c = 0; c + 1 < nst; c++) HIPCHK(h, hipStreamWaitEvent(st, h->st_join[c], 0));
    }
    else if (h->fast_pipeline) {
        // every env on the pad-free fast step; the ones with an active finger-pad row are handed off, untouched, to the
        // cooperative kernel (lists of at most eject_coop_cap envs) or to k_step (longer lists) - both launched, the one
        // out of its range exits at once (the count lives on the device).  Episodes that end in the hand-off go to a
        // list of their own (done_list_b): the reset of the ~98 % that ended in k_step_fast need not wait for it.
        pipelined = true;
        k_step_fast<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                     terminal_obs_dev, h->done_list, cnt, h->eject_list, h->eject_count);
        if (overlap) {
            // a reset is six sequential ticks of latency on a few hundred wavefronts (2.9 ms), the hand-off 0.55 ms on a
            // few hundred others: started now on the side stream, the first reset overlaps the hand-off.  The call still
            // waits for the second one - a few dozen envs, but the ones with a finger contact, whose reset carries the pad
            // rows through the homing ticks (worth 0.16 ms per call at 16 384 envs, nothing at 65 536)
            HIPCHK(h, hipEventRecord(h->ev_fork, st));
            HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
            launch_pnp_reset(h, h->done_list, cnt, obs_dev, ag_dev, dg_dev, h->side);
            HIPCHK(h, hipEventRecord(h->ev_join, h->side));
        }
        const int64_t cap = h->kp.num_envs < (int64_t)h->kp.eject_coop_cap ? h->kp.num_envs : (int64_t)h->kp.eject_coop_cap;
        const unsigned cgrid = (unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS) < 1024u ? (unsigned)((cap + COOP_ENVS - 1) / COOP_ENVS) : 1024u;
        // without the side stream (XARM_RESET_OVERLAP=0): one list, one reset after the hand-off
        int *list_b = overlap ? h->done_list_b : h->done_list, *cnt_b = overlap ? h->eject_count + 1 : cnt;
        k_step_coop_list<<<dim3(cgrid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                           terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
        if (h->kp.num_envs > cap)
            k_step<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                    terminal_obs_dev, list_b, cnt_b, h->eject_list, h->eject_count);
    } else
        k_step<<<dim3(grid), dim3(WG), 0, st>>>(h->kp, actions_dev, obs_dev, ag_dev, dg_dev, reward_dev, done_dev, success_dev,
                                                terminal_obs_dev, h->done_list, cnt, nullptr, nullptr);
    if (timed) HIPCHK(h, hipEventRecord(h->ev1[h->ev_n], st));
    if (h->kp.auto_reset && pipelined && overlap) {     // PickAndPlace only (Handover keeps one list, xarm_create)
        launch_pnp_reset(h, h->done_list_b, h->eject_count + 1, obs_dev, ag_dev, dg_dev, st);
        HIPCHK(h, hipStreamWaitEvent(st, h->ev_join, 0));
    } else if (h->kp.auto_reset) {
        if (reach) launch_reach_reset(h, h->done_list, cnt, obs_dev, ag_dev, dg_dev, st);
        else if (handover) launch_ho_reset(h, 2 * grid, h->done_list, cnt, obs_dev, ag_dev, dg_dev, st);
        else if (stack) k_st_reset<<<dim3(2 * grid), dim3(WG), 0, st>>>(h->kp, h->done_list, cnt, obs_dev, ag_dev, dg_dev, h->class_key);
        else launch_pnp_reset(h, h->done_list, cnt, obs_dev, ag_dev, dg_dev, st);
    }
    if (timed) { HIPCHK(h, hipEventRecord(h->ev2[h->ev_n], st)); h->ev_n++; }
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_compute_reward(xarm_handle *h, const float *ag_dev, const float *g_dev, int64_t n, float *out_dev, void *stream) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (n < 0 || (n > 0 && (!ag_dev || !g_dev || !out_dev))) return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: bad argument");
    if (h->cfg.env_kind == XARM_ENV_STACK_TOWER) {
        if (n == 0) return XARM_OK;
        k_st_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->cfg.reward_type, ag_dev, g_dev, n, out_dev);
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    if (h->cfg.env_kind == XARM_ENV_HANDOVER) {
        if (h->cfg.reward_type == XARM_REWARD_DENSE)
            return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: reward_type 'dense' depends on the grasp flags and gripper positions and cannot be relabelled");
        if (n == 0) return XARM_OK;
        if (h->cfg.num_obj == 2) k_ho2_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(ag_dev, g_dev, n, out_dev);
        else k_ho_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(ag_dev, g_dev, n, out_dev);
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    if (h->cfg.env_kind == XARM_ENV_REACH) {
        if (h->cfg.reward_type == XARM_REACH_REWARD_DENSE_DIFF)
            return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: reward_type 'dense_diff' is stateful (d_old) and cannot be relabelled");
        if (n == 0) return XARM_OK;
        k_reach_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->cfg.reward_type, ag_dev, g_dev, n, out_dev);
        HIPCHK(h, hipGetLastError());
        return XARM_OK;
    }
    if (h->cfg.reward_type == XARM_REWARD_DENSE)
        return fail(h, XARM_E_INVALID, "%s", "xarm_compute_reward: reward_type 'dense' depends on the contact state and cannot be relabelled");
    if (n == 0) return XARM_OK;
    k_compute_reward<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->cfg.reward_type, ag_dev, g_dev, n, out_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_get_state(xarm_handle *h, float *state_dev, void *stream) {
    if (!h || !state_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    const int64_t n = h->kp.num_envs * h->kp.state_dim;
    k_get_state<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->kp, state_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}
int xarm_set_state(xarm_handle *h, const float *state_dev, void *stream) {
    if (!h || !state_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    const int64_t n = h->kp.num_envs * h->kp.state_dim;
    k_set_state<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->kp, state_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_episode_steps(xarm_handle *h, int32_t *steps_dev, void *stream) {
    if (!h || !steps_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    const int field = h->cfg.env_kind == XARM_ENV_REACH ? (int)xr::R_STEPS : (h->cfg.env_kind == XARM_ENV_HANDOVER ? (h->cfg.num_obj == 2 ? (int)xh2::G_STEPS : (int)xh::H_STEPS) :
                      (h->cfg.env_kind == XARM_ENV_STACK_TOWER ? (int)xs::K_STEPS : (int)xk::S_STEPS));
    k_episode_steps<<<dim3((unsigned)((h->kp.num_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(h->kp, field, steps_dev);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_debug_substeps(xarm_handle *h, const float *qtarget_dev, int32_t n, void *stream) {
    if (!h || !qtarget_dev || n < 0) return XARM_E_INVALID;
    DEVGUARD(h);
    if (h->cfg.env_kind != XARM_ENV_PICK_AND_PLACE) return fail(h, XARM_E_INVALID, "%s", "xarm_debug_substeps: PickAndPlace only");
    k_substeps<<<dim3((unsigned)(h->kp.stride / WG)), dim3(WG), 0, (hipStream_t)stream>>>(h->kp, qtarget_dev, n);
    HIPCHK(h, hipGetLastError());
    return XARM_OK;
}

int xarm_timing_enable(xarm_handle *h, int32_t enable) {
    if (!h) return XARM_E_INVALID;
    DEVGUARD(h);
    if (enable && !h->ev_created) {
        for (int i = 0; i < xarm_handle::NEV; i++) {
            HIPCHK(h, hipEventCreate(&h->ev0[i]));
            HIPCHK(h, hipEventCreate(&h->ev1[i]));
            HIPCHK(h, hipEventCreate(&h->ev2[i]));
        }
        h->ev_created = true;
    }
    if (enable) { h->ev_n = 0; h->ev_ms = 0; h->ev_reset_ms = 0; h->ev_launches = 0; }
    h->timing = enable;
    return XARM_OK;
}
int xarm_timing_read(xarm_handle *h, double *ms_total, int64_t *launches) {
    if (!h || !ms_total || !launches) return XARM_E_INVALID;
    DEVGUARD(h);
    timing_flush(h);
    *ms_total = h->ev_ms;
    *launches = h->ev_launches;
    return XARM_OK;
}
int xarm_timing_read_reset(xarm_handle *h, double *reset_ms_total, int64_t *launches) {
    if (!h || !reset_ms_total || !launches) return XARM_E_INVALID;
    DEVGUARD(h);
    timing_flush(h);
    *reset_ms_total = h->ev_reset_ms;
    *launches = h->ev_launches;
    return XARM_OK;
}
int xarm_class_keys(xarm_handle *h, uint8_t *keys_dev, void *stream) {
    if (!h || !keys_dev) return XARM_E_INVALID;
    DEVGUARD(h);
    if (!h->class_key) return fail(h, XARM_E_INVALID, "%s", "xarm_class_keys: StackTower handles with the class order enabled only");
    HIPCHK(h, hipMemcpyAsync(keys_dev, h->class_key, (size_t)h->kp.num_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return XARM_OK;
}
int xarm_kernel_limits(const xarm_handle *h, int32_t *reset_coop_limit, int32_t *step_coop_limit) {
    if (!h || !reset_coop_limit || !step_coop_limit) return XARM_E_INVALID;
    *reset_coop_limit = h->kp.coop_limit;
    *step_coop_limit = h->coop_step_limit;
    return XARM_OK;
}
int xarm_debug_counts(xarm_handle *h, int32_t *finished, int32_t *handed_off, void *stream) {
    if (!h || !finished || !handed_off) return XARM_E_INVALID;
    DEVGUARD(h);
    int c[4 + xarm_handle::MAX_ST] = {0};
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(h, hipMemcpy(c, h->counters, sizeof c, hipMemcpyDeviceToHost));
    *finished = c[0];
    *handed_off = h->fast_pipeline ? c[1] : 0;
    for (int k = 1; k < h->ho_stages; k++) *handed_off += c[3 + k];   // staged Handover step: one list per stage
    return XARM_OK;
}
int xarm_stage_info(const xarm_handle *h, int32_t *stages, int32_t *ticks) {
    if (!h || !stages || !ticks) return XARM_E_INVALID;
    const bool staged = h->fast_pipeline && h->ho_stages > 1 && h->kp.num_envs > (int64_t)h->coop_step_limit;
    *stages = staged ? h->ho_stages : 1;
    for (int c = 0; c <= XARM_HO_MAX_STAGES; c++) ticks[c] = 0;
    if (staged) for (int c = 0; c <= h->ho_stages; c++) ticks[c] = h->ho_tick[c];
    else if (h->cfg.env_kind == XARM_ENV_HANDOVER || h->cfg.env_kind == XARM_ENV_PICK_AND_PLACE) ticks[1] = xm::HO_N_TICKS;   // (15 ticks / 15 substeps)
    return XARM_OK;
}
int xarm_pipeline_info(const xarm_handle *h, int32_t *fast_pipeline, int32_t *reset_overlap, int32_t *eject_coop_cap,
                       int32_t *solver_iterations) {
    if (!h || !fast_pipeline || !reset_overlap || !eject_coop_cap || !solver_iterations) return XARM_E_INVALID;
    const bool small = h->kp.num_envs <= (int64_t)h->coop_step_limit;   // (the limit is 0 for the env kinds without a cooperative step kernel)
    *fast_pipeline = (h->fast_pipeline && !small) ? 1 : 0;
    *reset_overlap = (*fast_pipeline && h->reset_overlap) ? 1 : 0;
    *eject_coop_cap = h->kp.eject_coop_cap;
#if defined(XARM_SWEEP_VARIANT)
    *solver_iterations = XARM_SWEEP_VARIANT;
#else
    *solver_iterations = xm::NUM_ITERATIONS;
#endif
    return XARM_OK;
}

} // extern "C"
