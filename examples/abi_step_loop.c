/* A host program on the C ABI alone (no torch, no Python): what a non-Python binding of include/xarm_hip.h does.
 * Creates a PickAndPlace handle, resets, steps with pseudo-random actions on its own HIP stream, prints the rate and a
 * few sums.  Build (gym_xarm_amd/build.py build_example, or by hand):
 *   hipcc -x c -O2 -I include examples/abi_step_loop.c -L gym_xarm_amd/csrc -lxarm_hip -Wl,-rpath,'$ORIGIN/../gym_xarm_amd/csrc' -o examples/abi_step_loop
 * usage: abi_step_loop [num_envs [steps]] */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "xarm_hip.h"

#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
#define XARM(call) do { int r_ = (call); if (r_ != XARM_OK) { fprintf(stderr, "%s: %d %s\n", #call, r_, xarm_last_error(h)); return 3; } } while (0)

int main(int argc, char **argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 4096;
    const int steps = argc > 2 ? atoi(argv[2]) : 50;
    xarm_handle *h = NULL;
    xarm_config cfg = {0};
    cfg.num_envs = E; cfg.seed = 7; cfg.env_kind = XARM_ENV_PICK_AND_PLACE; cfg.num_obj = 1;
    cfg.reward_type = XARM_REWARD_SPARSE; cfg.goal_shape = XARM_GOAL_AIR;
    cfg.init_grasp_rate = 0.5f; cfg.goal_ground_rate = 0.5f; cfg.auto_reset = 1; cfg.device = 0;
    XARM(xarm_create(&cfg, &h));
    xarm_dims_t d;
    XARM(xarm_dims(h, &d));
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    float *obs, *ag, *dg, *rew, *term, *act;
    uint8_t *done, *succ;
    HIP(hipMalloc((void **)&obs, sizeof(float) * E * d.obs_dim)); HIP(hipMalloc((void **)&term, sizeof(float) * E * d.obs_dim));
    HIP(hipMalloc((void **)&ag, sizeof(float) * E * d.goal_dim)); HIP(hipMalloc((void **)&dg, sizeof(float) * E * d.goal_dim));
    HIP(hipMalloc((void **)&rew, sizeof(float) * E)); HIP(hipMalloc((void **)&done, E)); HIP(hipMalloc((void **)&succ, E));
    HIP(hipMalloc((void **)&act, sizeof(float) * E * d.act_dim * 8));
    /* eight action batches from a 64-bit LCG, uniform in [-1, 1) */
    float *hact = (float *)malloc(sizeof(float) * E * d.act_dim * 8);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int64_t i = 0; i < E * d.act_dim * 8; i++) { s = s * 6364136223846793005ull + 1442695040888963407ull; hact[i] = (float)((s >> 40) * (2.0 / 16777216.0) - 1.0); }
    HIP(hipMemcpy(act, hact, sizeof(float) * E * d.act_dim * 8, hipMemcpyHostToDevice));
    XARM(xarm_reset(h, NULL, obs, ag, dg, st));
    for (int k = 0; k < 5; k++) XARM(xarm_step(h, act + (k % 8) * E * d.act_dim, obs, ag, dg, rew, done, succ, term, st));
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
    HIP(hipEventRecord(e0, st));
    for (int k = 0; k < steps; k++) XARM(xarm_step(h, act + (k % 8) * E * d.act_dim, obs, ag, dg, rew, done, succ, term, st));
    HIP(hipEventRecord(e1, st));
    HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP(hipEventElapsedTime(&ms, e0, e1));
    float *hrew = (float *)malloc(sizeof(float) * E), *hobs = (float *)malloc(sizeof(float) * E * d.obs_dim);
    uint8_t *hdone = (uint8_t *)malloc(E);
    HIP(hipMemcpy(hrew, rew, sizeof(float) * E, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hdone, done, E, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hobs, obs, sizeof(float) * E * d.obs_dim, hipMemcpyDeviceToHost));
    double rsum = 0, osum = 0; int64_t nd = 0, bad = 0;
    for (int64_t i = 0; i < E; i++) { rsum += hrew[i]; nd += hdone[i]; }
    for (int64_t i = 0; i < E * d.obs_dim; i++) { osum += hobs[i]; bad += !(hobs[i] == hobs[i]) || hobs[i] > 1e6f || hobs[i] < -1e6f; }
    printf("%s: %lld envs x %d steps in %.3f ms = %.4g env steps/s; obs_dim %d, last step: %lld done, reward sum %.1f, obs sum %.6g, non-finite %lld\n",
           xarm_version(), (long long)E, steps, ms, (double)E * steps / (ms * 1e-3), d.obs_dim, (long long)nd, rsum, osum, (long long)bad);
    XARM(xarm_destroy(h));
    return bad ? 4 : 0;
}
