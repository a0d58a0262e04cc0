// Diagnostic only (not part of the product): the pair step of the cooperative sweep (xarm_coop_core.h sweep_all) with its
// lane-i commit of the new impulse pair written three ways - two v_cndmask_b32 with an SGPR mask (what hipcc emits for
// lv2_commit), one v_mov_b64 under a narrowed EXEC (s_mov_b64 exec, mask / v_mov_b64 / s_mov_b64 exec, -1), and no commit at
// all (lower bound).  One wavefront per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o commit_probe commit_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __device__ __forceinline__ void chain(f2 &g, f2 &lam, f2 b, f2 c, float lo, float hi, unsigned long long m) {
#pragma unroll
    for (int k = 0; k < 64; k++) {
        f2 nl = __builtin_elementwise_fma(g, b, lam);
        nl.x = __builtin_fmaxf(nl.x, 0.f);
        nl.y = __builtin_amdgcn_fmed3f(nl.y, lo, hi);
        const f2 dl = nl - lam;
        if (MODE == 0) {
            asm volatile("v_cndmask_b32_e64 %0, %0, %2, %4\n v_cndmask_b32_e64 %1, %1, %3, %4" : "+v"(lam.x), "+v"(lam.y) : "v"(nl.x), "v"(nl.y), "s"(m));
        } else if (MODE == 1) {
            asm volatile("s_mov_b64 exec, %2\n v_mov_b64 %0, %1\n s_mov_b64 exec, -1" : "+v"(lam) : "v"(nl), "s"(m));
        }
        long long in = __builtin_bit_cast(long long, dl);
        const long long o = __builtin_amdgcn_update_dpp(0ll, in, 0x150 + 3, 0xf, 0xf, true);
        g = __builtin_elementwise_fma(c, __builtin_bit_cast(f2, o), g);
    }
}
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *clk, int iters) {
    f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    float lo = -1.f, hi = 1.f;
    unsigned long long m = 0x0008000800080008ull;      // lane 3 of every 16-lane row
    asm volatile("" : "+s"(m));
    unsigned long long t[4];
    f2 g = {threadIdx.x * 1e-3f, 0.5f}, lam = c;
    t[0] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) chain<0>(g, lam, b, c, lo, hi, m);
    t[1] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) chain<1>(g, lam, b, c, lo, hi, m);
    t[2] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) chain<2>(g, lam, b, c, lo, hi, m);
    t[3] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = g.x + g.y + lam.x + lam.y;
    if (threadIdx.x == 0) for (int k = 0; k < 3; k++) clk[blockIdx.x * 4 + k] = t[k + 1] - t[k];
}
int main() {
    const int blocks = 1024, iters = 2000;
    float *out; unsigned long long *clk;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 4 * 8);
    for (int rep = 0; rep < 3; rep++) probe<<<blocks, 64>>>(out, clk, iters);
    hipDeviceSynchronize();
    static unsigned long long h[1024 * 4];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const char *names[3] = {"pair step, commit = 2 x v_cndmask_b32", "pair step, commit = v_mov_b64 under EXEC", "pair step, no commit"};
    for (int k = 0; k < 3; k++) {
        double c = 0;
        for (int b = 0; b < blocks; b++) c += h[b * 4 + k];
        printf("%-44s %.2f s_memtime ticks per pair step\n", names[k], c / blocks / (iters * 64.0));
    }
    return 0;
}
