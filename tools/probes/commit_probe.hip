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
// the same recurrence as it is written in the product (clamp c itself), old form and SPLIT form: c' = fma(A, bcast(nl), fma(-A, bcast(lam_old), c)) -
// the subtraction leaves the dependent chain (c -> clamp -> broadcast -> fma), bcast(lam_old) is last sweep's bcast(nl)
template <int MODE> __device__ __forceinline__ void chain2(f2 &g, f2 (&keep)[8], f2 c, float lo, float hi) {
#pragma unroll
    for (int k = 0; k < 64; k++) {
        f2 nl;
        nl.x = __builtin_fmaxf(g.x, 0.f);
        nl.y = __builtin_amdgcn_fmed3f(g.y, lo, hi);
        if (MODE == 0) {
            const f2 dl = nl - keep[k & 7];
            keep[k & 7] = nl;
            const long long o = __builtin_amdgcn_update_dpp(0ll, __builtin_bit_cast(long long, dl), 0x150 + 3, 0xf, 0xf, true);
            g = __builtin_elementwise_fma(c, __builtin_bit_cast(f2, o), g);
        } else {
            const long long o = __builtin_amdgcn_update_dpp(0ll, __builtin_bit_cast(long long, nl), 0x150 + 3, 0xf, 0xf, true);
            const f2 bn = __builtin_bit_cast(f2, o);
            const f2 t = __builtin_elementwise_fma(-c, keep[k & 7], g);
            keep[k & 7] = bn;
            g = __builtin_elementwise_fma(c, bn, t);
        }
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
    f2 keep[8];
    for (int k = 0; k < 8; k++) keep[k] = lam;
    const f2 cs = {1e-3f, -2e-3f};
    unsigned long long u[3];
    u[0] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) chain2<0>(g, keep, cs, lo, hi);
    u[1] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) chain2<1>(g, keep, cs, lo, hi);
    u[2] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = g.x + g.y + lam.x + lam.y + keep[0].x + keep[3].y;
    if (threadIdx.x == 0) { for (int k = 0; k < 3; k++) clk[blockIdx.x * 8 + k] = t[k + 1] - t[k]; for (int k = 0; k < 2; k++) clk[blockIdx.x * 8 + 3 + k] = u[k + 1] - u[k]; }
}
int main() {
    const int blocks = 1024, iters = 2000;
    float *out; unsigned long long *clk;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 8 * 8);
    for (int rep = 0; rep < 3; rep++) probe<<<blocks, 64>>>(out, clk, iters);
    hipDeviceSynchronize();
    static unsigned long long h[1024 * 8];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const char *names[5] = {"pair step, commit = 2 x v_cndmask_b32", "pair step, commit = v_mov_b64 under EXEC", "pair step, no commit",
                            "product form: clamp - sub - bcast - fma", "split form: clamp - bcast - fma (+ fma beside it)"};
    for (int k = 0; k < 5; k++) {
        double c = 0;
        for (int b = 0; b < blocks; b++) c += h[b * 8 + k];
        printf("%-44s %.2f s_memtime ticks per pair step\n", names[k], c / blocks / (iters * 64.0));
    }
    return 0;
}
