// Diagnostic only (not part of the product): cycles per instruction of DEPENDENT chains of the instructions the
// cooperative sweep is made of (v_pk_fma_f32, v_pk_add_f32, v_mov_b64_dpp row_newbcast, v_mov_b32_dpp, v_med3_f32,
// v_cndmask_b32 with an SGPR mask), one wavefront per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o dpp_probe dpp_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *clk, int iters) {
    f2 a = {threadIdx.x * 1e-3f, 0.5f}, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    float s = threadIdx.x * 1e-3f, lo = -1.f, hi = 1.f;
    unsigned long long t[8];
    unsigned long long m = 0x0001000100010001ull;
    t[0] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    t[1] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
    t[2] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("s_nop 1\n v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a));) }
    t[3] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s));) }
    t[4] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s) : "v"(lo), "v"(hi));) }
    t[5] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP64(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(s) : "v"(lo), "s"(m));) }
    t[6] = __builtin_amdgcn_s_memtime();
    // the sweep's pair step in C: pk_fma -> med3 / max -> pk_sub -> b64 dpp broadcast -> pk_fma
    f2 g = a, lam = c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            f2 nl = __builtin_elementwise_fma(g, b, lam);
            nl.x = __builtin_fmaxf(nl.x, 0.f);
            nl.y = __builtin_amdgcn_fmed3f(nl.y, lo, hi);
            const f2 dl = nl - lam;
            long long in = __builtin_bit_cast(long long, dl);
            const long long o = __builtin_amdgcn_update_dpp(0ll, in, 0x150 + 3, 0xf, 0xf, true);
            g = __builtin_elementwise_fma(c, __builtin_bit_cast(f2, o), g);
        }
    }
    t[7] = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a.x + a.y + s + g.x + g.y;
    if (threadIdx.x == 0) for (int k = 0; k < 7; k++) clk[blockIdx.x * 8 + k] = t[k + 1] - t[k];
}
int main() {
    const int blocks = 1024, iters = 2000;
    float *out; unsigned long long *clk;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 8 * 8);
    for (int rep = 0; rep < 3; rep++) probe<<<blocks, 64>>>(out, clk, iters);
    hipDeviceSynchronize();
    static unsigned long long h[1024 * 8];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const char *names[7] = {"v_pk_fma_f32", "v_pk_add_f32", "s_nop 1 + v_mov_b64_dpp row_newbcast", "s_nop 1 + v_mov_b32_dpp row_newbcast", "v_med3_f32", "v_cndmask_b32 (sgpr mask)", "pair step chain (5 instr + s_nop 1)"};
    for (int k = 0; k < 7; k++) {
        double c = 0;
        for (int b = 0; b < blocks; b++) c += h[b * 8 + k];
        printf("%-42s %.2f s_memtime ticks per link (x clock ratio for cycles)\n", names[k], c / blocks / (iters * 64.0));
    }
    return 0;
}
