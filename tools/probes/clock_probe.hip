// Diagnostic only (not part of the product): shader clock under a VALU-dense single-wave-per-SIMD load,
// and the issue rate of dependent / independent v_fma_f32 chains from one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *clk, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) a = __builtin_fmaf(a, b, c);   // dependent chain
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) { a = __builtin_fmaf(a, b, c); c = __builtin_fmaf(c, b, d); d = __builtin_fmaf(d, b, e); e = __builtin_fmaf(e, b, a * 0.f + 0.1f); }
    }
    unsigned long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + c + d + e;
    if (threadIdx.x == 0) { clk[blockIdx.x * 4 + 0] = t1 - t0; clk[blockIdx.x * 4 + 1] = r1 - r0; clk[blockIdx.x * 4 + 2] = t2 - t1; clk[blockIdx.x * 4 + 3] = r2 - r1; }
}
int main() {
    const int blocks = 1024, iters = 20000;
    float *out; unsigned long long *clk;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 4 * 8);
    for (int rep = 0; rep < 3; rep++) probe<<<blocks, 64>>>(out, clk, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024 * 4];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double c0 = 0, r0 = 0, c1 = 0, r1 = 0;
    for (int b = 0; b < blocks; b++) { c0 += h[b * 4]; r0 += h[b * 4 + 1]; c1 += h[b * 4 + 2]; r1 += h[b * 4 + 3]; }
    printf("dependent chain:   %.2f cycles per v_fma, clock %.3f GHz\n", c0 / blocks / (iters * 64.0), c0 / r0 * 0.1);
    printf("4 independent chains: %.2f cycles per v_fma, clock %.3f GHz\n", c1 / blocks / (iters * 80.0), c1 / r1 * 0.1);
    return 0;
}
