// Diagnostic only (not part of the product): does a lone wavefront issue VALU instructions faster when only
// 16 / 32 of its 64 lanes are active (EXEC-masked passes skipped?), and what does v_pk_fma_f32 cost?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void probe(float *out, unsigned long long *clk, int iters, int active, int packed) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f;
    float2v pa = {a, a + 1.f}, pb = {b, b}, pc = {c, d}, pd = {d, e}, pe = {e, c};
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < active) {
        t0 = __builtin_amdgcn_s_memtime();
        if (!packed) {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int k = 0; k < 16; k++) { a = __builtin_fmaf(a, b, c); c = __builtin_fmaf(c, b, d); d = __builtin_fmaf(d, b, e); e = __builtin_fmaf(e, b, 0.1f); }
            }
        } else {
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    pa = __builtin_elementwise_fma(pa, pb, pc); pc = __builtin_elementwise_fma(pc, pb, pd);
                    pd = __builtin_elementwise_fma(pd, pb, pe); pe = __builtin_elementwise_fma(pe, pb, pa);
                }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + c + d + e + pa.x + pa.y + pc.x + pc.y + pd.x + pd.y + pe.x + pe.y;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    const int iters = 20000;
    float *out; unsigned long long *clk;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&clk, 1024 * 8);
    static unsigned long long h[1024];
    for (int packed = 0; packed < 2; packed++)
        for (int blocks : {1024, 64})
            for (int active : {64, 32, 16, 1}) {
                for (int rep = 0; rep < 2; rep++) probe<<<blocks, 64>>>(out, clk, iters, active, packed);
                hipDeviceSynchronize();
                hipMemcpy(h, clk, blocks * 8, hipMemcpyDeviceToHost);
                double c = 0;
                for (int b = 0; b < blocks; b++) c += h[b];
                printf("%s blocks %4d active lanes %2d: %.2f cycles per instruction\n", packed ? "v_pk_fma_f32" : "v_fma_f32   ", blocks, active,
                       c / blocks / (iters * 64.0));
            }
    return 0;
}
