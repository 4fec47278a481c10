// Probe: VALU issue cost per wave64 instruction on gfx950 for int32 / f32 / f64 ops at 1, 2, 4, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_valu.hip -o tools/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void k(unsigned *out, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = (a0 << 2) + a1; a1 = (a1 << 2) + a2; a2 = (a2 << 2) + a3; a3 = (a3 << 2) + a4;
                a4 = (a4 << 2) + a5; a5 = (a5 << 2) + a6; a6 = (a6 << 2) + a7; a7 = (a7 << 2) + a0;
            }
        } else if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f0 = fmaf(f0, 1.0001f, f1); f1 = fmaf(f1, 1.0001f, f2); f2 = fmaf(f2, 1.0001f, f3); f3 = fmaf(f3, 1.0001f, f4);
                f4 = fmaf(f4, 1.0001f, f5); f5 = fmaf(f5, 1.0001f, f6); f6 = fmaf(f6, 1.0001f, f7); f7 = fmaf(f7, 1.0001f, f0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d0 = fma(d0, 1.0001, d1); d1 = fma(d1, 1.0001, d2); d2 = fma(d2, 1.0001, d3); d3 = fma(d3, 1.0001, d4);
                d4 = fma(d4, 1.0001, d5); d5 = fma(d5, 1.0001, d6); d6 = fma(d6, 1.0001, d7); d7 = fma(d7, 1.0001, d0);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) +
                                               (unsigned)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}
template <int KIND>
void run(const char *name)
{
    unsigned *out;
    hipMalloc(&out, 256 * 32 * 64 * 4 * 2);
    const int iters = 2000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        // blocks of 256 threads (1 wave per SIMD each); wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<blocks, 256>>>(out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * 64;
        // per SIMD: wps waves, each instr_per_wave instructions; assume 2.4 GHz
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("%s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (%.3f ms)\n", name, wps,
               cycles / (instr_per_wave * wps), ms);
    }
}
int main()
{
    run<0>("int32 v_lshl_add");
    run<1>("f32 v_fma     ");
    run<2>("f64 v_fma     ");
    return 0;
}
