// Probe: cost of a vector load instruction on the texture-address path of one CU (gfx950), by width:
// every wave issues back-to-back coalesced loads (lane * width bytes apart) from a small L2/L1-resident buffer.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_ta.hip -o tools/probe_ta
#include <hip/hip_runtime.h>
#include <cstdio>

template <int W>   // dwords per lane: 1, 2, 3, 4
__global__ void k(const unsigned *__restrict__ src, unsigned *out, int iters, int span_dwords)
{
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    unsigned off = (unsigned)((wave * 64 * W) % span_dwords);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned *p = src + ((off + (unsigned)(u * 64 * W)) % (unsigned)span_dwords) + lane * W;
            if constexpr (W == 1) { acc ^= *p; }
            if constexpr (W == 2) { const uint2 v = *reinterpret_cast<const uint2 *>(p); acc ^= v.x ^ v.y; }
            if constexpr (W == 3) { const uint3 v = *reinterpret_cast<const uint3 *>(p); acc ^= v.x ^ v.y ^ v.z; }
            if constexpr (W == 4) { const uint4 v = *reinterpret_cast<const uint4 *>(p); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
        off += 8 * 64 * W;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int W>
void run(const unsigned *src, unsigned *out, int span)
{
    const int iters = 2000, blocks = 256 * 4;           // 4 workgroups of 4 waves per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<W><<<blocks, 256>>>(src, out, 10, span);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<W><<<blocks, 256>>>(src, out, iters, span);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)iters * 8 * 16;       // 16 waves per CU
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("global_load x%d (%2d B/lane, span %d KiB): %.1f cycles per wave-instruction per CU, %.1f B/clk/CU\n", W, 4 * W,
           span / 256, cyc / instr_per_cu, 64.0 * 4 * W * instr_per_cu / cyc);
}

int main()
{
    unsigned *src, *out;
    hipMalloc(&src, 64 << 20);
    hipMemset(src, 1, 64 << 20);
    hipMalloc(&out, 256 * 4 * 256 * 4);
    for (int span : {4096, 1 << 20}) {                        // 16 KiB (L1-resident), 4 MiB (L2-resident), in dwords
        run<1>(src, out, span); run<2>(src, out, span); run<3>(src, out, span); run<4>(src, out, span);
    }
    return 0;
}
