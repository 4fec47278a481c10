// Probe: LDS cost per wave64 instruction per CU on gfx950 for the operations of the histogram phase:
// ds_add_u32 (no return) with conflict-free / random / skewed bins, ds_read_u8, ds_write_b32, ds_read_b128.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_lds.hip -o tools/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t rnd(uint32_t &s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

// KIND 0: ds_add, bin = lane (distinct banks)      1: ds_add, uniform random bin of 256
//      2: ds_add, skewed bins (min of three draws)  3: ds_read_u8 random byte of own 32
//      4: ds_write_b32 stride 9 dwords              5: ds_read_b128 contiguous
//      6: ds_add, random bin, only ~half of the lanes active
//      7: ds_write_b128, lane stride 48 B   8: ds_write_b64, lane stride 40 B   9: ds_write_b128, lane stride 32 B
//     10: ds_read_u8 random byte, lane stride 48 B   11: lane stride 40 B   12: lane stride 32 B
template <int KIND>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters)
{
    __shared__ uint32_t hist[4][256];
    __shared__ uint32_t park[4][64 * 12];
    __shared__ uint4 img[256];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 256; i += 64) hist[w][i] = 0;
    for (int i = lane; i < 64 * 12; i += 64) park[w][i] = i;
    img[threadIdx.x] = make_uint4(threadIdx.x, 1, 2, 3);
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t idx[8];
    for (int u = 0; u < 8; ++u) {
        const uint32_t a = rnd(s) & 255u, b = rnd(s) & 255u, c = rnd(s) & 255u;
        idx[u] = KIND == 0 ? (uint32_t)lane : KIND == 2 ? min(a, min(b, c)) : a;
    }
    uint32_t acc = 0;
    uint4 acc4 = make_uint4(0, 0, 0, 0);
    const bool active = KIND != 6 || (rnd(s) & 1u);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND <= 2) {
                __hip_atomic_fetch_add(&hist[w][idx[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (KIND == 6) {
                if (active) __hip_atomic_fetch_add(&hist[w][idx[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (KIND == 3) {
                acc += reinterpret_cast<volatile uint8_t *>(park[w])[lane * 36 + (idx[u] & 31)];
            } else if (KIND == 4) {
                reinterpret_cast<volatile uint32_t *>(park[w])[lane * 9 + u] = acc + u;
            } else if (KIND == 7 || KIND == 9) {
                const uint32_t st = KIND == 7 ? 48 : 32;
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                u32x4 vv = {acc, acc + 1, acc + 2, (uint32_t)u};
                asm volatile("ds_write_b128 %0, %1" :: "v"((uint32_t)(uintptr_t)park[w] + lane * st + (u & 1) * 16), "v"(vv) : "memory");
            } else if (KIND == 8) {
                asm volatile("ds_write_b64 %0, %1" :: "v"((uint32_t)(uintptr_t)park[w] + lane * 40 + (u & 3) * 8), "v"((uint64_t)acc * 3u + u) : "memory");
            } else if (KIND >= 10) {
                const uint32_t st = KIND == 10 ? 48 : KIND == 11 ? 40 : 32;
                acc += reinterpret_cast<volatile uint8_t *>(park[w])[lane * st + (idx[u] & 31)];
            } else {
                uint4 v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)&img[(lane + 64 * (u & 3)) & 255]) : "memory");
                acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
            }
        }
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + hist[w][lane] + acc4.x + acc4.y + acc4.z + acc4.w;
}

template <int KIND>
void run(const char *name)
{
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 2000;
    for (int bpc = 1; bpc <= 4; bpc *= 2) {          // blocks (4 waves) per CU
        const int blocks = 256 * bpc;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<KIND><<<blocks, 256>>>(out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<blocks, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_cu = (double)iters * 8 * 4 * bpc;      // wave-instructions per CU
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("%-34s waves/CU %2d: %6.2f cycles per wave-instruction per CU (%.3f ms)\n", name, 4 * bpc,
               cycles / instr_per_cu, ms);
    }
    hipFree(out);
}

int main()
{
    run<0>("ds_add_u32 distinct banks");
    run<1>("ds_add_u32 random bin of 256");
    run<2>("ds_add_u32 skewed bins");
    run<6>("ds_add_u32 random, half the lanes");
    run<3>("ds_read_u8 random byte (stride 36)");
    run<4>("ds_write_b32 stride 9 dwords");
    run<5>("ds_read_b128 contiguous");
    run<7>("ds_write_b128 stride 48 B");
    run<9>("ds_write_b128 stride 32 B");
    run<8>("ds_write_b64 stride 40 B");
    run<10>("ds_read_u8 random, stride 48 B");
    run<11>("ds_read_u8 random, stride 40 B");
    run<12>("ds_read_u8 random, stride 32 B");
    return 0;
}
