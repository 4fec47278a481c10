// Probe: issue cost of the integer VALU instructions the scan kernel is made of (gfx950), 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_intops.hip -o tools/probe_intops
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) \
    X(0, "v_and_b32", "v_and_b32 %0, %0, %1") \
    X(1, "v_or_b32", "v_or_b32 %0, %0, %1") \
    X(2, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
    X(3, "v_add_u32", "v_add_u32 %0, %0, %1") \
    X(4, "v_lshlrev_b32 (imm)", "v_lshlrev_b32 %0, 3, %0") \
    X(5, "v_lshrrev_b32 (imm)", "v_lshrrev_b32 %0, 3, %0") \
    X(6, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %1") \
    X(7, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 2, %1") \
    X(8, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 2, %1") \
    X(9, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 8") \
    X(10, "v_or3_b32", "v_or3_b32 %0, %0, %1, %1") \
    X(11, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1") \
    X(12, "v_xad_u32", "v_xad_u32 %0, %0, %1, %1") \
    X(13, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1") \
    X(14, "v_cndmask_b32 (vcc)", "v_cndmask_b32 %0, %0, %1, vcc") \
    X(15, "v_ffbl_b32", "v_ffbl_b32 %0, %0") \
    X(16, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %0, %1") \
    X(17, "v_mov_b32", "v_mov_b32 %0, %1") \
    X(18, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %1") \
    X(19, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 8") \
    X(20, "v_cmp_lt_u32+nothing", "v_cmp_lt_u32 vcc, %0, %1") \
    X(21, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
    X(22, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
    X(23, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %1") \
    X(24, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
    X(25, "v_fma_f64", "v_fma_f64 %0, %0, %1, %1") \
    X(26, "v_rsq_f64", "v_rsq_f64 %0, %0") \
    X(27, "v_rcp_f64", "v_rcp_f64 %0, %0") \
    X(28, "v_mul_f64", "v_mul_f64 %0, %0, %1") \
    X(29, "v_add_f64", "v_add_f64 %0, %0, %1") \
    X(30, "v_cmp_gt_f64", "v_cmp_gt_f64 vcc, %0, %1") \
    X(31, "v_mov_b32 dpp quad_perm", "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(32, "v_max_f64", "v_max_f64 %0, %0, %1") \
    X(33, "v_cvt_f64_u32", "v_cvt_f64_u32 %0, %2") \
    X(34, "v_cndmask_b32 (sgpr pair)", "v_cndmask_b32_e64 %0, %0, %1, %2") \
    X(35, "v_cmp_lt_u32 + v_cndmask (vcc)", "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") \
    X(36, "v_add_co_u32 (vcc out)", "v_add_co_u32 %0, vcc, %0, %1") \
    X(37, "ds_read_u8 + wait", "ds_read_u8 %0, %0\n s_waitcnt lgkmcnt(0)")

template <int KIND>
__global__ void k(unsigned *out, int iters)
{
    unsigned a[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; d[i] = 1.0 + threadIdx.x * 1e-3 + i; }
    unsigned b = threadIdx.x | 1;
    double bd = 1.0000001;
    const unsigned long long smask = 0x5555aaaa5555aaaaull ^ (unsigned long long)iters;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(ID, NAME, ASM) \
    if (KIND == ID) { \
        if (ID >= 25 && ID <= 32 && ID != 31) asm volatile(ASM : "+v"(d[i]) : "v"(bd) : "vcc"); \
        else if (ID == 33) asm volatile(ASM : "+v"(d[i]) : "v"(bd), "v"(a[i]) : "vcc"); \
        else if (ID == 34) asm volatile(ASM : "+v"(a[i]) : "v"(b), "s"(smask) : "vcc"); \
        else if (ID == 37) { a[i] &= 1023u; asm volatile(ASM : "+v"(a[i]) : "v"(b) : "vcc", "memory"); } \
        else asm volatile(ASM : "+v"(a[i]) : "v"(b) : "vcc"); \
    }
                OPS(X)
#undef X
            }
        }
    }
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (unsigned)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, unsigned *out)
{
    const int iters = 400, wps = 8, blocks = 256 * wps;
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
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz nominal)\n", name, cycles / ((double)iters * 64 * wps));
}

int main()
{
    unsigned *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
#define X(ID, NAME, ASM) run<ID>(NAME, out);
    OPS(X)
#undef X
    return 0;
}
