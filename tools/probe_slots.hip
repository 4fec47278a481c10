// Probe: cost of one histogram "slot" (one site of every lane -> one LDS atomic) on gfx950 for the forms the
// scan kernels use or could use, at the occupancy of tq_scan_pb_kernel (64 KiB of LDS per 4-wave workgroup: two
// workgroups = 8 waves per CU) and at 8 workgroups per CU (the 256-bin forms).
//   KIND 0: v_add_co c,vcc,c,c ; v_bfe ; v_lshl_add ; s_and exec,save,vcc ; ds_add ; s_mov exec,save     (tq_scan_wg_kernel, 256 bins per wave)
//   KIND 1: v_add_co ; v_perm ; s_and exec ; ds_add ; s_mov exec            (bank-private counters, one slot at a time)
//   KIND 2: 4 x (v_add_co_e64 -> SGPR pair ; v_perm) then 4 x (s_mov exec,mask ; ds_add) ; s_mov exec,save
//   KIND 3: v_add_co ; v_perm ; v_cndmask inc ; ds_add   (no EXEC masking: uncounted sites add 0)
//   KIND 4: v_perm ; ds_add (all lanes, no mask: the floor of the bank-private form)
//   KIND 5: as 2 with 8 slots per block
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_slots.hip -o tools/probe_slots
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int KIND, int LDS_KB>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters, uint32_t density_thr)
{
    __shared__ uint32_t hist[LDS_KB * 256];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int i = tid; i < LDS_KB * 256; i += 256) hist[i] = 0;
    __syncthreads();
    uint32_t s = tid * 2654435761u + blockIdx.x * 977u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    uint32_t pat[8], C = 0;
    for (int j = 0; j < 8; ++j) pat[j] = (rnd() & 0xFFFFu) | (rnd() << 16);
    for (int b = 0; b < 32; ++b) C |= ((rnd() & 0xFFFFu) < density_thr ? 1u : 0u) << b;
    const uint32_t hist_off = (uint32_t)(uintptr_t)hist;
    const uint32_t lanebase = hist_off + (uint32_t)(w >> 1) * 128u + (uint32_t)(lane & 31) * 4u;   // KIND >= 1 (needs LDS_KB = 64)
    const uint32_t wbase = hist_off + (uint32_t)w * 1024u;                                           // KIND 0
    const uint32_t inc = (w & 1) ? 0x10000u : 1u;
    uint32_t sel0, sel1, sel2, sel3;
    uint64_t save;
    asm volatile("s_mov_b32 %0, 0x03020400\n\ts_mov_b32 %1, 0x03020500\n\ts_mov_b32 %2, 0x03020600\n\t"
                 "s_mov_b32 %3, 0x03020700\n\ts_mov_b64 %4, exec"
                 : "=s"(sel0), "=s"(sel1), "=s"(sel2), "=s"(sel3), "=s"(save));
    for (int it = 0; it < iters; ++it) {
        uint32_t c = C;
        if (KIND == 0) {
            uint32_t a, one = 1u;
#define S0(J, K)                                                                                              \
            asm volatile("v_add_co_u32_e32 %[c], vcc, %[c], %[c]\n\t"                                           \
                         "v_bfe_u32 %[a], %[p], " #K "*8, 8\n\t"                                                 \
                         "v_lshl_add_u32 %[a], %[a], 2, %[hist]\n\t"                                             \
                         "s_and_b64 exec, %[save], vcc\n\t"                                                      \
                         "ds_add_u32 %[a], %[one]\n\t"                                                           \
                         "s_mov_b64 exec, %[save]"                                                                \
                         : [c] "+v"(c), [a] "=&v"(a)                                                              \
                         : [p] "v"(pat[J]), [hist] "s"(wbase), [one] "v"(one), [save] "s"(save)                   \
                         : "vcc", "memory");
#define S04(J) S0(J, 3) S0(J, 2) S0(J, 1) S0(J, 0)
            S04(7) S04(6) S04(5) S04(4) S04(3) S04(2) S04(1) S04(0)
        } else if (KIND == 1) {
            uint32_t a;
#define S1(J, K)                                                                                              \
            asm volatile("v_add_co_u32_e32 %[c], vcc, %[c], %[c]\n\t"                                           \
                         "v_perm_b32 %[a], %[p], %[lb], %[sel]\n\t"                                              \
                         "s_and_b64 exec, %[save], vcc\n\t"                                                      \
                         "ds_add_u32 %[a], %[inc]\n\t"                                                           \
                         "s_mov_b64 exec, %[save]"                                                                \
                         : [c] "+v"(c), [a] "=&v"(a)                                                              \
                         : [p] "v"(pat[J]), [lb] "v"(lanebase), [sel] "s"(sel##K), [inc] "v"(inc), [save] "s"(save) \
                         : "vcc", "memory");
#define S14(J) S1(J, 3) S1(J, 2) S1(J, 1) S1(J, 0)
            S14(7) S14(6) S14(5) S14(4) S14(3) S14(2) S14(1) S14(0)
        } else if (KIND == 2) {
#define S24(J)                                                                                                 \
            {                                                                                                  \
                uint32_t a0, a1, a2, a3;                                                                       \
                uint64_t m0, m1, m2, m3;                                                                       \
                asm volatile("v_add_co_u32_e64 %[c], %[m3], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a3], %[p], %[lb], %[sel3]\n\t"                                        \
                             "v_add_co_u32_e64 %[c], %[m2], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a2], %[p], %[lb], %[sel2]\n\t"                                        \
                             "v_add_co_u32_e64 %[c], %[m1], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a1], %[p], %[lb], %[sel1]\n\t"                                        \
                             "v_add_co_u32_e64 %[c], %[m0], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a0], %[p], %[lb], %[sel0]\n\t"                                        \
                             "s_mov_b64 exec, %[m3]\n\t"                                                         \
                             "ds_add_u32 %[a3], %[inc]\n\t"                                                      \
                             "s_mov_b64 exec, %[m2]\n\t"                                                         \
                             "ds_add_u32 %[a2], %[inc]\n\t"                                                      \
                             "s_mov_b64 exec, %[m1]\n\t"                                                         \
                             "ds_add_u32 %[a1], %[inc]\n\t"                                                      \
                             "s_mov_b64 exec, %[m0]\n\t"                                                         \
                             "ds_add_u32 %[a0], %[inc]\n\t"                                                      \
                             "s_mov_b64 exec, %[save]"                                                            \
                             : [c] "+v"(c), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),          \
                               [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)                        \
                             : [p] "v"(pat[J]), [lb] "v"(lanebase), [sel0] "s"(sel0), [sel1] "s"(sel1),               \
                               [sel2] "s"(sel2), [sel3] "s"(sel3), [inc] "v"(inc), [save] "s"(save)                   \
                             : "memory");                                                                          \
            }
            S24(7) S24(6) S24(5) S24(4) S24(3) S24(2) S24(1) S24(0)
        } else if (KIND == 3) {
            uint32_t a, v;
#define S3(J, K)                                                                                              \
            asm volatile("v_add_co_u32_e32 %[c], vcc, %[c], %[c]\n\t"                                           \
                         "v_perm_b32 %[a], %[p], %[lb], %[sel]\n\t"                                              \
                         "v_cndmask_b32_e32 %[v], 0, %[inc], vcc\n\t"                                            \
                         "ds_add_u32 %[a], %[v]"                                                                  \
                         : [c] "+v"(c), [a] "=&v"(a), [v] "=&v"(v)                                                \
                         : [p] "v"(pat[J]), [lb] "v"(lanebase), [sel] "s"(sel##K), [inc] "v"(inc)                 \
                         : "vcc", "memory");
#define S34(J) S3(J, 3) S3(J, 2) S3(J, 1) S3(J, 0)
            S34(7) S34(6) S34(5) S34(4) S34(3) S34(2) S34(1) S34(0)
        } else if (KIND == 4) {
            uint32_t a;
#define S4(J, K)                                                                                              \
            asm volatile("v_perm_b32 %[a], %[p], %[lb], %[sel]\n\t"                                              \
                         "ds_add_u32 %[a], %[inc]"                                                                \
                         : [a] "=&v"(a)                                                                           \
                         : [p] "v"(pat[J]), [lb] "v"(lanebase), [sel] "s"(sel##K), [inc] "v"(inc)                 \
                         : "memory");
#define S44(J) S4(J, 3) S4(J, 2) S4(J, 1) S4(J, 0)
            S44(7) S44(6) S44(5) S44(4) S44(3) S44(2) S44(1) S44(0)
        } else if (KIND == 5) {
#define S58(J, J2)                                                                                             \
            {                                                                                                  \
                uint32_t a0, a1, a2, a3, a4, a5, a6, a7;                                                       \
                uint64_t m0, m1, m2, m3, m4, m5, m6, m7;                                                       \
                asm volatile("v_add_co_u32_e64 %[c], %[m7], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a7], %[p], %[lb], %[sel3]\n\t"                                        \
                             "v_add_co_u32_e64 %[c], %[m6], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a6], %[p], %[lb], %[sel2]\n\t"                                        \
                             "v_add_co_u32_e64 %[c], %[m5], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a5], %[p], %[lb], %[sel1]\n\t"                                        \
                             "v_add_co_u32_e64 %[c], %[m4], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a4], %[p], %[lb], %[sel0]\n\t"                                        \
                             "v_add_co_u32_e64 %[c], %[m3], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a3], %[p2], %[lb], %[sel3]\n\t"                                       \
                             "v_add_co_u32_e64 %[c], %[m2], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a2], %[p2], %[lb], %[sel2]\n\t"                                       \
                             "v_add_co_u32_e64 %[c], %[m1], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a1], %[p2], %[lb], %[sel1]\n\t"                                       \
                             "v_add_co_u32_e64 %[c], %[m0], %[c], %[c]\n\t"                                      \
                             "v_perm_b32 %[a0], %[p2], %[lb], %[sel0]\n\t"                                       \
                             "s_mov_b64 exec, %[m7]\n\tds_add_u32 %[a7], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[m6]\n\tds_add_u32 %[a6], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[m5]\n\tds_add_u32 %[a5], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[m4]\n\tds_add_u32 %[a4], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[m3]\n\tds_add_u32 %[a3], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[m2]\n\tds_add_u32 %[a2], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[m1]\n\tds_add_u32 %[a1], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[m0]\n\tds_add_u32 %[a0], %[inc]\n\t"                              \
                             "s_mov_b64 exec, %[save]"                                                            \
                             : [c] "+v"(c), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),          \
                               [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7),                        \
                               [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3),                        \
                               [m4] "=&s"(m4), [m5] "=&s"(m5), [m6] "=&s"(m6), [m7] "=&s"(m7)                         \
                             : [p] "v"(pat[J]), [p2] "v"(pat[J2]), [lb] "v"(lanebase), [sel0] "s"(sel0), [sel1] "s"(sel1), \
                               [sel2] "s"(sel2), [sel3] "s"(sel3), [inc] "v"(inc), [save] "s"(save)                   \
                             : "memory");                                                                          \
            }
            S58(7, 6) S58(5, 4) S58(3, 2) S58(1, 0)
        }
        // a little vector work between steps, as in the kernel (keeps the compiler from merging iterations too)
        pat[it & 7] = pat[it & 7] * 5u + c;
    }
    __syncthreads();
    uint32_t acc = 0;
    for (int i = tid; i < LDS_KB * 256; i += 256) acc += hist[i];
    out[blockIdx.x * 256 + tid] = acc + pat[0];
}

template <int KIND, int LDS_KB>
void run(const char *name, int wg_per_cu, uint32_t thr)
{
    uint32_t *out;
    hipMalloc(&out, 256 * 16 * 256 * 4);
    const int iters = 4000;
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, LDS_KB>), dim3(blocks), dim3(256), 0, 0, out, 10, thr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, LDS_KB>), dim3(blocks), dim3(256), 0, 0, out, iters, thr);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-slots per CU: wg_per_cu * 4 waves * iters * 32 slots; cycles at 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9;
    const double per_slot_cu = cyc / ((double)wg_per_cu * 4 * iters * 32);
    printf("%-58s wg/CU %d  density %.2f  %.3f ms  %.2f CU-cycles per wave-slot (%.0f per 32-slot step)\n", name, wg_per_cu,
           thr / 65536.0, ms, per_slot_cu, per_slot_cu * 32);
    hipFree(out);
}

int main()
{
    for (uint32_t thr : {26214u, 11469u}) {          // 40 % (full mode), 17.5 % (subsample mode)
        run<0, 4>("0: 256 bins/wave, add_co+bfe+lshl_add, exec per slot", 8, thr);
        run<0, 4>("0: same", 2, thr);
        run<1, 64>("1: bank-private, add_co+perm, exec per slot", 2, thr);
        run<2, 64>("2: bank-private, 4 x (add_co_e64+perm) then 4 x (exec,atomic)", 2, thr);
        run<5, 64>("5: bank-private, 8 x (add_co_e64+perm) then 8 x (exec,atomic)", 2, thr);
        run<3, 64>("3: bank-private, add_co+perm+cndmask, no exec", 2, thr);
        run<4, 64>("4: bank-private, perm + atomic, all lanes", 2, thr);
    }
    return 0;
}
