// scan_pb.hpp -- site scan with BANK-PRIVATE histograms: an A/B form of the scan (tq_set_option "scan_method" 6), built for
// full mode (subsample_snps=False: resolve_quartets.py:76-104 -- the reference's default, cli_init.py:61).
// MEASURED SLOWER THAN tq_scan_wg_kernel<false,0,4> AND OFF BY DEFAULT (c3 full mode 11.1 ms against 9.1; DESIGN.md 4.1,
// profiles/r04_scan/): it removes the bank conflicts it was built to remove (SQ_LDS_BANK_CONFLICT -97 %), but a conflict-free
// ds_add_u32 still costs a CU the 4.2 cycles its two source registers take to reach the LDS (tools/probe_slots.hip), only 23 %
// under the shared-bin form, and the 64 KiB of counters leave two workgroups per CU.  Parity-tested like every other form
// (tests/test_gpu_configs.py, the fuzz).
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace, after scan.hpp).
//
// Why it was built.  Full mode counts 12.8 of a lane's 32 sites per 2048-site step (c3).  tq_scan_wg_kernel<false,0,4> issues one
// EXEC-masked ds_add_u32 per site slot into a 256-bin histogram per wave: 32 atomics per step at 40 % lane occupancy,
// 13 random bins per 32-lane group into 32 banks = the birthday bound, 2.3 LDS cycles per group where 1 is the floor
// (profiles/r03_scan/full_mode_method0.txt: SQ_LDS_IDX_ACTIVE 77 % of the CU-busy cycles, half of it conflicts; the
// kernel is bound by them, 9.1 ms per 1e6 c3 quartets against 5.8 in subsample mode).  Denser instructions do not
// help (the walk: 18 trips at 71 %, each with a byte read), re-hashed bins do not help (the conflicts are those of
// uniformly random bins), K interleaved copies do not help (same load factor).  What removes them is giving every
// lane of a 32-lane group ITS OWN BANK: bin b of lane l lives at dword  b * 64 + pair * 32 + (l mod 32)  -- no two
// lanes of a group can meet, an atomic costs its 2 LDS cycles whatever the bins are.  Lanes l and l + 32 share a
// counter (they are served in different LDS cycles).
//
// Capacity.  256 bins x 32 copies x 4 B = 32 KiB per quartet would leave one wave per SIMD.  The counters are
// therefore 16 bits wide and TWO quartets (an even and an odd wave of the workgroup = a "pair") share a dword: the
// even wave adds 1, the odd wave adds 65536 -- the increment is a per-wave constant, no per-site arithmetic.  A
// counter takes at most 64 per step (32 slots x the two lanes that share it), so a half cannot overflow into its
// neighbour for up to 1023 steps = 2.09 M sites; the host selects this kernel only then (launch_scan_n).
// LDS: 64 KiB of counters + 6.5 KiB of shared-row image per 4-wave workgroup, two workgroups per CU (gfx950: 160 KiB).
//
// Address arithmetic.  The counters start at LDS offset 0 (64 KiB-aligned), a bin's row of 64 dwords is 256 bytes:
// byte 1 of the address IS the pattern byte, byte 0 = pair * 128 + (lane mod 32) * 4, bytes 2-3 = 0.  One
// v_perm_b32 picks pattern byte k of a pattern dword into byte 1 of the lane's constant base: extract + scale + add
// in ONE vector instruction (the 256-bin form needs v_bfe + v_lshl_add).  A slot is 2 vector instructions
// (v_add_co c,vcc,c,c: shifts the count mask and delivers the slot's bit as the lane mask; v_perm) + the atomic.
//
// Epilogue per block: the two waves of a pair fold the pair's 32 copies (bins 0-127 / 128-255, both halves of every
// dword, i.e. for both quartets), clear them and store 256 counts per quartet.  c3: 25 steps per quartet.
#pragma once

constexpr int PB_NW = 4;                 // waves (= quartets) per workgroup: two pairs
constexpr int PB_MAX_TILES = 1023;       // 16-bit counters: 64 increments per step at most

struct PbLds {
    uint32_t hist[256 * 64];             // [bin][pair][copy]: low half = even wave of the pair, high half = odd wave
    uint4 image[2][SHARED_SLOTS];        // shared rows a, b of one step (scan.hpp), double-buffered
};

template <bool SUB>
__global__ void __launch_bounds__(PB_NW *WAVE)
tq_scan_pb_kernel(DevData d, const uint32_t *__restrict__ quartets, const uint32_t *__restrict__ order, int64_t Q,
                  uint32_t *__restrict__ cm, int64_t xcd_chunk, uint32_t *__restrict__ diag)
{
    constexpr int NW = PB_NW;
    __shared__ PbLds S;
    const int tid = threadIdx.x;
    const uint32_t hist_off = lds_offset(S.hist);
    // the address trick needs the counters on a 64 KiB boundary (they are the kernel's only LDS object besides the image,
    // which follows them: offset 0; the compiler folds this test away).  Q < 0 = the host's one-time probe of exactly that.
    if (Q < 0 || (hist_off & 0xFFFFu)) {
        if (tid == 0 && blockIdx.x == 0 && diag) diag[0] = 0x80000000u | (hist_off & 0xFFFFu);
        return;
    }
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    {
        uint4 *hz = reinterpret_cast<uint4 *>(S.hist);
#pragma unroll
        for (int k = 0; k < 16; ++k) hz[tid + 256 * k] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t T = (uint32_t)d.T;
    const int last = d.ntiles - 1;
    const int64_t nblk = (Q + NW - 1) / NW;
    const uint8_t *rows = d.rows;
    const uint8_t *nib = d.nib;
    const uint8_t *planes = reinterpret_cast<const uint8_t *>(d.planes);
    const uint8_t *planes3 = reinterpret_cast<const uint8_t *>(d.planes3);
    const uint32_t pitch = (uint32_t)d.pitch, npitch = pitch / 2, wpitch = (uint32_t)d.W * 16u,
                   w3pitch = (uint32_t)d.W * 12u;
    // this lane's counter column and this wave's increment
    const uint32_t lanebase = hist_off + (uint32_t)(w >> 1) * 128u + (uint32_t)(lane & 31) * 4u;
    const uint32_t inc = (w & 1) ? 0x10000u : 1u;
    __syncthreads();

    const int64_t nslot = xcd_chunk > 0 ? 8 * xcd_chunk : nblk;
    for (int64_t blk0 = blockIdx.x; blk0 < nslot; blk0 += gridDim.x) {
        int64_t blk = blk0;
        if (xcd_chunk > 0) {                                    // XCD-contiguous block ids (scan.hpp)
            const int64_t x = blk0 & 7, j = blk0 >> 3;
            blk = x * xcd_chunk + j;
            if (j >= xcd_chunk || blk >= nblk) continue;       // uniform for the whole workgroup
        }
        const int64_t it0 = blk * NW;
        const int64_t lqi = order ? (int64_t)order[it0] : it0;
        const uint4 lq = reinterpret_cast<const uint4 *>(quartets)[lqi];
        uint32_t la = __builtin_amdgcn_readfirstlane(lq.x), lb = __builtin_amdgcn_readfirstlane(lq.y);
        const bool leader_ok = (la < T) & (lb < T);
        if (!leader_ok) la = lb = 0;
        const int64_t it = it0 + w;
        const bool have = it < Q;
        const int64_t qi = have ? (order ? (int64_t)order[it] : it) : 0;
        const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[qi];
        uint32_t q[4];
        q[0] = __builtin_amdgcn_readfirstlane(qv.x);
        q[1] = __builtin_amdgcn_readfirstlane(qv.y);
        q[2] = __builtin_amdgcn_readfirstlane(qv.z);
        q[3] = __builtin_amdgcn_readfirstlane(qv.w);
        const bool bad = (q[0] >= T) | (q[1] >= T) | (q[2] >= T) | (q[3] >= T);
        const bool work = have && !bad;                     // wave-uniform
        const bool shares = work && leader_ok && q[0] == la && q[1] == lb;
        const uint32_t qc = work ? q[2] : 0, qd = work ? q[3] : 0;
        OwnOff oo;
        oo.c = qc * npitch + (uint32_t)lane * 16u;
        oo.d = qd * npitch + (uint32_t)lane * 16u;
        oo.pc = qc * w3pitch + (uint32_t)lane * 12u;
        oo.pd = qd * w3pitch + (uint32_t)lane * 12u;
        // Cooperative jobs (scan.hpp): wave 0 = nibble codes of a and b, wave 1 = their plane records, waves 2-3 none.
        // The step loop exists in four copies chosen by wave-uniform facts that do not change inside it -- JOB (0, 1,
        // none) for waves that work and share the leader's rows (FAST), and one generic copy for the rest (a wave at a
        // group boundary of the sorted order, a wave without a quartet).  Inside a FAST copy every load is
        // unconditional, so the compiler's s_waitcnt counts are exact: with the job and the boundary case as branches
        // INSIDE the loop it has to assume the fewest loads in flight at every join and turns "wait for the rows
        // requested a step ago" into vmcnt(0) -- the job waves then wait for the image loads they issued a few
        // instructions earlier, a full L2 round trip per step (ISA of rounds 2-3).
        auto loop = [&](auto job_tag, auto fast_tag) {
            constexpr int JOB = decltype(job_tag)::value;
            constexpr bool FAST = decltype(fast_tag)::value;
            auto fetch_x = [=](int tile) -> uint4 {
                if (JOB == 0) return ld16(nib, la * npitch + (uint32_t)lane * 16u + (uint32_t)tile * (TILE / 2));
                if (JOB == 1) return ld16(planes, la * wpitch + (uint32_t)lane * 16u + (uint32_t)tile * (WAVE * 16));
                return make_uint4(0, 0, 0, 0);
            };
            auto fetch_y = [=](int tile) -> uint4 {
                if (JOB == 0) return ld16(nib, lb * npitch + (uint32_t)lane * 16u + (uint32_t)tile * (TILE / 2));
                if (JOB == 1) return ld16(planes, lb * wpitch + (uint32_t)lane * 16u + (uint32_t)tile * (WAVE * 16));
                return make_uint4(0, 0, 0, 0);
            };
            auto publish = [=](uint4 *buf, uint4 x, uint4 y) {
                if (JOB == 0) {
                    const uint32_t h = 0xF0F0F0F0u;
                    const uint32_t s0 = (x.x << 2) + y.x, s1 = (x.y << 2) + y.y, s2 = (x.z << 2) + y.z, s3 = (x.w << 2) + y.w;
                    buf[lane] = make_uint4((s0 << 4) & h, s0 & h, (s1 << 4) & h, s1 & h);           // sites 0-15 of the lane
                    buf[64 + lane] = make_uint4((s2 << 4) & h, s2 & h, (s3 << 4) & h, s3 & h);      // sites 16-31
                } else if (JOB == 1) {                           // x, y = {miss, p0, p1, run-begin} of a, of b
                    buf[128 + lane] = make_uint4(x.y, x.z, x.x | y.x, (x.y ^ y.y) | (x.z ^ y.z));
                    if (SUB) reinterpret_cast<uint32_t *>(buf + 192)[lane] = x.w;
                }
            };
            // the shared image of one step as this lane reads it
            struct Img {
                uint4 ab0, ab1, r1;
                uint32_t Bw;
            };
            auto read_image = [=](Img &I, const uint4 *buf) {
                if (!(FAST || shares)) return;
                I.ab0 = buf[lane];
                I.ab1 = buf[64 + lane];
                I.r1 = buf[128 + lane];
                if (SUB) I.Bw = reinterpret_cast<const uint32_t *>(buf + 192)[lane];
            };
            uint32_t tile_carry = 0;
            // One step.  Two register sets (own rows R, image I) alternate between even and odd steps: what step t has
            // consumed is refilled for step t + 2 -- the own rows as soon as the patterns are built, the image after the
            // barrier that ends the step -- so every load has a whole step of the other parity to land.
            auto step = [&](OwnRegs &R, Img &I, uint4 *buf, int t) {
                const int t2 = min(t + 2, last);
                uint4 sx = fetch_x(t2), sy = fetch_y(t2);
                __builtin_amdgcn_sched_barrier(0);
                if (FAST || work) {
                    if (!(FAST || shares)) {                        // group boundary: private rows a and b
                        const uint32_t o0 = q[0] * pitch + (uint32_t)t * TILE + (uint32_t)lane * 16u;
                        const uint32_t o1 = q[1] * pitch + (uint32_t)t * TILE + (uint32_t)lane * 16u;
                        const uint4 a0 = ld16(rows, o0), a1 = ld16(rows, o0 + 1024u);
                        const uint4 b0 = ld16(rows, o1), b1 = ld16(rows, o1 + 1024u);
                        I.ab0 = make_uint4(((a0.x << 2) + b0.x) << 4, ((a0.y << 2) + b0.y) << 4, ((a0.z << 2) + b0.z) << 4,
                                           ((a0.w << 2) + b0.w) << 4);
                        I.ab1 = make_uint4(((a1.x << 2) + b1.x) << 4, ((a1.y << 2) + b1.y) << 4, ((a1.z << 2) + b1.z) << 4,
                                           ((a1.w << 2) + b1.w) << 4);
                        const uint4 pa = ld16(planes, q[0] * wpitch + (uint32_t)t * (WAVE * 16) + (uint32_t)lane * 16u);
                        const uint4 pb = ld16(planes, q[1] * wpitch + (uint32_t)t * (WAVE * 16) + (uint32_t)lane * 16u);
                        I.r1 = make_uint4(pa.y, pa.z, pa.x | pb.x, (pa.y ^ pb.y) | (pa.z ^ pb.z));
                        I.Bw = pa.w;
                    }
                    uint32_t c = count_mask_shared<SUB>(I.r1, SUB ? I.Bw : 0u, R.pc, R.pd, lane, tile_carry);
                    uint32_t m;
                    m = sgpr_const_0f();
                    const uint32_t s0 = (R.c.x << 2) + R.d.x, s1 = (R.c.y << 2) + R.d.y, s2 = (R.c.z << 2) + R.d.z,
                                   s3 = (R.c.w << 2) + R.d.w;
                    uint32_t pat[8];
                    pat[0] = and_or(s0, m, I.ab0.x); pat[1] = and_or(s0 >> 4, m, I.ab0.y);
                    pat[2] = and_or(s1, m, I.ab0.z); pat[3] = and_or(s1 >> 4, m, I.ab0.w);
                    pat[4] = and_or(s2, m, I.ab1.x); pat[5] = and_or(s2 >> 4, m, I.ab1.y);
                    pat[6] = and_or(s3, m, I.ab1.z); pat[7] = and_or(s3 >> 4, m, I.ab1.w);
                    // the own rows of step t + 2 into the registers this step has just consumed.  (The first barrier keeps
                    // the scheduler from hoisting the loads above the last uses of the old values: it then needs copies of
                    // the loaded words, placed -- with their s_waitcnt -- right behind the loads.)
                    __builtin_amdgcn_sched_barrier(0);
                    load_own(R, nib, planes3, oo, t2);
                    __builtin_amdgcn_sched_barrier(0);
                    // 32 slots, most significant site first, four (one pattern dword) per block: v_add_co shifts the count mask
                    // and delivers the slot's bit as a lane mask (its carry-out, into an SGPR pair); v_perm_b32 builds the
                    // counter address {base.3, base.2, pattern byte, base.0}; then the four atomics, each under its mask.
                    // Vector work and atomics are kept apart so that no atomic waits for a mask that is still on its way
                    // from the vector to the scalar unit (with two waves per SIMD nobody hides that latency).
#ifdef TQ_NO_ASM
                    // plain C++ form of the 32 slots (reference form, scan.hpp): counter of bin b for this lane =
                    // dword b * 64 + pair * 32 + (lane mod 32)
                    {
                        uint32_t *col = S.hist + (uint32_t)(w >> 1) * 32u + (uint32_t)(lane & 31);
#pragma unroll
                        for (int j = 7; j >= 0; --j) {
#pragma unroll
                            for (int k = 3; k >= 0; --k) {
                                if (c & (1u << (4 * j + k)))
                                    __hip_atomic_fetch_add(col + ((pat[j] >> (8 * k)) & 0xFFu) * 64u, inc, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
#else
                    uint32_t sel0, sel1, sel2, sel3;                 // D.byte1 <- S0.byte K (selector 4 + K), the rest <- S1
                    uint64_t save;
                    asm volatile("s_mov_b32 %0, 0x03020400\n\ts_mov_b32 %1, 0x03020500\n\ts_mov_b32 %2, 0x03020600\n\t"
                                 "s_mov_b32 %3, 0x03020700\n\ts_mov_b64 %4, exec"
                                 : "=s"(sel0), "=s"(sel1), "=s"(sel2), "=s"(sel3), "=s"(save));
#define TQ_PB_SLOT4(J)                                                                                          \
                    {                                                                                            \
                        uint32_t a0, a1, a2, a3;                                                                 \
                        uint64_t m0, m1, m2, m3;                                                                 \
                        asm volatile("v_add_co_u32_e64 %[c], %[m3], %[c], %[c]\n\t"                              \
                                     "v_perm_b32 %[a3], %[p], %[lb], %[sel3]\n\t"                                \
                                     "v_add_co_u32_e64 %[c], %[m2], %[c], %[c]\n\t"                              \
                                     "v_perm_b32 %[a2], %[p], %[lb], %[sel2]\n\t"                                \
                                     "v_add_co_u32_e64 %[c], %[m1], %[c], %[c]\n\t"                              \
                                     "v_perm_b32 %[a1], %[p], %[lb], %[sel1]\n\t"                                \
                                     "v_add_co_u32_e64 %[c], %[m0], %[c], %[c]\n\t"                              \
                                     "v_perm_b32 %[a0], %[p], %[lb], %[sel0]\n\t"                                \
                                     "s_mov_b64 exec, %[m3]\n\t"                                                 \
                                     "ds_add_u32 %[a3], %[inc]\n\t"                                              \
                                     "s_mov_b64 exec, %[m2]\n\t"                                                 \
                                     "ds_add_u32 %[a2], %[inc]\n\t"                                              \
                                     "s_mov_b64 exec, %[m1]\n\t"                                                 \
                                     "ds_add_u32 %[a1], %[inc]\n\t"                                              \
                                     "s_mov_b64 exec, %[m0]\n\t"                                                 \
                                     "ds_add_u32 %[a0], %[inc]\n\t"                                              \
                                     "s_mov_b64 exec, %[save]"                                                    \
                                     : [c] "+v"(c), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),  \
                                       [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)                \
                                     : [p] "v"(pat[J]), [lb] "v"(lanebase), [sel0] "s"(sel0), [sel1] "s"(sel1),       \
                                       [sel2] "s"(sel2), [sel3] "s"(sel3), [inc] "v"(inc), [save] "s"(save)           \
                                     : "memory");                                                                  \
                    }
                    TQ_PB_SLOT4(7) TQ_PB_SLOT4(6) TQ_PB_SLOT4(5) TQ_PB_SLOT4(4) TQ_PB_SLOT4(3) TQ_PB_SLOT4(2) TQ_PB_SLOT4(1)
                    TQ_PB_SLOT4(0)
#undef TQ_PB_SLOT4
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                // the image words of step t + 2 are "defined" here for the compiler: the copies that assemble the 16-byte
                // LDS stores (and the s_waitcnt they need) then land here and not right behind the loads at the top
                if (JOB < 2) {
                    pin4(sx);
                    pin4(sy);
                }
                publish(buf, sx, sy);
                __syncthreads();
                read_image(I, buf);                                 // image of step t + 2
            };

            // prologue: images of steps 0 and 1, own rows of steps 0 and 1
            OwnRegs R0, R1;
            Img I0, I1;
            I0.Bw = I1.Bw = 0;
            {
                const int t1 = min(1, last);
                uint4 sx = fetch_x(0), sy = fetch_y(0);
                uint4 ux = fetch_x(t1), uy = fetch_y(t1);
                load_own(R0, nib, planes3, oo, 0);
                load_own(R1, nib, planes3, oo, t1);
                publish(S.image[0], sx, sy);
                publish(S.image[1], ux, uy);
                __syncthreads();
                read_image(I0, S.image[0]);
                read_image(I1, S.image[1]);
                __syncthreads();        // step 0 overwrites buffer 0 at its end: every wave must have read it by then
            }
            for (int t = 0; t < d.ntiles; t += 2) {
                step(R0, I0, S.image[0], t);
                if (t + 1 >= d.ntiles) break;
                step(R1, I1, S.image[1], t + 1);
            }
        };
        using std::integral_constant;
        if (shares) {
            if (w == 0) loop(integral_constant<int, 0>{}, integral_constant<bool, true>{});
            else if (w == 1) loop(integral_constant<int, 1>{}, integral_constant<bool, true>{});
            else loop(integral_constant<int, 2>{}, integral_constant<bool, true>{});
        } else {
            if (w == 0) loop(integral_constant<int, 0>{}, integral_constant<bool, false>{});
            else if (w == 1) loop(integral_constant<int, 1>{}, integral_constant<bool, false>{});
            else loop(integral_constant<int, 2>{}, integral_constant<bool, false>{});
        }
        // fold: wave h of pair p takes bins [128 h, 128 h + 128) of the pair's counters, both halves of every dword (the
        // low halves are the even wave's quartet, the high halves the odd wave's), clears them and stores the counts.
        // (the barrier that ended the last step also made every wave's atomics visible.)
        {
            const int p = w >> 1, h = w & 1;
            const int64_t ite = it0 + 2 * p, ito = ite + 1;
            const bool have_e = ite < Q, have_o = ito < Q;
            const int64_t qe = have_e ? (order ? (int64_t)order[ite] : ite) : 0;
            const int64_t qo = have_o ? (order ? (int64_t)order[ito] : ito) : 0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int bin = h * 128 + r * 64 + lane;
                uint4 *row = reinterpret_cast<uint4 *>(S.hist + bin * 64 + p * 32);
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int pc = (k + lane) & 7;             // rotated by the lane: 2-way instead of 16-way bank conflicts
                    const uint4 v = row[pc];
                    row[pc] = make_uint4(0, 0, 0, 0);
                    lo += (v.x & 0xFFFFu) + (v.y & 0xFFFFu) + (v.z & 0xFFFFu) + (v.w & 0xFFFFu);
                    hi += (v.x >> 16) + (v.y >> 16) + (v.z >> 16) + (v.w >> 16);
                }
                if (have_e) __builtin_nontemporal_store(lo, &cm[qe * 256 + bin]);
                if (have_o) __builtin_nontemporal_store(hi, &cm[qo * 256 + bin]);
            }
        }
        __syncthreads();
    }
}
