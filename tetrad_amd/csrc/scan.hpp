// scan.hpp -- site scan -> 256-bin pattern histogram (per-wave and workgroup-cooperative kernels)
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
#pragma once

// The hand-written instruction sequences below (walk_set_bits, the slot macros, the subsample carry, the park stores)
// are gfx9 / CDNA encodings for 64-lane wavefronts: vcc and exec as 64-bit masks, v_add_co_u32 with an SGPR-pair carry-out,
// LDS instructions without M0 set-up, ds_write_addtid_b32.  They are entered with ALL 64 lanes active (a working wave of
// a full workgroup; count_from_candidates reads carries as lane masks).  -DTQ_NO_ASM builds the plain C++ form of every one
// of them: the reference form for A/B runs (tools/ab) and the starting point of a port -- same results, bit for bit.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(TQ_NO_ASM)
#error "scan.hpp: the inline assembly is written for gfx950 (CDNA4, wave64); build with --offload-arch=gfx950 or -DTQ_NO_ASM"
#endif

// ------------------------------------------------------------------------------------
// kernel 1: site scan -> 256-bin pattern histogram
// ------------------------------------------------------------------------------------
struct TileRegs {
    uint4 a0, a1, b0, b1, c0, c1, d0, d1;   // 32 site bytes of each of the four rows
    uint4 pa, pb, pc, pd;                    // plane records {miss, p0, p1, runbeg} of the four rows
};

__device__ __forceinline__ void load_tile(TileRegs &r, const DevData &d, const uint32_t (&q)[4], int tile,
                                          int lane)
{
    const int64_t boff = (int64_t)tile * TILE + lane * 16;            // panel layout: see row_offset
    const uint4 *pa = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[0] * d.pitch + boff);
    const uint4 *pb = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[1] * d.pitch + boff);
    const uint4 *pc = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[2] * d.pitch + boff);
    const uint4 *pd = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[3] * d.pitch + boff);
    r.a0 = pa[0]; r.a1 = pa[64];
    r.b0 = pb[0]; r.b1 = pb[64];
    r.c0 = pc[0]; r.c1 = pc[64];
    r.d0 = pd[0]; r.d1 = pd[64];
    const int64_t woff = (int64_t)tile * WAVE + lane;
    r.pa = d.planes[(int64_t)q[0] * d.W + woff];
    r.pb = d.planes[(int64_t)q[1] * d.W + woff];
    r.pc = d.planes[(int64_t)q[2] * d.W + woff];
    r.pd = d.planes[(int64_t)q[3] * d.W + woff];
}

// Sites to count among this lane's 32 (bit i = site i).
//   U = variable among the four taxa and none missing (resolve_quartets.py:216-218).
//   full mode      : C = U.
//   subsample mode : C = sites of U that are the first unmasked site of their locus run
//                    (resolve_quartets.py:58-64: a site is counted iff unmasked and its locus
//                    differs from the locus of the previous unmasked site).  seen(i) = "an
//                    unmasked site precedes i in the same run" obeys
//                    t(i) = U(i) | (P(i) & t(i-1)), P = ~runbegin, seen(i) = P(i) & t(i-1),
//                    which is the carry recurrence of the addition (U|P) + U.
// U = candidate sites (variable among the four taxa, none missing), B = run-begin bits
template <bool SUB>
__device__ __forceinline__ uint32_t count_from_candidates(uint32_t U, uint32_t B, int lane, uint32_t &tile_carry)
{
    if (!SUB) return U;
    (void)lane;                                              // (precondition of the asm form: all 64 lanes active)
    const uint32_t P = ~B;
    const uint32_t X = U | P;
#ifdef TQ_NO_ASM
    const uint64_t s64 = (uint64_t)X + (uint64_t)U;
    const uint32_t sum = (uint32_t)s64;
    const uint64_t Gm = __ballot((uint32_t)(s64 >> 32) != 0u);      // lanes whose in-lane add carries out: "generate"
    const uint32_t cin0 = sum ^ X ^ U;
    const uint32_t seen_local = P & cin0;
    const uint64_t Pm = __ballot(B == 0);
    const uint64_t Xm = Gm | Pm;
    const uint32_t cin_tile = __builtin_amdgcn_readfirstlane(tile_carry) != 0u ? 1u : 0u;
    const unsigned __int128 t128 = (unsigned __int128)Xm + (unsigned __int128)Gm + cin_tile;
    const uint64_t cinm = (uint64_t)t128 ^ Xm ^ Gm;
    tile_carry = (uint32_t)(t128 >> 64);
    const uint32_t notfirst = B | (0u - B);
    const uint32_t inherit = ((cinm >> lane) & 1ull) ? notfirst : 0xFFFFFFFFu;
#else
    // in-lane: X + U; its carry-out IS t(31), so the per-lane "generate" flags of the lane-to-lane chain come out of
    // the add itself as a 64-bit lane mask (the sdst of v_add_co_u32) -- no bit fiddling, no compare, no ballot
    uint32_t sum;
    uint64_t Gm;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(sum), "=s"(Gm) : "v"(X), "v"(U));
    const uint32_t cin0 = sum ^ X ^ U;                       // carry into each bit, lane carry-in = 0
    const uint32_t seen_local = P & cin0;
    // cross-lane: T(l) = gen(l) | (allprop(l) & T(l-1)): the same adder trick on two 64-bit lane masks, one full
    // adder on the scalar unit -- carry-in = the carry out of the previous 2048-site step, put into SCC; the carry
    // out of lane 63 is the SCC the second s_addc leaves (the compiler's own u64 add-with-overflow compares on the
    // vector unit and keeps the step-to-step carry in a VGPR)
    const uint64_t Pm = __ballot(B == 0);
    const uint64_t Xm = Gm | Pm;
    uint32_t lo, hi, cout;
    asm("s_cmp_lg_u32 %3, 0\n\t"
        "s_addc_u32 %0, %4, %6\n\t"
        "s_addc_u32 %1, %5, %7\n\t"
        "s_cselect_b32 %2, 1, 0"
        : "=&s"(lo), "=&s"(hi), "=s"(cout)
        : "s"(__builtin_amdgcn_readfirstlane(tile_carry)), "s"((uint32_t)Xm), "s"((uint32_t)(Xm >> 32)), "s"((uint32_t)Gm), "s"((uint32_t)(Gm >> 32))
        : "scc");
    const uint64_t cinm = (((uint64_t)hi << 32) | lo) ^ Xm ^ Gm;      // carry into each lane: bit l = lane l inherits "seen"
    tile_carry = cout;                                       // carry out of lane 63
    // sites before this lane's first run-begin inherit the incoming "seen" state: B | -B marks the first run-begin
    // and everything above it (0 when the lane has none), its complement is that first segment
    const uint32_t notfirst = B | (0u - B);
    // cinm is used as what it is -- a lane mask -- by one v_cndmask (the compiler would shift it by the lane id)
    uint32_t inherit;
    asm("v_cndmask_b32_e64 %0, -1, %1, %2" : "=v"(inherit) : "v"(notfirst), "s"(cinm));      // cin ? notfirst : ~0
#endif
    return U & ~seen_local & inherit;                        // = U & ~(seen_local | (cin ? ~notfirst : 0))
}

template <bool SUB>
__device__ __forceinline__ uint32_t count_mask(const uint4 &pa, const uint4 &pb, const uint4 &pc, const uint4 &pd,
                                               int lane, uint32_t &tile_carry, uint32_t inv = 0u)
{
    const uint32_t M = pa.x | pb.x | pc.x | pd.x;
    const uint32_t V = (pa.y ^ pb.y) | (pa.z ^ pb.z) | (pa.y ^ pc.y) | (pa.z ^ pc.z) |
                       (pa.y ^ pd.y) | (pa.z ^ pd.z) | inv;
    return count_from_candidates<SUB>(V & ~M, pa.w, lane, tile_carry);
}

// the same with the (a,b) part pre-combined by the workgroup: r1 = {p0a, p1a, Ma|Mb, (p0a^p0b)|(p1a^p1b)}
template <bool SUB, typename PC>
__device__ __forceinline__ uint32_t count_mask_shared(const uint4 &r1, uint32_t B, const PC &pc, const PC &pd,
                                                      int lane, uint32_t &tile_carry)
{
    const uint32_t M = r1.z | pc.x | pd.x;
    const uint32_t V = r1.w | (r1.x ^ pc.y) | (r1.y ^ pc.z) | (r1.x ^ pd.y) | (r1.y ^ pd.z);
    return count_from_candidates<SUB>(V & ~M, B, lane, tile_carry);
}

// METHOD 0: one EXEC-masked ds_add per site slot (32 per step, whatever the density).
// METHOD 1: the lane parks its 32 pattern bytes in LDS and walks the set bits of C: the number of
//           ds_add per step is the largest per-lane count in the wave (~10 of 32 in subsample
//           mode, where at most one site per locus run is counted).
// METHOD 2 / 3: timing diagnostics (no histogram / no per-step barrier); results are wrong.
constexpr int PAT_STRIDE = 36;   // bytes per lane in the pattern park: 9 dwords, so that stores and the byte reads of
                                 // lanes that sit at the same site index never share a bank (a 32-byte stride with
                                 // two b128 stores is cheaper in isolation, tools/probe_lds.hip, but 14 % slower here:
                                 // the first counted sites of neighbouring lanes tend to have similar indices)

// LDS byte offset of a __shared__ object (the low half of its flat address: the LDS aperture is 4 GiB aligned)
__device__ __forceinline__ uint32_t lds_offset(const void *p) { return (uint32_t)(uintptr_t)p; }

// The set-bit walk of METHOD 1: for every set bit i of `c`, hist[park[i]] += 1 (park = the lane's 32 parked pattern
// bytes, hist = the wave's 256 u32 bins; both given as LDS byte offsets).  Hand-written because the loop is half of
// the scan kernel's vector instructions and the compiler's version spends 6.3 of them per counted site (plus a dozen
// scalar ones for its two-phase control flow); this one spends 5:
//   v_ffbl (index of the lowest set bit) . v_and (clear it, with c-1 from the previous trip) . v_add (byte address)
//   . v_add_co (c-1 for the next trip AND "c != 0" as its carry-out: adding 0xFFFFFFFF carries iff c >= 1)
//   . v_lshl_add (bin address)
// and leaves the loop by and-ing EXEC with that carry mask on the scalar unit.  The wave runs for its longest lane
// (8.2 trips against a mean of 5.6 counted sites on c3); LDS operations of one wave execute in order, so the byte
// read sees the park stores issued just before without a wait.
// TRANSPOSED park: the wave's 8 pattern dwords per lane are stored row-major [dword j][lane] (256-byte rows), so a
// byte read touches bank (lane mod 32) whatever its site index -- conflict-free, where the lane-contiguous layout
// (stride 36 bytes) pays 2.5 extra LDS cycles per read instruction on c3 (rocprofv3: SQ_LDS_BANK_CONFLICT of the
// walk without atomics) -- at the price of one more vector instruction per counted site for the address
// lanebase + (i >> 2) * 256 + (i & 3): the wave's park is 2 KiB-aligned, so the address is a bit field
// [k = i & 3 : bits 0-1][lane : 2-7][j = i >> 2 : 8-10][wave : 11+] = v_bfi(0x703, i | i << 6, lanebase).
// Wave priority during the walk (s_setprio): a wave in its walk is a chain of LDS round trips, a wave outside it issues
// vector instructions back to back; raised priority for the walking waves keeps the LDS pipe fed (c3 scan 5.99 -> 5.90 ms;
// the opposite setting 6.05 ms; A/B builds with -DTQ_WALK_PRIO=0|1|2 through TQ_LIB_PATH).
#ifndef TQ_WALK_PRIO
#define TQ_WALK_PRIO 1
#endif
#if TQ_WALK_PRIO == 1
#define TQ_WALK_PRIO_ON "s_setprio 2\n\t"
#define TQ_WALK_PRIO_OFF "s_setprio 0\n\t"
#elif TQ_WALK_PRIO == 2
#define TQ_WALK_PRIO_ON "s_setprio 0\n\t"
#define TQ_WALK_PRIO_OFF "s_setprio 2\n\t"
#else
#define TQ_WALK_PRIO_ON
#define TQ_WALK_PRIO_OFF
#endif
#ifdef TQ_NO_ASM
#define TQ_PARK_ADDTID 0
constexpr bool USE_ASM = false;
#else
constexpr bool USE_ASM = true;
#endif
#ifndef TQ_PARK_ADDTID
#define TQ_PARK_ADDTID 1
#endif
template <bool ATOMICS = true, bool TRANSPOSED = false>
__device__ __forceinline__ void walk_set_bits(uint32_t c, uint32_t park_off, uint32_t hist_off)
{
    if (ATOMICS && TRANSPOSED) {
        uint32_t t, i, j, b;
        uint64_t save;
        uint32_t one = 1u, k252 = 0x703u;                       // address bits taken from the site index: k (0-1) and j (8-10)
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            TQ_WALK_PRIO_ON
            "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execz 1f\n"
            "0:\n\t"
            "v_ffbl_b32_e32 %[i], %[c]\n\t"
            "v_and_b32_e32 %[c], %[c], %[t]\n\t"
            "v_lshl_or_b32 %[j], %[i], 6, %[i]\n\t"              // i = 4 j + k: j now also at bits 8-10 (k at 0-1)
            "v_bfi_b32 %[i], %[k252], %[j], %[park]\n\t"         // bits 0-1 and 8-10 from there, the rest = lane base
            "ds_read_u8 %[b], %[i]\n\t"
            "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_lshl_add_u32 %[b], %[b], 2, %[hist]\n\t"
            "ds_add_u32 %[b], %[one]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execnz 0b\n"
            "1:\n\t"
            TQ_WALK_PRIO_OFF
            "s_mov_b64 exec, %[save]"
            : [c] "+v"(c), [t] "=&v"(t), [i] "=&v"(i), [j] "=&v"(j), [b] "=&v"(b), [save] "=&s"(save)
            : [park] "v"(park_off), [hist] "s"(hist_off), [one] "v"(one), [k252] "s"(k252)
            : "vcc", "memory");
        return;
    }
    if (!ATOMICS) {          // timing diagnostic (METHOD 4): the walk and its byte reads without the histogram increments
        uint32_t t, i, b, acc = 0;
        uint64_t save;
        asm volatile(
            "s_mov_b64 %[save], exec\n\t"
            "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execz 1f\n"
            "0:\n\t"
            "v_ffbl_b32_e32 %[i], %[c]\n\t"
            "v_and_b32_e32 %[c], %[c], %[t]\n\t"
            "v_add_u32_e32 %[i], %[park], %[i]\n\t"
            "ds_read_u8 %[b], %[i]\n\t"
            "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_lshl_add_u32 %[acc], %[b], 2, %[acc]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_cbranch_execnz 0b\n"
            "1:\n\t"
            "s_mov_b64 exec, %[save]"
            : [c] "+v"(c), [t] "=&v"(t), [i] "=&v"(i), [b] "=&v"(b), [save] "=&s"(save), [acc] "+v"(acc)
            : [park] "v"(park_off)
            : "vcc", "memory");
        asm volatile("" ::"v"(acc));
        return;
    }
    uint32_t t, i, b;
    uint64_t save;
    uint32_t one = 1u;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execz 1f\n"
        "0:\n\t"
        "v_ffbl_b32_e32 %[i], %[c]\n\t"
        "v_and_b32_e32 %[c], %[c], %[t]\n\t"
        "v_add_u32_e32 %[i], %[park], %[i]\n\t"
        "ds_read_u8 %[b], %[i]\n\t"
        "v_add_co_u32_e32 %[t], vcc, -1, %[c]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshl_add_u32 %[b], %[b], 2, %[hist]\n\t"
        "ds_add_u32 %[b], %[one]\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_cbranch_execnz 0b\n"
        "1:\n\t"
        "s_mov_b64 exec, %[save]"
        : [c] "+v"(c), [t] "=&v"(t), [i] "=&v"(i), [b] "=&v"(b), [save] "=&s"(save)
        : [park] "v"(park_off), [hist] "s"(hist_off), [one] "v"(one)
        : "vcc", "memory");
}

// pat[j] holds the 8-bit patterns (a<<6|b<<4|c<<2|d) of sites 4j..4j+3 of this lane, one per byte
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};

// `after_build` runs once the patterns are in their final place (registers for METHOD 0, the LDS park
// for METHOD 1) and before the histogram increments: the cooperative kernel issues the next step's
// row loads there, so that they fly under the increments and need no second register set.
template <int NREP, int METHOD, typename Hook = NoHook, bool PARK_T = false>
__device__ __forceinline__ void hist_patterns(const uint32_t (&pat)[8], uint32_t C, uint32_t *hrep, uint8_t *park,
                                              Hook after_build = Hook())
{
    if (METHOD == 2 || METHOD == 5) {
        if (METHOD == 5) {               // timing diagnostic: the park stores, no walk
            uint32_t *pw = reinterpret_cast<uint32_t *>(park);
#pragma unroll
            for (int j = 0; j < 8; ++j) pw[j] = pat[j];
        }
        uint32_t acc = C;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= pat[j];
        asm volatile("" ::"v"(acc));
        after_build();
    } else if (METHOD == 0) {
        after_build();
        if (NREP == 1 && USE_ASM) {
            // one EXEC-masked ds_add per site slot, most significant bit first: v_add_co c, vcc, c, c shifts the count
            // mask and hands the slot's bit over AS the lane mask (its carry-out), so a slot is 3 vector instructions
            // (that add, the byte extract, the bin address) where the compiler's version tests the bit with an AND and
            // a compare (4)
            uint32_t c = C, a, one = 1u;
            const uint32_t hist_off = lds_offset(hrep);
            uint64_t save;
            asm volatile("s_mov_b64 %0, exec" : "=s"(save));
#define TQ_SLOT(J, K)                                                                                          \
            asm volatile("v_add_co_u32_e32 %[c], vcc, %[c], %[c]\n\t"                                            \
                         "v_bfe_u32 %[a], %[p], " #K "*8, 8\n\t"                                                  \
                         "v_lshl_add_u32 %[a], %[a], 2, %[hist]\n\t"                                              \
                         "s_and_b64 exec, %[save], vcc\n\t"                                                       \
                         "ds_add_u32 %[a], %[one]\n\t"                                                            \
                         "s_mov_b64 exec, %[save]"                                                                 \
                         : [c] "+v"(c), [a] "=&v"(a)                                                               \
                         : [p] "v"(pat[J]), [hist] "s"(hist_off), [one] "v"(one), [save] "s"(save)                 \
                         : "vcc", "memory");
#define TQ_SLOT4(J) TQ_SLOT(J, 3) TQ_SLOT(J, 2) TQ_SLOT(J, 1) TQ_SLOT(J, 0)
            TQ_SLOT4(7) TQ_SLOT4(6) TQ_SLOT4(5) TQ_SLOT4(4) TQ_SLOT4(3) TQ_SLOT4(2) TQ_SLOT4(1) TQ_SLOT4(0)
#undef TQ_SLOT4
#undef TQ_SLOT
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (C & (1u << (4 * j + k)))
                        __hip_atomic_fetch_add(&hrep[((pat[j] >> (8 * k)) & 0xFFu) * NREP], 1u, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    } else {
        if (PARK_T && NREP == 1 && TQ_PARK_ADDTID) {
            // transposed park, row j = 64 consecutive dwords: exactly the addressing of ds_write_addtid_b32 (M0 + offset +
            // 4 * lane, no address register).  An LDS store moves its address and data registers to the LDS at 2 cycles
            // per dword per wave-instruction (MI355X_MICROARCH.md, LDS): 8 x 2 cycles here against 4 x 6 for the
            // ds_write2st64_b32 pairs the compiler makes of the plain stores -- and tools/probe_slots.hip shows that this
            // transfer, not the LDS array, is what an LDS instruction of this kernel costs.
            const uint32_t base = __builtin_amdgcn_readfirstlane(lds_offset(park));     // lane 0's slot = the wave's row 0
            asm volatile("s_mov_b32 m0, %[base]\n\t"
                         "s_nop 0\n\t"
                         "ds_write_addtid_b32 %[p0]\n\t"
                         "ds_write_addtid_b32 %[p1] offset:256\n\t"
                         "ds_write_addtid_b32 %[p2] offset:512\n\t"
                         "ds_write_addtid_b32 %[p3] offset:768\n\t"
                         "ds_write_addtid_b32 %[p4] offset:1024\n\t"
                         "ds_write_addtid_b32 %[p5] offset:1280\n\t"
                         "ds_write_addtid_b32 %[p6] offset:1536\n\t"
                         "ds_write_addtid_b32 %[p7] offset:1792"
                         :
                         : [base] "s"(base), [p0] "v"(pat[0]), [p1] "v"(pat[1]), [p2] "v"(pat[2]), [p3] "v"(pat[3]),
                           [p4] "v"(pat[4]), [p5] "v"(pat[5]), [p6] "v"(pat[6]), [p7] "v"(pat[7])
                         : "memory", "m0");
        } else {
            uint32_t *pw = reinterpret_cast<uint32_t *>(park);
#pragma unroll
            for (int j = 0; j < 8; ++j) pw[PARK_T ? j * WAVE : j] = pat[j];     // PARK_T: `park` = row 0 of the wave + lane * 4
        }
        after_build();
        if (NREP == 1 && USE_ASM) {
            walk_set_bits<METHOD != 4, PARK_T>(C, lds_offset(park), lds_offset(hrep));
        } else {
            // set-bit walk in plain C++ (replicated histograms: the one-wave-per-quartet kernel's A/B option; every form
            // under -DTQ_NO_ASM).  Transposed park: byte i of the lane sits in row i >> 2 (256 bytes per row) at lane * 4 + (i & 3)
            uint32_t c = C;
            while (c) {
                const int i = __builtin_ctz(c);
                const uint32_t b = PARK_T ? park[(i >> 2) * (WAVE * 4) + (i & 3)] : park[i];
                c &= c - 1;
                if (METHOD != 4)
                    __hip_atomic_fetch_add(hrep + b * NREP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

// base codes are 0..3, so (a<<6|b<<4|c<<2|d) never crosses a byte and + == | (one v_lshl_add per field)
__device__ __forceinline__ uint32_t pat4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return ((((((a << 2) + b) << 2) + c) << 2) + d);
}

template <int NREP, bool SUB, int METHOD>
__device__ __forceinline__ void process_tile(const TileRegs &t, int lane, uint32_t &tile_carry, uint32_t *hrep,
                                             uint8_t *park, uint32_t inv)
{
    const uint32_t C = count_mask<SUB>(t.pa, t.pb, t.pc, t.pd, lane, tile_carry, inv);
    uint32_t pat[8];
    pat[0] = pat4(t.a0.x, t.b0.x, t.c0.x, t.d0.x);
    pat[1] = pat4(t.a0.y, t.b0.y, t.c0.y, t.d0.y);
    pat[2] = pat4(t.a0.z, t.b0.z, t.c0.z, t.d0.z);
    pat[3] = pat4(t.a0.w, t.b0.w, t.c0.w, t.d0.w);
    pat[4] = pat4(t.a1.x, t.b1.x, t.c1.x, t.d1.x);
    pat[5] = pat4(t.a1.y, t.b1.y, t.c1.y, t.d1.y);
    pat[6] = pat4(t.a1.z, t.b1.z, t.c1.z, t.d1.z);
    pat[7] = pat4(t.a1.w, t.b1.w, t.c1.w, t.d1.w);
    hist_patterns<NREP, METHOD>(pat, C, hrep, park);
}

// Double-buffered scan: the loads of step t+1 are issued before step t is processed; the
// sched_barriers keep the compiler from sinking them next to their first use.  The prefetch is
// unconditional (the index is clamped, so the last step re-reads its own tile): a conditional
// load would make the compiler merge the buffers with copies that wait for the loads at once.
template <int NREP, bool SUB, int METHOD>
__device__ __forceinline__ void scan_quartet(const DevData &d, const uint32_t (&q)[4], uint32_t *hist,
                                             uint8_t *park, int lane)
{
    uint32_t *hrep = hist + (lane & (NREP - 1));
    uint32_t tile_carry = 0;
    const int last = d.ntiles - 1;
    TileRegs A, B;
    load_tile(A, d, q, 0, lane);
    for (int t = 0; t < d.ntiles; t += 2) {
        load_tile(B, d, q, min(t + 1, last), lane);
        __builtin_amdgcn_sched_barrier(0);
        process_tile<NREP, SUB, METHOD>(A, lane, tile_carry, hrep, park, d.inv);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 >= d.ntiles) break;
        load_tile(A, d, q, min(t + 2, last), lane);
        __builtin_amdgcn_sched_barrier(0);
        process_tile<NREP, SUB, METHOD>(B, lane, tile_carry, hrep, park, d.inv);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// cm layout: u32 [Q][256], cm[q][64*i0 + 16*i1 + 4*i2 + i3] = number of counted sites with pattern
// (i0,i1,i2,i3) -- i.e. mats[0] of resolve_quartets.py:55-64 / :89-95 in row-major order.
template <int NREP, bool SUB, int METHOD>
__global__ void __launch_bounds__(WAVE)
tq_scan_kernel(DevData d, const uint32_t *__restrict__ quartets, const uint32_t *__restrict__ order, int64_t Q,
               uint32_t *__restrict__ cm)
{
    __shared__ uint32_t hist[256 * NREP + (METHOD ? WAVE * PAT_STRIDE / 4 : 0)];
    const int lane = threadIdx.x;
    uint8_t *park = reinterpret_cast<uint8_t *>(hist + 256 * NREP) + lane * PAT_STRIDE;
    for (int i = lane; i < 256 * NREP; i += WAVE) hist[i] = 0;
    __syncthreads();
    for (int64_t it = blockIdx.x; it < Q; it += gridDim.x) {
        // waves that run together work on neighbours of the (a,b)-sorted order, so rows a and b
        // are L2 hits for all of them; results go to the quartet's original slot
        const int64_t qi = order ? (int64_t)order[it] : it;
        const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[qi];
        uint32_t q[4];
        q[0] = __builtin_amdgcn_readfirstlane(qv.x);
        q[1] = __builtin_amdgcn_readfirstlane(qv.y);
        q[2] = __builtin_amdgcn_readfirstlane(qv.z);
        q[3] = __builtin_amdgcn_readfirstlane(qv.w);
        const uint32_t T = (uint32_t)d.T;
        const bool bad = (q[0] >= T) | (q[1] >= T) | (q[2] >= T) | (q[3] >= T);
        uint32_t *out = cm + qi * 256;
        if (bad) {                              // flagged by the SVD kernel; never dereferenced
#pragma unroll
            for (int k = 0; k < 4; ++k) out[lane + WAVE * k] = 0;
            continue;
        }
        scan_quartet<NREP, SUB, METHOD>(d, q, hist, park, lane);
        __syncthreads();
        // fold the replicas, clear them for the next quartet, store the 256 counts (coalesced)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bin = lane + WAVE * k;
            uint32_t s = 0;
#pragma unroll
            for (int r = 0; r < NREP; ++r) {
                const int rr = (r + lane) & (NREP - 1);
                s += hist[bin * NREP + rr];
                hist[bin * NREP + rr] = 0;
            }
            out[bin] = s;
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------
// kernel 1, workgroup-cooperative form: NW wavefronts = NW neighbours of the (a,b,c)-sorted order,
// one block of NW quartets per workgroup (grid = number of blocks, dispatched in sorted order).
// Quartets that share their first two taxa share two of their four rows, so what depends only on
// (a,b) is fetched and pre-combined ONCE per workgroup per 2048-site step by three 64-lane jobs dealt
// round-robin to the waves: jobs 0-1 turn the nibble codes of a and b into the pattern
// partial ((a<<2)+b)<<4 of 16 sites per lane, job 2 turns the 12-byte plane records of a and b and
// the run-begin word into {p0a, p1a, Ma|Mb, (p0a^p0b)|(p1a^p1b)} + B.  The image (3.25 KiB) is
// double-buffered in LDS with one barrier per step.  Every wave streams its own rows c and d
// (nibble copy) and their 12-byte plane records straight to registers (one register set:
// the loads of step t+1 are issued inside step t once the pattern bytes are parked, 72 VGPRs = 7 waves/SIMD):
// 3.5 KiB per wave-step + 3.75/NW KiB shared (12 KiB for independent waves on byte rows).  A wave
// whose (a,b) differs from the leader's (group boundary in the sorted order) builds its own partial
// from the byte rows instead; it still takes part in the loads and barriers.  On c3 the kernel keeps
// VALU 77 %, LDS 78 % and the L2 -> CU path ~90 % busy at the same time.
// ------------------------------------------------------------------------------------
// LLVM vector types (HIP's uint3 / uint4 are structs of scalars): a value of such a type stays ONE 96- / 128-bit
// register tuple through the loop, so the tuple a load writes and the loop-carried value coalesce.  With scalars the
// allocator copied two words of each 12-byte record right behind the load -- and that copy dragged an `s_waitcnt vmcnt`
// in front of the histogram phase: three of the four prefetch loads of every step were waited for a few instructions
// after their issue (ISA of rounds 2-3).
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct OwnRegs {
    u32x4 c, d;              // nibble codes of rows c and d (32 sites each)
    u32x3 pc, pd;            // their plane records {miss, p0, p1}
};
struct OwnRegsS {            // the same as HIP structs (the two-quartets-per-wave kernel)
    uint4 c, d, pc, pd;
};
__device__ __forceinline__ u32x4 ldv16(const uint8_t *base, uint32_t off)
{
    return *reinterpret_cast<const u32x4 *>(base + off);
}
__device__ __forceinline__ u32x3 ldv12(const uint8_t *base, uint32_t off)
{
    return *reinterpret_cast<const u32x3 *>(base + off);
}

// 16-byte load at a wave-uniform base + 32-bit per-lane byte offset (lets the compiler use the
// SGPR-base addressing form instead of 64-bit VGPR pointer arithmetic for every load; the host
// guarantees T*Sp < 2^32 before it selects this kernel)
__device__ __forceinline__ uint4 ld16(const uint8_t *base, uint32_t off)
{
    return *reinterpret_cast<const uint4 *>(base + off);
}

// makes the compiler treat the four words as defined HERE (copies into store tuples and the waits for the loads that
// produced them cannot be scheduled earlier)
__device__ __forceinline__ void pin4(uint4 &v)
{
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

// 0x0f0f0f0f in an SGPR (as a literal it cannot be encoded in the three-operand v_and_or_b32)
__device__ __forceinline__ uint32_t sgpr_const_0f()
{
#ifdef TQ_NO_ASM
    return 0x0f0f0f0fu;
#else
    uint32_t m;
    asm("s_mov_b32 %0, 0x0f0f0f0f" : "=s"(m));
    return m;
#endif
}

// (a & m) | c in one instruction; m must sit in an SGPR (the compiler emits v_and + v_or otherwise)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t m, uint32_t c)
{
#ifdef TQ_NO_ASM
    return (a & m) | c;
#else
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m), "v"(c));
    return r;
#endif
}

// 12-byte load (one compact plane record {miss, p0, p1}); .w of the result is 0
__device__ __forceinline__ uint4 ld12(const uint8_t *base, uint32_t off)
{
    const uint3 v = *reinterpret_cast<const uint3 *>(base + off);
    return make_uint4(v.x, v.y, v.z, 0u);
}

// per-lane byte offsets of a wave's own rows c, d (nibble array) and their compact plane records
struct OwnOff {
    uint32_t c, d, pc, pd;
};

__device__ __forceinline__ void load_own(OwnRegs &r, const uint8_t *nib, const uint8_t *planes3,
                                         const OwnOff &o, int tile)
{
    const uint32_t tn = (uint32_t)tile * (TILE / 2), tp = (uint32_t)tile * (WAVE * 12);
#ifndef TQ_DIAG_NO_NIB
    r.c = ldv16(nib, o.c + tn);                  // plain codes; the factor 4 is the shift of a v_lshl_add_u32
    r.d = ldv16(nib, o.d + tn);
#endif
    r.pc = ldv12(planes3, o.pc + tp);
    r.pd = ldv12(planes3, o.pd + tp);
#ifdef TQ_DIAG_NO_NIB
    // A/B build only (wrong results): the two nibble loads of the own rows dropped -- the upper bound of what a layout
    // with ONE record per taxon and lane-step could gain on the load side, before its extra vector work
    (void)nib; (void)tn;
    r.c = u32x4{r.pc.y, r.pc.z, r.pc.y >> 2, r.pc.z >> 2} & 0x33333333u;      // base bits of other sites: uniform codes
    r.d = u32x4{r.pd.y, r.pd.z, r.pd.y >> 2, r.pd.z >> 2} & 0x33333333u;
#endif
}

// (c<<2|d) code bytes of 8 sites from one nibble dword of each row: lo = sites 0-3, hi = sites 4-7
__device__ __forceinline__ void cd_pair(uint32_t cw, uint32_t dw, uint32_t &lo, uint32_t &hi)
{
    lo = ((cw & 0x0F0F0F0Fu) << 2) + (dw & 0x0F0F0F0Fu);
    hi = (((cw >> 4) & 0x0F0F0F0Fu) << 2) + ((dw >> 4) & 0x0F0F0F0Fu);
}

// LDS image of the shared part of one step: abp = pattern partial ((a<<2)+b)<<4 as 2 x 64 uint4
// panels (sites 0-15 / 16-31 of every lane), then the plane records of a and of b (64 uint4 each)
constexpr int SHARED_SLOTS = 208;
// SHC ("share row c"): when every quartet of a block has the leader's (a,b,c) -- the normal case for
// lexicographic enumerations, combinations.py:40-55, and frequent in an (a,b,c)-sorted random sample -- one more
// job fetches the nibble codes and the plane record of row c once per workgroup and step into two more
// panels of the image (slots 208-271, 272-335); the waves then read their row c from LDS and stream only row d
// from the L2: 2.6 KiB instead of 4.4 KiB per quartet-step through the L2 -> CU path.
// MEASURED SLOWER (profiles/r02_*/share_c_ab.txt): c2 lexicographic, where every block qualifies, 1.75 -> 2.13 ms;
// c3 random 6.7 -> 8.3 ms.  Two more LDS reads per wave-step on an LDS pipe that is ~80 % busy, 85 instead of 72
// VGPRs and 24 instead of 20 KiB of LDS (6 instead of 7 workgroups per CU) cost more than the 40 % fewer bytes from
// L2 save: the kernel is not bound by bytes.  Off by default (option "share_c"), parity-tested.
constexpr int SHARED_SLOTS_C = SHARED_SLOTS + 128;

template <bool SUB, int METHOD, int NW, bool SHC = false, bool PARK_T = false>
__global__ void __launch_bounds__(NW *WAVE)
tq_scan_wg_kernel(DevData d, const uint32_t *__restrict__ quartets, const uint32_t *__restrict__ order, int64_t Q,
                  uint32_t *__restrict__ cm, int64_t xcd_chunk)
{
    static_assert(NW >= 1 && NW <= 16, "waves per workgroup");
    static_assert(!SHC || NW >= 3, "the row-c job needs a third wave");
    __shared__ uint4 shared_ab[2][SHC ? SHARED_SLOTS_C : SHARED_SLOTS];
    // (two copies of the wave's histogram, odd and even lanes apart, were measured: SQ_LDS_ADDR_CONFLICT 563M -> 350M
    // per dispatch, SQ_LDS_BANK_CONFLICT and the LDS-busy cycles unchanged to the digit, 6.42 -> 6.58 ms with the lower
    // occupancy -- profiles/r03_scan/)
    __shared__ uint32_t hist_all[NW][256];
    __shared__ __attribute__((aligned(2048))) uint32_t park_all[NW][PARK_T ? WAVE * 8 : WAVE * PAT_STRIDE / 4];
    const int tid = threadIdx.x;
    // the wave's number as a scalar: everything derived from it (does this wave have a quartet, does it share the
    // leader's rows, its histogram) is then wave-uniform for the compiler too -- scalar branches instead of EXEC
    // masking, and the step-to-step carry of subsample mode stays in an SGPR
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    uint32_t *hist = hist_all[w];
    uint8_t *park = reinterpret_cast<uint8_t *>(park_all[w]) + lane * (PARK_T ? 4 : PAT_STRIDE);
    for (int i = lane; i < 256; i += WAVE) hist[i] = 0;
    const uint32_t T = (uint32_t)d.T;
    const int last = d.ntiles - 1;
    const int64_t nblk = (Q + NW - 1) / NW;
    const uint8_t *rows = d.rows;
    const uint8_t *nib = d.nib;
    const uint8_t *planes = reinterpret_cast<const uint8_t *>(d.planes);
    const uint8_t *planes3 = reinterpret_cast<const uint8_t *>(d.planes3);
    const uint32_t pitch = (uint32_t)d.pitch, npitch = pitch / 2, wpitch = (uint32_t)d.W * 16u,
                   w3pitch = (uint32_t)d.W * 12u;
    // cooperative jobs per step: 0,1 = nibble codes of rows a and b for sites 0-15 / 16-31
    // of every lane -> abp panels 0,1; 2 = plane records of a and b (+ run-begin bits) -> r1, B.
    // Wave w takes the jobs j with j % NW == w.  A vector load costs the CU's texture-address path ~18 cycles per
    // wave-instruction whatever its width (tools/probe_ta.hip) and that path is what bounds the load side of this
    // kernel (TA_TA_BUSY 95 % of the CU-busy cycles with the histogram taken out, profiles/r03_scan), so the shared
    // part is fetched with the FEWEST instructions: job 0 = nibble codes of a and of b, one 16-byte load each (it was
    // four 8-byte loads in two jobs); job 1 = the 16-byte plane records {miss, p0, p1, run-begin} of a and b (it was two
    // 12-byte records plus a run-begin load).  9 -> 4 loads per workgroup-step, 6.25 -> 5 per wave-step.
    // With SHC: job 2 = nibble codes and plane record of the block's common row c.
    constexpr int NJOB = SHC ? 3 : 2, JPW = (NJOB + NW - 1) / NW;        // jobs per wave
    __syncthreads();

    const int64_t nslot = xcd_chunk > 0 ? 8 * xcd_chunk : nblk;
    for (int64_t blk0 = blockIdx.x; blk0 < nslot; blk0 += gridDim.x) {
        // Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  With xcd_chunk > 0
        // the ids that land on one XCD walk one contiguous eighth of the sorted order, so a row pair
        // (a,b) is pulled into one L2 instead of eight.
        int64_t blk = blk0;
        if (xcd_chunk > 0) {
            const int64_t x = blk0 & 7, j = blk0 >> 3;
            blk = x * xcd_chunk + j;
            if (j >= xcd_chunk || blk >= nblk) continue;       // uniform for the whole workgroup
        }
        // leader = first quartet of the block; its (a,b) is what the workgroup shares
        const int64_t it0 = blk * NW;
        const int64_t lqi = order ? (int64_t)order[it0] : it0;
        const uint4 lq = reinterpret_cast<const uint4 *>(quartets)[lqi];
        uint32_t la = __builtin_amdgcn_readfirstlane(lq.x), lb = __builtin_amdgcn_readfirstlane(lq.y);
        const bool leader_ok = (la < T) & (lb < T);
        if (!leader_ok) la = lb = 0;
        // every wave looks at all NW quartets of the block, so all of them reach the same verdict without
        // talking to each other: do they all have the leader's (a,b,c)?
        bool shc = false;
        uint32_t lc = 0;
        if (SHC) {
            lc = __builtin_amdgcn_readfirstlane(lq.z);
            shc = leader_ok && lc < T && it0 + NW <= Q;
#pragma unroll
            for (int k = 1; k < NW; ++k) {
                const int64_t itk = min(it0 + k, Q - 1);
                const int64_t qk = order ? (int64_t)order[itk] : itk;
                const uint4 v = reinterpret_cast<const uint4 *>(quartets)[qk];
                shc = shc && __builtin_amdgcn_readfirstlane(v.x) == la && __builtin_amdgcn_readfirstlane(v.y) == lb &&
                      __builtin_amdgcn_readfirstlane(v.z) == lc && __builtin_amdgcn_readfirstlane(v.w) < T;
            }
            if (!shc) lc = 0;
        }
        // this wave's quartet
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
        // cooperative loads of this thread (values are passed and returned by value: address-taken
        // locals end up in scratch memory, which costs a memory round trip per step)
        auto job_of = [=](int i) { return w + i * NW; };                       // wave-uniform
        auto fetch_x = [=](int job, int tile) -> uint4 {
            if (job == 0) return ld16(nib, la * npitch + (uint32_t)lane * 16u + (uint32_t)tile * (TILE / 2));
            if (job == 1) return ld16(planes, la * wpitch + (uint32_t)lane * 16u + (uint32_t)tile * (WAVE * 16));
            if (SHC && job == 2 && shc) return ld16(nib, lc * npitch + (uint32_t)lane * 16u + (uint32_t)tile * (TILE / 2));
            return make_uint4(0, 0, 0, 0);
        };
        auto fetch_y = [=](int job, int tile) -> uint4 {
            if (job == 0) return ld16(nib, lb * npitch + (uint32_t)lane * 16u + (uint32_t)tile * (TILE / 2));
            if (job == 1) return ld16(planes, lb * wpitch + (uint32_t)lane * 16u + (uint32_t)tile * (WAVE * 16));
            if (SHC && job == 2 && shc) return ld12(planes3, lc * w3pitch + (uint32_t)lane * 12u + (uint32_t)tile * (WAVE * 12));
            return make_uint4(0, 0, 0, 0);
        };
        // what goes into the LDS image of one step (uint4 slots): abp panels 0-63 / 64-127 =
        // ((a<<2)+b)<<4 per site byte; r1 128-191 = {p0a, p1a, Ma|Mb, (p0a^p0b)|(p1a^p1b)};
        // run-begin words 192-207 (64 dwords)
        auto publish = [=](uint4 *buf, int job, uint4 x, uint4 y) {
            if (job == 0) {
                // a*4+b per nibble: codes are 0..3, so the packed word can be shifted as a whole
                const uint32_t h = 0xF0F0F0F0u;
                const uint32_t s0 = (x.x << 2) + y.x, s1 = (x.y << 2) + y.y, s2 = (x.z << 2) + y.z, s3 = (x.w << 2) + y.w;
                buf[lane] = make_uint4((s0 << 4) & h, s0 & h, (s1 << 4) & h, s1 & h);           // sites 0-15 of the lane
                buf[64 + lane] = make_uint4((s2 << 4) & h, s2 & h, (s3 << 4) & h, s3 & h);      // sites 16-31
            } else if (job == 1) {                           // x, y = {miss, p0, p1, run-begin} of a, of b
                buf[128 + lane] = make_uint4(x.y, x.z, x.x | y.x, (x.y ^ y.y) | (x.z ^ y.z));
                reinterpret_cast<uint32_t *>(buf + 192)[lane] = x.w;
            } else if (SHC && job == 2 && shc) {
                buf[SHARED_SLOTS + lane] = x;                  // nibble codes of row c
                buf[SHARED_SLOTS + 64 + lane] = y;             // its plane record {miss, p0, p1, 0}
            }
        };
        // a wave's own rows: c and d, or d alone when row c comes through the image
        auto load_mine = [=](OwnRegs &r, int tile) {
            if (SHC && shc) {
                const uint32_t tn = (uint32_t)tile * (TILE / 2), tp = (uint32_t)tile * (WAVE * 12);
                r.d = ldv16(nib, oo.d + tn);
                r.pd = ldv12(planes3, oo.pd + tp);
            } else {
                load_own(r, nib, planes3, oo, tile);
            }
        };

        // The step loop exists in up to four copies chosen by wave-uniform facts that do not change inside it: SPEC = the
        // wave's job (0, 1, none) for waves that work and share the leader's rows (FAST), and one generic copy for the rest
        // (a wave at a group boundary of the sorted order, a wave without a quartet; every wave of the A/B forms with
        // other job layouts).  Inside a FAST copy every load is unconditional, so the compiler's s_waitcnt counts are exact:
        // with the job and the boundary case as branches INSIDE the loop it assumes the fewest loads in flight at every
        // join and turns "wait for the rows requested a step ago" into vmcnt(0) -- the job waves then also wait for the
        // image loads they issued a few instructions earlier, an L2 round trip per step (ISA of rounds 2-3).
        auto run = [&](auto spec_tag, auto fast_tag) {
            constexpr int SPEC = decltype(spec_tag)::value;      // -1: jobs by the wave's number at run time
            constexpr bool FAST = decltype(fast_tag)::value;
            auto jobx = [=](int i) { return SPEC >= 0 ? (i == 0 ? SPEC : NJOB) : job_of(i); };
            // prologue: step 0 into buffer 0
            uint4 sx[JPW], sy[JPW];
    #pragma unroll
            for (int i = 0; i < JPW; ++i) {
                sx[i] = fetch_x(jobx(i), 0);
                sy[i] = fetch_y(jobx(i), 0);
            }
            OwnRegs A;
            load_mine(A, 0);
    #pragma unroll
            for (int i = 0; i < JPW; ++i) publish(shared_ab[0], jobx(i), sx[i], sy[i]);
            uint32_t tile_carry = 0;
            __syncthreads();

            auto step = [&](OwnRegs &own, int t, int tnext) {
                if (!(FAST || work)) return;
                uint4 ab0, ab1, r1;                            // abp panels (sites 0-15, 16-31), combined planes
                uint32_t Bw;                                   // run-begin bits
                if (FAST || shares) {
                    const uint4 *buf = shared_ab[t & 1];
                    ab0 = buf[lane];
                    ab1 = buf[64 + lane];
                    r1 = buf[128 + lane];
                    Bw = reinterpret_cast<const uint32_t *>(buf + 192)[lane];
                    if (SHC && shc) {
                        const uint4 cc = buf[SHARED_SLOTS + lane], pp = buf[SHARED_SLOTS + 64 + lane];
                        own.c = u32x4{cc.x, cc.y, cc.z, cc.w};
                        own.pc = u32x3{pp.x, pp.y, pp.z};
                    }
                } else {                                        // group boundary: private rows a and b
                    const uint32_t o0 = q[0] * pitch + (uint32_t)t * TILE + (uint32_t)lane * 16u;
                    const uint32_t o1 = q[1] * pitch + (uint32_t)t * TILE + (uint32_t)lane * 16u;
                    const uint4 a0 = ld16(rows, o0), a1 = ld16(rows, o0 + 1024u);
                    const uint4 b0 = ld16(rows, o1), b1 = ld16(rows, o1 + 1024u);
                    ab0 = make_uint4(((a0.x << 2) + b0.x) << 4, ((a0.y << 2) + b0.y) << 4, ((a0.z << 2) + b0.z) << 4,
                                     ((a0.w << 2) + b0.w) << 4);
                    ab1 = make_uint4(((a1.x << 2) + b1.x) << 4, ((a1.y << 2) + b1.y) << 4, ((a1.z << 2) + b1.z) << 4,
                                     ((a1.w << 2) + b1.w) << 4);
                    const uint4 pa = ld16(planes, q[0] * wpitch + (uint32_t)t * (WAVE * 16) + (uint32_t)lane * 16u);
                    const uint4 pb = ld16(planes, q[1] * wpitch + (uint32_t)t * (WAVE * 16) + (uint32_t)lane * 16u);
                    r1 = make_uint4(pa.y, pa.z, pa.x | pb.x, (pa.y ^ pb.y) | (pa.z ^ pb.z));
                    Bw = pa.w;
                }
                const uint32_t C = count_mask_shared<SUB>(r1, Bw, own.pc, own.pd, lane, tile_carry);
                // own.c and own.d hold the codes of rows c and d, one site per nibble (codes are 0..3, so the packed
                // word can be shifted as a whole): (c << 2) + d is the (c<<2|d) nibble, one v_lshl_add_u32 per dword;
                // the (a<<6|b<<4) byte of the same site sits in the high nibbles of abp
                // (the mask lives in an SGPR so that (s & m) | ab is one v_and_or_b32; as a literal it
                // cannot be encoded in a three-operand instruction and costs a second one)
                uint32_t m;
                m = sgpr_const_0f();
                const uint32_t s0 = (own.c.x << 2) + own.d.x, s1 = (own.c.y << 2) + own.d.y, s2 = (own.c.z << 2) + own.d.z,
                               s3 = (own.c.w << 2) + own.d.w;
                uint32_t pat[8];
                pat[0] = and_or(s0, m, ab0.x); pat[1] = and_or(s0 >> 4, m, ab0.y);
                pat[2] = and_or(s1, m, ab0.z); pat[3] = and_or(s1 >> 4, m, ab0.w);
                pat[4] = and_or(s2, m, ab1.x); pat[5] = and_or(s2 >> 4, m, ab1.y);
                pat[6] = and_or(s3, m, ab1.z); pat[7] = and_or(s3 >> 4, m, ab1.w);
                // (the first barrier keeps the scheduler from hoisting the loads above the last uses of the old values: it
                // then needs copies of the loaded words and puts them -- with their s_waitcnt -- right behind the loads, in
                // front of the histogram phase: the ISA of rounds 2-3 waited for three of the four loads there in every step)
                auto hook = [&]() {
                    __builtin_amdgcn_sched_barrier(0);
                    load_mine(own, tnext);
                    __builtin_amdgcn_sched_barrier(0);
                };
                hist_patterns<1, METHOD == 3 ? 1 : METHOD, decltype(hook), PARK_T>(pat, C, hist, park, hook);
            };

            for (int t = 0; t < d.ntiles; ++t) {
                // the shared pieces of step t+1 go to registers now and to LDS after this step's work; the own
                // rows of step t+1 are requested from inside step(), as soon as this step no longer needs the
                // registers they land in (the index is clamped: the last step re-reads its own tile)
                const int tn = min(t + 1, last);
    #pragma unroll
                for (int i = 0; i < JPW; ++i) {
                    sx[i] = fetch_x(jobx(i), tn);
                    sy[i] = fetch_y(jobx(i), tn);
                }
                __builtin_amdgcn_sched_barrier(0);
                step(A, t, tn);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int i = 0; i < JPW; ++i) {
                    // the image words of step t+1 are "defined" here for the compiler: the copies that assemble the 16-byte LDS
                    // stores (and the s_waitcnt they need) land here and not right behind the loads at the top of the step
                    if (jobx(i) < NJOB) {
                        pin4(sx[i]);
                        pin4(sy[i]);
                    }
                    publish(shared_ab[(t + 1) & 1], jobx(i), sx[i], sy[i]);
                }
                if (METHOD != 3) __syncthreads();      // METHOD 3 = timing diagnostic without the barrier
            }
        };
        {
            using std::integral_constant;
            constexpr bool CAN_SPEC = !SHC && JPW == 1 && NW >= 2;
            if (CAN_SPEC && shares) {
                if (w == 0) run(integral_constant<int, 0>{}, integral_constant<bool, true>{});
                else if (w == 1) run(integral_constant<int, 1>{}, integral_constant<bool, true>{});
                else run(integral_constant<int, CAN_SPEC ? NJOB : -1>{}, integral_constant<bool, true>{});
            } else {
                run(integral_constant<int, -1>{}, integral_constant<bool, false>{});
            }
        }
        // store the 256 counts of this wave's quartet and clear its histogram
        if (have) {
            uint32_t *out = cm + qi * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bin = lane + WAVE * k;
                __builtin_nontemporal_store(work ? hist[bin] : 0u, &out[bin]);   // read next by another kernel: keep it out of L2
                hist[bin] = 0;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------
// kernel 1, two quartets per wavefront ("pair" form of the cooperative kernel).
// The load side of tq_scan_wg_kernel is bound by the number of vector load instructions a CU can issue (~18 cycles
// of its texture-address path each, whatever the width): 4 per wave-step for the wave's own rows c, d + 4 per
// workgroup-step for the shared rows a, b = 5 per quartet-step.  Here a wave owns TWO neighbours of the sorted order
// and works through them one after the other inside every 2048-site step: the shared image is fetched once per EIGHT
// quartets (NW = 4 waves) and read from LDS once per two, and when the two quartets have the same third taxon -- the
// rule in lexicographic enumerations (combinations.py:40-55), 60 % of the pairs of a sorted 1e6-of-10.7e6 sample --
// row c is loaded once: 6 (or 8) own loads per wave-step for two quartets, 3.5 - 4.5 loads per quartet-step.
// Everything else (count mask, subsample carry per quartet, pattern build, transposed park, set-bit walk into the
// quartet's own histogram) is the code of the one-quartet form.  METHOD 0 / 1 as there.
// ------------------------------------------------------------------------------------
// MEASURED SLOWER, off by default (option "scan_pair", parity-tested; profiles/r03_scan/pair_kernel_ab.txt): c3 5.85 -> 7.34 ms,
// c2 1.57 -> 1.90 ms.  The state of two quartets costs 109 VGPRs = 4 waves per SIMD (61 = 8 in the one-quartet form; forced to
// 80 registers the compiler spills 53 of them, i.e. more vector memory instructions), so a CU holds the same 32 quartets in
// half as many, twice as long instruction streams: the 22 - 30 % fewer loads do not make up for the latency no longer hidden.
template <bool SUB, int METHOD, int NW>
__global__ void __launch_bounds__(NW *WAVE)
tq_scan_wg2_kernel(DevData d, const uint32_t *__restrict__ quartets, const uint32_t *__restrict__ order, int64_t Q,
                   uint32_t *__restrict__ cm, int64_t xcd_chunk)
{
    static_assert(NW >= 2 && NW <= 8, "waves per workgroup");
    static_assert(METHOD == 0 || METHOD == 1, "histogram method");
    constexpr int QB = 2 * NW;                                  // quartets per block
    constexpr bool PARK_T = METHOD == 1;
    __shared__ uint4 shared_ab[2][SHARED_SLOTS];
    __shared__ uint32_t hist_all[NW][2][256];
    __shared__ __attribute__((aligned(2048))) uint32_t park_all[NW][WAVE * 8];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    uint32_t *hist0 = hist_all[w][0], *hist1 = hist_all[w][1];
    uint8_t *park = reinterpret_cast<uint8_t *>(park_all[w]) + lane * 4;
    for (int i = lane; i < 512; i += WAVE) hist0[i] = 0;
    const uint32_t T = (uint32_t)d.T;
    const int last = d.ntiles - 1;
    const int64_t nblk = (Q + QB - 1) / QB;
    const uint8_t *rows = d.rows;
    const uint8_t *nib = d.nib;
    const uint8_t *planes = reinterpret_cast<const uint8_t *>(d.planes);
    const uint8_t *planes3 = reinterpret_cast<const uint8_t *>(d.planes3);
    const uint32_t pitch = (uint32_t)d.pitch, npitch = pitch / 2, wpitch = (uint32_t)d.W * 16u,
                   w3pitch = (uint32_t)d.W * 12u;
    constexpr int NJOB = 2, JPW = (NJOB + NW - 1) / NW;
    __syncthreads();

    const int64_t nslot = xcd_chunk > 0 ? 8 * xcd_chunk : nblk;
    for (int64_t blk0 = blockIdx.x; blk0 < nslot; blk0 += gridDim.x) {
        int64_t blk = blk0;
        if (xcd_chunk > 0) {
            const int64_t x = blk0 & 7, j = blk0 >> 3;
            blk = x * xcd_chunk + j;
            if (j >= xcd_chunk || blk >= nblk) continue;       // uniform for the whole workgroup
        }
        // leader = first quartet of the block; its (a,b) is what the workgroup shares
        const int64_t it0 = blk * QB;
        const int64_t lqi = order ? (int64_t)order[it0] : it0;
        const uint4 lq = reinterpret_cast<const uint4 *>(quartets)[lqi];
        uint32_t la = __builtin_amdgcn_readfirstlane(lq.x), lb = __builtin_amdgcn_readfirstlane(lq.y);
        const bool leader_ok = (la < T) & (lb < T);
        if (!leader_ok) la = lb = 0;
        // this wave's two quartets (everything about them is wave-uniform)
        bool have[2], work[2], shares[2];
        int64_t qi[2];
        uint32_t qa[2], qb[2], qc[2], qd[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t it = it0 + 2 * w + k;
            have[k] = it < Q;
            qi[k] = have[k] ? (order ? (int64_t)order[it] : it) : 0;
            const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[qi[k]];
            qa[k] = __builtin_amdgcn_readfirstlane(qv.x);
            qb[k] = __builtin_amdgcn_readfirstlane(qv.y);
            qc[k] = __builtin_amdgcn_readfirstlane(qv.z);
            qd[k] = __builtin_amdgcn_readfirstlane(qv.w);
            const bool bad = (qa[k] >= T) | (qb[k] >= T) | (qc[k] >= T) | (qd[k] >= T);
            work[k] = have[k] && !bad;
            shares[k] = work[k] && leader_ok && qa[k] == la && qb[k] == lb;
            if (!work[k]) qc[k] = qd[k] = 0;
        }
        const bool any_work = work[0] | work[1];
        const bool same_c = work[0] && work[1] && qc[0] == qc[1];      // row c of the second quartet = the first one's
        const uint32_t l16 = (uint32_t)lane * 16u, l12 = (uint32_t)lane * 12u;
        const uint32_t oc0 = qc[0] * npitch + l16, od0 = qd[0] * npitch + l16, oc1 = qc[1] * npitch + l16,
                       od1 = qd[1] * npitch + l16;
        const uint32_t pc0 = qc[0] * w3pitch + l12, pd0 = qd[0] * w3pitch + l12, pc1 = qc[1] * w3pitch + l12,
                       pd1 = qd[1] * w3pitch + l12;
        auto job_of = [=](int i) { return w + i * NW; };                       // wave-uniform
        auto fetch_x = [=](int job, int tile) -> uint4 {
            if (job == 0) return ld16(nib, la * npitch + l16 + (uint32_t)tile * (TILE / 2));
            if (job == 1) return ld16(planes, la * wpitch + l16 + (uint32_t)tile * (WAVE * 16));
            return make_uint4(0, 0, 0, 0);
        };
        auto fetch_y = [=](int job, int tile) -> uint4 {
            if (job == 0) return ld16(nib, lb * npitch + l16 + (uint32_t)tile * (TILE / 2));
            if (job == 1) return ld16(planes, lb * wpitch + l16 + (uint32_t)tile * (WAVE * 16));
            return make_uint4(0, 0, 0, 0);
        };
        auto publish = [=](uint4 *buf, int job, uint4 x, uint4 y) {
            if (job == 0) {
                const uint32_t h = 0xF0F0F0F0u;
                const uint32_t s0 = (x.x << 2) + y.x, s1 = (x.y << 2) + y.y, s2 = (x.z << 2) + y.z, s3 = (x.w << 2) + y.w;
                buf[lane] = make_uint4((s0 << 4) & h, s0 & h, (s1 << 4) & h, s1 & h);
                buf[64 + lane] = make_uint4((s2 << 4) & h, s2 & h, (s3 << 4) & h, s3 & h);
            } else if (job == 1) {
                buf[128 + lane] = make_uint4(x.y, x.z, x.x | y.x, (x.y ^ y.y) | (x.z ^ y.z));
                reinterpret_cast<uint32_t *>(buf + 192)[lane] = x.w;
            }
        };
        // own rows of both quartets for one step: c0, d0, d1 and -- unless it is c0 again -- c1
        auto load_mine = [=](OwnRegsS &r0, OwnRegsS &r1_, int tile) {
            if (!any_work) return;
            const uint32_t tn = (uint32_t)tile * (TILE / 2), tp = (uint32_t)tile * (WAVE * 12);
            r0.c = ld16(nib, oc0 + tn);
            r0.d = ld16(nib, od0 + tn);
            r0.pc = ld12(planes3, pc0 + tp);
            r0.pd = ld12(planes3, pd0 + tp);
            r1_.d = ld16(nib, od1 + tn);
            r1_.pd = ld12(planes3, pd1 + tp);
            if (!same_c) {
                r1_.c = ld16(nib, oc1 + tn);
                r1_.pc = ld12(planes3, pc1 + tp);
            }
        };

        uint4 sx[JPW], sy[JPW];
#pragma unroll
        for (int i = 0; i < JPW; ++i) {
            sx[i] = fetch_x(job_of(i), 0);
            sy[i] = fetch_y(job_of(i), 0);
        }
        OwnRegsS A0, A1;
        A1.c = make_uint4(0, 0, 0, 0);
        A1.pc = make_uint4(0, 0, 0, 0);
        load_mine(A0, A1, 0);
#pragma unroll
        for (int i = 0; i < JPW; ++i) publish(shared_ab[0], job_of(i), sx[i], sy[i]);
        uint32_t carry0 = 0, carry1 = 0;
        __syncthreads();

        for (int t = 0; t < d.ntiles; ++t) {
            const int tn = min(t + 1, last);
#pragma unroll
            for (int i = 0; i < JPW; ++i) {
                sx[i] = fetch_x(job_of(i), tn);
                sy[i] = fetch_y(job_of(i), tn);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (any_work) {
                // the shared image of this step: read once for both quartets
                uint4 im0 = make_uint4(0, 0, 0, 0), im1 = im0, imr = im0;
                uint32_t imB = 0;
                if (shares[0] | shares[1]) {
                    const uint4 *buf = shared_ab[t & 1];
                    im0 = buf[lane];
                    im1 = buf[64 + lane];
                    imr = buf[128 + lane];
                    imB = reinterpret_cast<const uint32_t *>(buf + 192)[lane];
                }
                auto one = [&](int k, const uint4 &cc, const uint4 &dd, const uint4 &pcc, const uint4 &pdd, uint32_t &carry,
                               uint32_t *hist, auto &&hook) {
                    uint4 ab0 = im0, ab1 = im1, r1 = imr;
                    uint32_t Bw = imB;
                    if (!shares[k]) {                           // group boundary: private rows a and b
                        const uint32_t o0 = qa[k] * pitch + (uint32_t)t * TILE + l16;
                        const uint32_t o1 = qb[k] * pitch + (uint32_t)t * TILE + l16;
                        const uint4 a0 = ld16(rows, o0), a1 = ld16(rows, o0 + 1024u);
                        const uint4 b0 = ld16(rows, o1), b1 = ld16(rows, o1 + 1024u);
                        ab0 = make_uint4(((a0.x << 2) + b0.x) << 4, ((a0.y << 2) + b0.y) << 4, ((a0.z << 2) + b0.z) << 4,
                                         ((a0.w << 2) + b0.w) << 4);
                        ab1 = make_uint4(((a1.x << 2) + b1.x) << 4, ((a1.y << 2) + b1.y) << 4, ((a1.z << 2) + b1.z) << 4,
                                         ((a1.w << 2) + b1.w) << 4);
                        const uint4 pa = ld16(planes, qa[k] * wpitch + (uint32_t)t * (WAVE * 16) + l16);
                        const uint4 pb = ld16(planes, qb[k] * wpitch + (uint32_t)t * (WAVE * 16) + l16);
                        r1 = make_uint4(pa.y, pa.z, pa.x | pb.x, (pa.y ^ pb.y) | (pa.z ^ pb.z));
                        Bw = pa.w;
                    }
                    const uint32_t C = count_mask_shared<SUB>(r1, Bw, pcc, pdd, lane, carry);
                    uint32_t m;
                    m = sgpr_const_0f();
                    const uint32_t s0 = (cc.x << 2) + dd.x, s1 = (cc.y << 2) + dd.y, s2 = (cc.z << 2) + dd.z,
                                   s3 = (cc.w << 2) + dd.w;
                    uint32_t pat[8];
                    pat[0] = and_or(s0, m, ab0.x); pat[1] = and_or(s0 >> 4, m, ab0.y);
                    pat[2] = and_or(s1, m, ab0.z); pat[3] = and_or(s1 >> 4, m, ab0.w);
                    pat[4] = and_or(s2, m, ab1.x); pat[5] = and_or(s2 >> 4, m, ab1.y);
                    pat[6] = and_or(s3, m, ab1.z); pat[7] = and_or(s3 >> 4, m, ab1.w);
                    hist_patterns<1, METHOD, std::remove_reference_t<decltype(hook)>, PARK_T>(pat, C, hist, park, hook);
                };
                NoHook nohook;
                if (work[0]) one(0, A0.c, A0.d, A0.pc, A0.pd, carry0, hist0, nohook);
                // the rows of step t+1 are requested once the second quartet's patterns are parked: every register
                // they land in is dead by then, and they fly under its walk and the next step's first quartet
                auto hook = [&]() {
                    load_mine(A0, A1, tn);
                    __builtin_amdgcn_sched_barrier(0);
                };
                if (work[1]) {
                    if (same_c) one(1, A0.c, A1.d, A0.pc, A1.pd, carry1, hist1, hook);
                    else one(1, A1.c, A1.d, A1.pc, A1.pd, carry1, hist1, hook);
                } else {
                    hook();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < JPW; ++i) publish(shared_ab[(t + 1) & 1], job_of(i), sx[i], sy[i]);
            __syncthreads();
        }
        // store the 256 counts of both quartets and clear their histograms
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint32_t *hist = k ? hist1 : hist0;
            if (have[k]) {
                uint32_t *out = cm + qi[k] * 256;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int bin = lane + WAVE * j;
                    __builtin_nontemporal_store(work[k] ? hist[bin] : 0u, &out[bin]);
                    hist[bin] = 0;
                }
            }
        }
        __syncthreads();
    }
}
