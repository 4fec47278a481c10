// scan_dp.hpp -- full-mode site scan with a JOINT histogram for two quartets that share three taxa
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace, after scan.hpp).
//
// full_chunk_to_matrices (resolve_quartets.py:76-104) counts every site that is unmasked for the quartet; the
// worker's mask (:216-221) is "a taxon missing, or all four bases equal".  tq_scan_wg_kernel<false, 0> does that with
// one EXEC-masked LDS atomic per site slot and quartet: 32 atomic instructions per 2048-site step whatever the density
// (40 % of the lanes counted on the c3 benchmark), and round 4 measured that the NUMBER of LDS instructions -- 4 cycles
// of operand transfer each, whatever the conflicts -- is what bounds it (profiles/r04_scan/probe_slots.txt).
//
// This kernel halves that number.  Two quartets (a,b,c,d1) and (a,b,c,d2) -- neighbours of the (a,b,c)-sorted order:
// 85 % of a 1e6-of-10.7e6 random sample pair up, 97 % of a lexicographic enumeration (combinations.py:40-55) -- are
// scanned by ONE wave into ONE joint histogram over (a,b,c,d1',d2'), d' in {A,C,G,T,missing}: 64 x 25 bins, one
// atomic per site and PAIR.  At the end the wave folds it into the two 256-bin count tensors,
//     C1[abc][i] = sum_j J[abc][i][j],   C2[abc][j] = sum_i J[abc][i][j]      (i, j over the four bases + "missing"),
// so a site where d2 is missing still counts for the first quartet and vice versa.  Invariant sites need no mask of
// their own: they are exactly the sites with pattern (x,x,x,x), and those four bins are zeroed after the fold (the
// per-site "variable among the present taxa" test below only saves atomics; any superset of the counted sites that
// stays inside "a, b, c present" gives the same result).  A quartet without a partner runs as a unit whose d2 is
// missing everywhere.  Subsample mode cannot use this: its count rule (first unmasked site of a locus) depends on the
// whole quartet, so the two quartets count different sites.
//
// Bin address: LDS byte offset = wave base + E * 256 + (a<<6 | b<<4 | c<<2), E = 5 d1' + d2' (0..24): the low byte
// is the pattern byte of the one-quartet kernel with the d field empty, the high byte is E plus the wave's base / 256
// -- a 16-bit field per site, built two sites per dword by v_perm_b32 from byte lanes; a slot is then
// v_add_co (count mask -> lane mask) + v_bfe (the field IS the address) + ds_add_u32.
#pragma once

#ifndef TQ_DP_PRIO
#define TQ_DP_PRIO 2
#endif
#define TQ_DP_STR_(x) #x
#define TQ_DP_STR(x) TQ_DP_STR_(x)
#if TQ_DP_PRIO
#define TQ_DP_PRIO_ON "s_setprio " TQ_DP_STR(TQ_DP_PRIO)
#define TQ_DP_PRIO_OFF "s_setprio 0"
#else
#define TQ_DP_PRIO_ON ""
#define TQ_DP_PRIO_OFF ""
#endif
constexpr int DP_NW = 4;                    // waves (= pairs) per workgroup
constexpr int DP_ROWS = 25;                 // values of E
constexpr int DP_HIST_DW = DP_ROWS * 64;    // 1600 counters per wave (6400 bytes)
#ifndef TQ_DP_SWZ
#define TQ_DP_SWZ 0
#endif
// what the bank swizzle XORs into the counter index (a<<4|b<<2|c) of row E = 5 d1' + d2'
__host__ __device__ constexpr int dp_swz(int e)
{
    return TQ_DP_SWZ >= 2 ? ((e / 5) ^ (((e / 5 + e % 5) << 2) & 63)) : (TQ_DP_SWZ == 1 ? e / 5 : 0);
}
constexpr uint32_t DP_NONE = 0xFFFFFFFFu;   // units[].y of a quartet without a partner
constexpr int DP_SLOTS = 192;               // uint4 slots of the shared image (scan.hpp's without the run-begin words): with
                                            // 208 a workgroup needs 32 256 bytes of LDS and a CU holds four, with 192 five

// ---- building the unit list from the sorted order ---------------------------------------------------------------
// keys[] = sort keys (a*T+b)*T+c in ascending order, idx[] = the quartets' original indices in that order.  Inside a
// run of equal keys the elements at even distance from the run's first element are paired with their successor.
// flags[i] = 1 when element i starts a unit (first of a pair, or a quartet on its own); an inclusive sum over the
// flags gives the unit's position.  Quartets with a taxon index >= T are never paired (the kernel writes zeros for
// them, as the one-quartet kernel does).
__device__ __forceinline__ bool dp_valid(const uint32_t *__restrict__ quartets, uint32_t qi, uint32_t T)
{
    const uint4 q = reinterpret_cast<const uint4 *>(quartets)[qi];
    return (q.x < T) & (q.y < T) & (q.z < T) & (q.w < T);
}

// role of element i: 0 = second of a pair, 1 = first of a pair, 2 = on its own
__device__ __forceinline__ int dp_role(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ idx,
                                       const uint32_t *__restrict__ quartets, int64_t n, uint32_t T, int64_t i)
{
    const uint32_t k = keys[i];
    // first element of the run: a few steps back (runs are short in a random sample), else the lower bound of k in keys[0..i]
    int64_t lo = i;
    for (int step = 0; step < 8 && lo > 0 && keys[lo - 1] == k; ++step) --lo;
    if (lo > 0 && keys[lo - 1] == k) {
        int64_t hi = lo;
        lo = 0;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < k) lo = mid + 1; else hi = mid;
        }
    }
    const bool odd = ((i - lo) & 1) != 0;
    const bool ok = dp_valid(quartets, idx[i], T);
    if (odd) return (ok && dp_valid(quartets, idx[i - 1], T)) ? 0 : 2;
    const bool next = i + 1 < n && keys[i + 1] == k;
    return (next && ok && dp_valid(quartets, idx[i + 1], T)) ? 1 : 2;
}

__global__ void tq_dp_flag_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ idx,
                                  const uint32_t *__restrict__ quartets, int64_t n, uint32_t T,
                                  uint32_t *__restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flags[i] = dp_role(keys, idx, quartets, n, T, i) != 0 ? 1u : 0u;
}

// pos[] = inclusive sum of the flags.  units[u] = (first quartet, second quartet or DP_NONE); count[0] = units.
__global__ void tq_dp_units_kernel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ idx,
                                   const uint32_t *__restrict__ pos, int64_t n, uint2 *__restrict__ units,
                                   uint32_t *__restrict__ count)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pos[i], before = i ? pos[i - 1] : 0u;
    if (i == n - 1) count[0] = p;
    if (p == before) return;                                   // second of a pair
    (void)keys;
    const bool first = i + 1 < n && pos[i + 1] == p;           // the next element starts no unit: it is my partner
    units[p - 1] = make_uint2(idx[i], first ? idx[i + 1] : DP_NONE);
}

// ---- the scan -------------------------------------------------------------------------------------------------------
#ifndef TQ_DP_OWN_AB
#define TQ_DP_OWN_AB 1       // 1: every wave loads the nibble codes of rows a, b itself (two more vector loads per unit-step, L1 hits
                             // for three of the four waves) and the shared image carries the combined plane record only: the
                             // kernel is bound by LDS instructions (79 % busy against 37 % of the vector peak and 54 % of the
                             // load path), and the pattern-partial panels were two 16-byte image reads per wave-step + two
                             // stores per workgroup-step.  0: the panels come through the image as in tq_scan_wg_kernel (A/B)
#endif
struct DpOwn {
    u32x4 a, b;              // nibble codes of rows a, b (TQ_DP_OWN_AB)
    u32x4 c, d1, d2;         // nibble codes of row c (0..3) and of rows d1, d2 (0..3, 4 = missing: the nib5 copy)
    u32x3 pc, pd1, pd2;      // plane records {miss, p0, p1}
};

template <int NW>
__global__ void __launch_bounds__(NW *WAVE)
tq_scan_dp_kernel(DevData d, const uint32_t *__restrict__ quartets, const uint2 *__restrict__ units,
                  const uint32_t *__restrict__ nunits_dev, uint32_t *__restrict__ cm)
{
    static_assert(NW >= 2 && NW <= 8, "waves per workgroup");
    __shared__ uint4 shared_ab[2][TQ_DP_OWN_AB ? 64 : DP_SLOTS];
    __shared__ __attribute__((aligned(256))) uint32_t hist_all[NW][DP_HIST_DW];
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    uint32_t *hist = hist_all[w];
    for (int i = lane; i < DP_HIST_DW; i += WAVE) hist[i] = 0;
    const uint32_t T = (uint32_t)d.T;
    const int last = d.ntiles - 1;
    const int64_t nunits = (int64_t)nunits_dev[0];
    const int64_t nblk = (nunits + NW - 1) / NW;
    const int64_t xcd_chunk = (nblk + 7) / 8;
    const uint8_t *rows = d.rows;
    const uint8_t *nib = d.nib;
    const uint8_t *nib5 = d.nib5;
    const uint8_t *planes = reinterpret_cast<const uint8_t *>(d.planes);
    const uint8_t *planes3 = reinterpret_cast<const uint8_t *>(d.planes3);
    const uint32_t pitch = (uint32_t)d.pitch, npitch = pitch / 2, wpitch = (uint32_t)d.W * 16u,
                   w3pitch = (uint32_t)d.W * 12u;
    // the wave's histogram starts on a 256-byte boundary: its base goes into the high byte of every site's field
    const uint32_t hbase = __builtin_amdgcn_readfirstlane(lds_offset(hist));
    const uint32_t B4 = (hbase >> 8) * 0x01010101u;
    constexpr int NJOB = TQ_DP_OWN_AB ? 1 : 2, PJOB = NJOB - 1;          // PJOB: the job that fetches the plane records of a, b
    constexpr int R1_SLOT = TQ_DP_OWN_AB ? 0 : 128;
    __syncthreads();

    for (int64_t blk0 = blockIdx.x; blk0 < 8 * xcd_chunk; blk0 += gridDim.x) {
        // workgroup ids are dealt round-robin to the 8 XCDs; the ids of one XCD walk one contiguous eighth of the
        // sorted order (scan.hpp).  gridDim.x is a multiple of 8, so a workgroup's later blocks stay on its XCD.
        const int64_t x = blk0 & 7, jx = blk0 >> 3;
        const int64_t blk = x * xcd_chunk + jx;
        if (blk >= nblk) continue;                             // uniform for the whole workgroup
        const int64_t u0 = blk * NW;
        // leader = first quartet of the block's first unit; its (a,b) is what the workgroup shares
        const uint2 lu = units[u0];
        const uint4 lq = reinterpret_cast<const uint4 *>(quartets)[lu.x];
        uint32_t la = __builtin_amdgcn_readfirstlane(lq.x), lb = __builtin_amdgcn_readfirstlane(lq.y);
        const bool leader_ok = (la < T) & (lb < T);
        if (!leader_ok) la = lb = 0;
        // this wave's unit
        const int64_t u = u0 + w;
        const bool have = u < nunits;
        const uint2 un = units[have ? u : u0];
        const uint32_t qi1 = __builtin_amdgcn_readfirstlane(un.x), qi2 = __builtin_amdgcn_readfirstlane(un.y);
        const bool two = have && qi2 != DP_NONE;                // wave-uniform
        const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[qi1];
        uint32_t q[4];
        q[0] = __builtin_amdgcn_readfirstlane(qv.x);
        q[1] = __builtin_amdgcn_readfirstlane(qv.y);
        q[2] = __builtin_amdgcn_readfirstlane(qv.z);
        q[3] = __builtin_amdgcn_readfirstlane(qv.w);
        uint32_t qd2 = q[3];
        if (two) qd2 = __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint4 *>(quartets)[qi2].w);
        const bool bad = (q[0] >= T) | (q[1] >= T) | (q[2] >= T) | (q[3] >= T) | (qd2 >= T);
        const bool work = have && !bad;                         // (the unit list never pairs an invalid quartet)
        const bool shares = work && leader_ok && q[0] == la && q[1] == lb;
        const uint32_t qc = work ? q[2] : 0, qd1 = work ? q[3] : 0, qdd2 = work ? qd2 : 0;
        // a unit of one quartet: d2 reads d1's row again and is declared missing everywhere
        const uint32_t force_miss = two ? 0u : 0xFFFFFFFFu, force_code = two ? 0u : 0x44444444u, keep2 = ~force_miss;
        const uint32_t l16 = (uint32_t)lane * 16u, l12 = (uint32_t)lane * 12u;
        const uint32_t oc = qc * npitch + l16, od1 = qd1 * npitch + l16, od2 = qdd2 * npitch + l16;
        const uint32_t oa = (work ? q[0] : 0) * npitch + l16, ob = (work ? q[1] : 0) * npitch + l16;
        const uint32_t opc = qc * w3pitch + l12, opd1 = qd1 * w3pitch + l12, opd2 = qdd2 * w3pitch + l12;
        auto fetch_x = [=](int job, int tile) -> uint4 {
            if (!TQ_DP_OWN_AB && job == 0) return ld16(nib, la * npitch + l16 + (uint32_t)tile * (TILE / 2));
            if (job == PJOB) return ld16(planes, la * wpitch + l16 + (uint32_t)tile * (WAVE * 16));
            return make_uint4(0, 0, 0, 0);
        };
        auto fetch_y = [=](int job, int tile) -> uint4 {
            if (!TQ_DP_OWN_AB && job == 0) return ld16(nib, lb * npitch + l16 + (uint32_t)tile * (TILE / 2));
            if (job == PJOB) return ld16(planes, lb * wpitch + l16 + (uint32_t)tile * (WAVE * 16));
            return make_uint4(0, 0, 0, 0);
        };
        // image of one step (scan.hpp): panels 0-63 / 64-127 = ((a<<2)+b)<<4 per site byte, 128-191 = {p0a, p1a, Ma|Mb,
        // (p0a^p0b)|(p1a^p1b)}
        auto publish = [=](uint4 *buf, int job, uint4 x, uint4 y) {
            if (!TQ_DP_OWN_AB && job == 0) {
                const uint32_t h = 0xF0F0F0F0u;
                const uint32_t s0 = (x.x << 2) + y.x, s1 = (x.y << 2) + y.y, s2 = (x.z << 2) + y.z, s3 = (x.w << 2) + y.w;
                buf[lane] = make_uint4((s0 << 4) & h, s0 & h, (s1 << 4) & h, s1 & h);
                buf[64 + lane] = make_uint4((s2 << 4) & h, s2 & h, (s3 << 4) & h, s3 & h);
            } else if (job == PJOB) {
                buf[R1_SLOT + lane] = make_uint4(x.y, x.z, x.x | y.x, (x.y ^ y.y) | (x.z ^ y.z));
            }
        };
        auto load_mine = [=](DpOwn &r, int tile) {
            const uint32_t tn = (uint32_t)tile * (TILE / 2), tp = (uint32_t)tile * (WAVE * 12);
            if (TQ_DP_OWN_AB) {
                r.a = ldv16(nib, oa + tn);
                r.b = ldv16(nib, ob + tn);
            }
            r.c = ldv16(nib, oc + tn);
            r.d1 = ldv16(nib5, od1 + tn);
            r.d2 = ldv16(nib5, od2 + tn);
            r.pc = ldv12(planes3, opc + tp);
            r.pd1 = ldv12(planes3, opd1 + tp);
            r.pd2 = ldv12(planes3, opd2 + tp);
        };

        // loop copies by the wave's job (0, 1, none) for waves that share the leader's rows, one generic copy for the
        // rest: inside a copy every load is unconditional and the compiler's s_waitcnt counts are exact (scan.hpp)
        auto run = [&](auto spec_tag, auto fast_tag) {
            constexpr int SPEC = decltype(spec_tag)::value;
            constexpr bool FAST = decltype(fast_tag)::value;
            const int job = SPEC >= 0 ? SPEC : w;               // waves >= NJOB have no job
            uint4 sx = fetch_x(job, 0), sy = fetch_y(job, 0);
            DpOwn A;
            load_mine(A, 0);
            publish(shared_ab[0], job, sx, sy);
            __syncthreads();

            auto step = [&](DpOwn &own, int t, int tnext) {
                if (!(FAST || work)) return;
                uint4 ab0, ab1, r1;
                if (FAST || shares) {
                    const uint4 *buf = shared_ab[t & 1];
                    if (!TQ_DP_OWN_AB) {
                        ab0 = buf[lane];
                        ab1 = buf[64 + lane];
                    }
                    r1 = buf[R1_SLOT + lane];
                } else {                                        // group boundary: private rows a and b
                    if (!TQ_DP_OWN_AB) {
                        const uint32_t o0 = q[0] * pitch + (uint32_t)t * TILE + l16;
                        const uint32_t o1 = q[1] * pitch + (uint32_t)t * TILE + l16;
                        const uint4 a0 = ld16(rows, o0), a1 = ld16(rows, o0 + 1024u);
                        const uint4 b0 = ld16(rows, o1), b1 = ld16(rows, o1 + 1024u);
                        ab0 = make_uint4(((a0.x << 2) + b0.x) << 4, ((a0.y << 2) + b0.y) << 4, ((a0.z << 2) + b0.z) << 4,
                                         ((a0.w << 2) + b0.w) << 4);
                        ab1 = make_uint4(((a1.x << 2) + b1.x) << 4, ((a1.y << 2) + b1.y) << 4, ((a1.z << 2) + b1.z) << 4,
                                         ((a1.w << 2) + b1.w) << 4);
                    }
                    const uint4 pa = ld16(planes, q[0] * wpitch + (uint32_t)t * (WAVE * 16) + l16);
                    const uint4 pb = ld16(planes, q[1] * wpitch + (uint32_t)t * (WAVE * 16) + l16);
                    r1 = make_uint4(pa.y, pa.z, pa.x | pb.x, (pa.y ^ pb.y) | (pa.z ^ pb.z));
                }
                if (TQ_DP_OWN_AB) {                             // ((a<<2)+b)<<4 per site byte from the wave's own nibble words
                    const uint32_t h = 0xF0F0F0F0u;
                    const uint32_t s0 = (own.a.x << 2) + own.b.x, s1 = (own.a.y << 2) + own.b.y, s2 = (own.a.z << 2) + own.b.z,
                                   s3 = (own.a.w << 2) + own.b.w;
                    ab0 = make_uint4((s0 << 4) & h, s0 & h, (s1 << 4) & h, s1 & h);
                    ab1 = make_uint4((s2 << 4) & h, s2 & h, (s3 << 4) & h, s3 & h);
                }
                // sites that get an atomic: a, b, c present and, for at least one of the two quartets, d present and the
                // four bases not all equal (the second condition only saves atomics: invariant bins are zeroed at the end)
                const uint32_t Mabc = r1.z | own.pc.x;
                const uint32_t Vabc = r1.w | (r1.x ^ own.pc.y) | (r1.y ^ own.pc.z);
                const uint32_t m1 = own.pd1.x, m2 = own.pd2.x | force_miss;
                const uint32_t V1 = Vabc | (r1.x ^ own.pd1.y) | (r1.y ^ own.pd1.z);
                const uint32_t V2 = Vabc | (r1.x ^ own.pd2.y) | (r1.y ^ own.pd2.z);
                const uint32_t C = ~Mabc & ((V1 & ~m1) | (V2 & ~m2));
                // fields: low byte (a<<6|b<<4|c<<2), high byte E + base/256, E = 5 d1' + d2' = 4 d1' + (d1' + d2')
                const uint32_t m0f = sgpr_const_0f();
                const uint32_t m3c = 0x3C3C3C3Cu;
                uint32_t W[16];
                auto build8 = [&](int j, uint32_t cw, uint32_t x, uint32_t y, uint32_t abl, uint32_t abh) {
                    uint32_t hb03 = ((cw << 2) & m3c) | abl;                   // sites 8j .. 8j+3
                    uint32_t hb47 = ((cw >> 2) & m3c) | abh;                   // sites 8j+4 .. 8j+7
                    const uint32_t yy = (y & keep2) | force_code;              // (4 in every nibble for a unit of one quartet)
                    const uint32_t s = x + yy;                                 // d1' + d2' per nibble (<= 8)
                    const uint32_t t03 = (x << 2) & m3c, t47 = (x >> 2) & m3c;  // 4 d1' per byte
                    const uint32_t u03 = s & m0f, u47 = (s >> 4) & m0f;        // d1' + d2' per byte
                    const uint32_t e03 = t03 + u03 + B4;
                    const uint32_t e47 = t47 + u47 + B4;
                    // bank swizzle: the counter's bank is (a,b,c) mod 32 as it stands, and (a,b,c) is the most skewed part of
                    // the pattern; d1' (and d1' + d2') are XOR-ed into it -- dp_swz() below is the same function of the row E
                    if (TQ_DP_SWZ >= 1) { hb03 ^= t03; hb47 ^= t47; }
                    if (TQ_DP_SWZ >= 2) { hb03 ^= u03 << 4; hb47 ^= u47 << 4; }
                    W[4 * j + 0] = __builtin_amdgcn_perm(e03, hb03, 0x05010400u);   // sites 8j, 8j+1
                    W[4 * j + 1] = __builtin_amdgcn_perm(e03, hb03, 0x07030602u);   // sites 8j+2, 8j+3
                    W[4 * j + 2] = __builtin_amdgcn_perm(e47, hb47, 0x05010400u);
                    W[4 * j + 3] = __builtin_amdgcn_perm(e47, hb47, 0x07030602u);
                };
                build8(0, own.c.x, own.d1.x, own.d2.x, ab0.x, ab0.y);
                build8(1, own.c.y, own.d1.y, own.d2.y, ab0.z, ab0.w);
                build8(2, own.c.z, own.d1.z, own.d2.z, ab1.x, ab1.y);
                build8(3, own.c.w, own.d1.w, own.d2.w, ab1.z, ab1.w);
                // the rows of step t+1 are requested now: every register they land in is dead, and they fly under the slots
                __builtin_amdgcn_sched_barrier(0);
                load_mine(own, tnext);
                __builtin_amdgcn_sched_barrier(0);
#ifndef TQ_NO_ASM
                {
                    uint32_t c = C, a, one = 1u;
                    uint64_t save;
                    asm volatile("s_mov_b64 %0, exec\n\t" TQ_DP_PRIO_ON : "=s"(save));
#define TQ_DP_SLOT(J, K)                                                                                        \
                    asm volatile("v_add_co_u32_e32 %[c], vcc, %[c], %[c]\n\t"                                     \
                                 "v_bfe_u32 %[a], %[p], " #K "*16, 16\n\t"                                         \
                                 "s_and_b64 exec, %[save], vcc\n\t"                                                \
                                 "ds_add_u32 %[a], %[one]\n\t"                                                     \
                                 "s_mov_b64 exec, %[save]"                                                          \
                                 : [c] "+v"(c), [a] "=&v"(a)                                                        \
                                 : [p] "v"(W[J]), [one] "v"(one), [save] "s"(save)                                  \
                                 : "vcc", "memory");
#define TQ_DP_SLOT2(J) TQ_DP_SLOT(J, 1) TQ_DP_SLOT(J, 0)
                    TQ_DP_SLOT2(15) TQ_DP_SLOT2(14) TQ_DP_SLOT2(13) TQ_DP_SLOT2(12) TQ_DP_SLOT2(11) TQ_DP_SLOT2(10)
                    TQ_DP_SLOT2(9) TQ_DP_SLOT2(8) TQ_DP_SLOT2(7) TQ_DP_SLOT2(6) TQ_DP_SLOT2(5) TQ_DP_SLOT2(4)
                    TQ_DP_SLOT2(3) TQ_DP_SLOT2(2) TQ_DP_SLOT2(1) TQ_DP_SLOT2(0)
#undef TQ_DP_SLOT2
#undef TQ_DP_SLOT
                    asm volatile(TQ_DP_PRIO_OFF ::: "memory");
                }
#else
                {
                    uint32_t *lds0 = hist - hbase / 4;          // LDS offset 0 as a pointer into this address space
#pragma unroll
                    for (int i = 0; i < 32; ++i)
                        if (C & (1u << i))
                            __hip_atomic_fetch_add(lds0 + ((W[i >> 1] >> (16 * (i & 1))) & 0xFFFFu) / 4, 1u, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#endif
            };

            for (int t = 0; t < d.ntiles; ++t) {
                const int tn = min(t + 1, last);
                sx = fetch_x(job, tn);
                sy = fetch_y(job, tn);
                __builtin_amdgcn_sched_barrier(0);
                step(A, t, tn);
                __builtin_amdgcn_sched_barrier(0);
                if (job < NJOB) {
                    pin4(sx);
                    pin4(sy);
                }
                publish(shared_ab[(t + 1) & 1], job, sx, sy);
                __syncthreads();
            }
        };
        {
            using std::integral_constant;
            if (shares) {
                if (w == 0) run(integral_constant<int, 0>{}, integral_constant<bool, true>{});
                else if (w == 1) run(integral_constant<int, 1>{}, integral_constant<bool, true>{});
                else run(integral_constant<int, NJOB>{}, integral_constant<bool, true>{});
            } else {
                run(integral_constant<int, -1>{}, integral_constant<bool, false>{});
            }
        }
        // fold: lane = (a<<4|b<<2|c) reads its 25 counters, clears them, and writes four counts of each quartet
        if (have) {
            uint32_t v[DP_ROWS];
#pragma unroll
            for (int e = 0; e < DP_ROWS; ++e) {
                v[e] = hist[e * 64 + (lane ^ dp_swz(e))];
                hist[e * 64 + (lane ^ dp_swz(e))] = 0;
            }
            uint32_t c1[4], c2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c1[i] = v[5 * i] + v[5 * i + 1] + v[5 * i + 2] + v[5 * i + 3] + v[5 * i + 4];     // d1 = i, d2 anything
                c2[i] = v[i] + v[5 + i] + v[10 + i] + v[15 + i] + v[20 + i];                       // d2 = i, d1 anything
            }
            // invariant sites (all four bases equal) are masked by the worker (resolve_quartets.py:218)
            const uint32_t xa = (uint32_t)lane >> 4, xb = ((uint32_t)lane >> 2) & 3u, xc = (uint32_t)lane & 3u;
            const bool eq3 = xa == xb && xb == xc;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!work || (eq3 && xc == (uint32_t)i)) {
                    c1[i] = 0;
                    c2[i] = 0;
                }
            }
            u32x4 o1 = {c1[0], c1[1], c1[2], c1[3]}, o2 = {c2[0], c2[1], c2[2], c2[3]};
            __builtin_nontemporal_store(o1, reinterpret_cast<u32x4 *>(cm + (int64_t)qi1 * 256) + lane);
            if (two) __builtin_nontemporal_store(o2, reinterpret_cast<u32x4 *>(cm + (int64_t)qi2 * 256) + lane);
        }
        __syncthreads();
    }
}
