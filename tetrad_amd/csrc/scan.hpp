// scan.hpp -- site scan -> 256-bin pattern histogram (per-wave and workgroup-cooperative kernels)
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
#pragma once

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
template <bool SUB>
__device__ __forceinline__ uint32_t count_mask(const TileRegs &r, int lane, uint32_t &tile_carry)
{
    const uint32_t M = r.pa.x | r.pb.x | r.pc.x | r.pd.x;
    const uint32_t V = (r.pa.y ^ r.pb.y) | (r.pa.z ^ r.pb.z) | (r.pa.y ^ r.pc.y) | (r.pa.z ^ r.pc.z) |
                       (r.pa.y ^ r.pd.y) | (r.pa.z ^ r.pd.z);
    const uint32_t U = V & ~M;
    if (!SUB) return U;
    const uint32_t B = r.pa.w;
    const uint32_t P = ~B;
    const uint32_t X = U | P;
    const uint32_t sum = X + U;
    const uint32_t cin0 = sum ^ X ^ U;                       // carry into each bit, lane carry-in = 0
    const uint32_t seen_local = P & cin0;
    const uint32_t gen = ((X & U) | ((X | U) & ~sum)) >> 31; // carry out of bit 31 = t(31)
    // cross-lane: T(l) = gen(l) | (allprop(l) & T(l-1)); same adder trick on 64-bit ballots (SALU)
    const uint64_t Gm = __ballot(gen != 0);
    const uint64_t Pm = __ballot(B == 0);
    const uint64_t Xm = Gm | Pm;
    const uint64_t s1 = Xm + Gm;
    const uint64_t s2 = s1 + (uint64_t)tile_carry;
    const uint64_t cinm = s2 ^ Xm ^ Gm;                      // carry into each lane
    tile_carry = (uint32_t)((s1 < Xm) | (s2 < s1));          // carry out of lane 63
    const uint32_t cin = (uint32_t)(cinm >> lane) & 1u;
    // sites before this lane's first run-begin inherit the incoming "seen" state
    const uint32_t firstseg = B ? ((B & (0u - B)) - 1u) : 0xFFFFFFFFu;
    const uint32_t seen = seen_local | (cin ? firstseg : 0u);
    return U & ~seen;
}

// four sites (one dword of each row): EXEC-masked histogram increments for the counted ones
template <int NREP>
__device__ __forceinline__ void hist_dword(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t C, int site0,
                                           uint32_t *hrep)
{
    // base codes are 0..3, so the per-byte pattern (a<<6|b<<4|c<<2|d) never crosses a byte
    const uint32_t pat = (((((a << 2) + b) << 2) + c) << 2) + d;     // fields never overlap: + == |, one v_lshl_add each
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (C & (1u << (site0 + k)))
            __hip_atomic_fetch_add(&hrep[((pat >> (8 * k)) & 0xFFu) * NREP], 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// METHOD 0: one EXEC-masked ds_add per site slot (32 per step, whatever the density).
// METHOD 1: the lane parks its 32 pattern bytes in LDS and walks the set bits of C: the number of
//           ds_add per step is the largest per-lane count in the wave (~10 of 32 in subsample
//           mode, where at most one site per locus run is counted).
constexpr int PAT_STRIDE = 36;   // bytes per lane in the pattern park (9 dwords: conflict-free b32 stores)

template <int NREP, bool SUB, int METHOD>
__device__ __forceinline__ void process_tile(const TileRegs &t, int lane, uint32_t &tile_carry, uint32_t *hrep,
                                             uint8_t *park)
{
    const uint32_t C = count_mask<SUB>(t, lane, tile_carry);
    if (METHOD == 2) {          // timing diagnostic: everything but the histogram (results are wrong)
#define TQ_PAT(a, b, c, d) (((((((a) << 2) + (b)) << 2) + (c)) << 2) + (d))
        uint32_t acc = C;
        acc ^= TQ_PAT(t.a0.x, t.b0.x, t.c0.x, t.d0.x) ^ TQ_PAT(t.a0.y, t.b0.y, t.c0.y, t.d0.y);
        acc ^= TQ_PAT(t.a0.z, t.b0.z, t.c0.z, t.d0.z) ^ TQ_PAT(t.a0.w, t.b0.w, t.c0.w, t.d0.w);
        acc ^= TQ_PAT(t.a1.x, t.b1.x, t.c1.x, t.d1.x) ^ TQ_PAT(t.a1.y, t.b1.y, t.c1.y, t.d1.y);
        acc ^= TQ_PAT(t.a1.z, t.b1.z, t.c1.z, t.d1.z) ^ TQ_PAT(t.a1.w, t.b1.w, t.c1.w, t.d1.w);
#undef TQ_PAT
        asm volatile("" ::"v"(acc));
    } else if (METHOD == 0) {
        hist_dword<NREP>(t.a0.x, t.b0.x, t.c0.x, t.d0.x, C, 0, hrep);
        hist_dword<NREP>(t.a0.y, t.b0.y, t.c0.y, t.d0.y, C, 4, hrep);
        hist_dword<NREP>(t.a0.z, t.b0.z, t.c0.z, t.d0.z, C, 8, hrep);
        hist_dword<NREP>(t.a0.w, t.b0.w, t.c0.w, t.d0.w, C, 12, hrep);
        hist_dword<NREP>(t.a1.x, t.b1.x, t.c1.x, t.d1.x, C, 16, hrep);
        hist_dword<NREP>(t.a1.y, t.b1.y, t.c1.y, t.d1.y, C, 20, hrep);
        hist_dword<NREP>(t.a1.z, t.b1.z, t.c1.z, t.d1.z, C, 24, hrep);
        hist_dword<NREP>(t.a1.w, t.b1.w, t.c1.w, t.d1.w, C, 28, hrep);
    } else {
        uint32_t *pw = reinterpret_cast<uint32_t *>(park);
#define TQ_PAT(a, b, c, d) (((((((a) << 2) + (b)) << 2) + (c)) << 2) + (d))
        pw[0] = TQ_PAT(t.a0.x, t.b0.x, t.c0.x, t.d0.x);
        pw[1] = TQ_PAT(t.a0.y, t.b0.y, t.c0.y, t.d0.y);
        pw[2] = TQ_PAT(t.a0.z, t.b0.z, t.c0.z, t.d0.z);
        pw[3] = TQ_PAT(t.a0.w, t.b0.w, t.c0.w, t.d0.w);
        pw[4] = TQ_PAT(t.a1.x, t.b1.x, t.c1.x, t.d1.x);
        pw[5] = TQ_PAT(t.a1.y, t.b1.y, t.c1.y, t.d1.y);
        pw[6] = TQ_PAT(t.a1.z, t.b1.z, t.c1.z, t.d1.z);
        pw[7] = TQ_PAT(t.a1.w, t.b1.w, t.c1.w, t.d1.w);
#undef TQ_PAT
        // set-bit walk, software pipelined: the pattern byte of the NEXT counted site is requested
        // before the histogram increment of the current one, so the LDS read latency of one step
        // hides behind the previous step instead of stalling every iteration
        uint32_t c = C;
        if (c) {
            // two alternating registers instead of a copy: a copy would wait for the read it copies
            uint32_t b0 = park[__builtin_ctz(c)], b1 = 0;
            c &= c - 1;
            for (;;) {
                if (!c) {
                    __hip_atomic_fetch_add(&hrep[b0 * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                b1 = park[__builtin_ctz(c)];
                c &= c - 1;
                __hip_atomic_fetch_add(&hrep[b0 * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!c) {
                    __hip_atomic_fetch_add(&hrep[b1 * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                b0 = park[__builtin_ctz(c)];
                c &= c - 1;
                __hip_atomic_fetch_add(&hrep[b1 * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
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
        process_tile<NREP, SUB, METHOD>(A, lane, tile_carry, hrep, park);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 >= d.ntiles) break;
        load_tile(A, d, q, min(t + 2, last), lane);
        __builtin_amdgcn_sched_barrier(0);
        process_tile<NREP, SUB, METHOD>(B, lane, tile_carry, hrep, park);
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
// kernel 1, workgroup-cooperative form: NW wavefronts = NW neighbours of the (a,b)-sorted order.
// Quartets that share their first two taxa share two of their four rows, so the workgroup fetches
// the rows (and plane records) of taxa a and b ONCE per 2048-site step into LDS (all NW*64 threads
// cooperate: 384 x 16 B), double-buffered with one barrier per step; every wave still streams its
// own rows c and d straight to registers.  Cache traffic per quartet falls from 12 KiB to
// 6 + 6/NW KiB per step -- the scan kernel is L2 / Infinity-Cache bandwidth bound.  A wave whose
// (a,b) differs from the leader's (group boundary in the sorted order) reads its own a and b
// from global memory instead; it still takes part in the loads and barriers.
// ------------------------------------------------------------------------------------
struct OwnRegs {
    uint4 c0, c1, d0, d1, pc, pd;
};

// 16-byte load at a wave-uniform base + 32-bit per-lane byte offset (lets the compiler use the
// SGPR-base addressing form instead of 64-bit VGPR pointer arithmetic for every load; the host
// guarantees T*Sp < 2^32 before it selects this kernel)
__device__ __forceinline__ uint4 ld16(const uint8_t *base, uint32_t off)
{
    return *reinterpret_cast<const uint4 *>(base + off);
}

// per-lane byte offsets of a wave's own rows c, d (rows array) and their plane records
struct OwnOff {
    uint32_t c, d, pc, pd;
};

__device__ __forceinline__ void load_own(OwnRegs &r, const uint8_t *rows, const uint8_t *planes, const OwnOff &o,
                                         int tile)
{
    const uint32_t tb = (uint32_t)tile * TILE, tp = (uint32_t)tile * (WAVE * 16);
    r.c0 = ld16(rows, o.c + tb);
    r.c1 = ld16(rows, o.c + tb + 1024);
    r.d0 = ld16(rows, o.d + tb);
    r.d1 = ld16(rows, o.d + tb + 1024);
    r.pc = ld16(planes, o.pc + tp);
    r.pd = ld16(planes, o.pd + tp);
}

constexpr int SHARED_PIECES = 384;   // uint4 per step: row a 128, row b 128, planes a 64, planes b 64

template <bool SUB, int METHOD, int NW>
__global__ void __launch_bounds__(NW *WAVE)
tq_scan_wg_kernel(DevData d, const uint32_t *__restrict__ quartets, const uint32_t *__restrict__ order, int64_t Q,
                  uint32_t *__restrict__ cm)
{
    static_assert(NW * WAVE >= SHARED_PIECES, "one cooperative piece per thread");
    __shared__ uint4 shared_ab[2][SHARED_PIECES];
    __shared__ uint32_t hist_all[NW][256];
    __shared__ uint32_t park_all[NW][WAVE * PAT_STRIDE / 4];
    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int lane = tid & 63;
    uint32_t *hist = hist_all[w];
    uint8_t *park = reinterpret_cast<uint8_t *>(park_all[w]) + lane * PAT_STRIDE;
    for (int i = lane; i < 256; i += WAVE) hist[i] = 0;
    const uint32_t T = (uint32_t)d.T;
    const int last = d.ntiles - 1;
    const int64_t nblk = (Q + NW - 1) / NW;
    const uint8_t *rows = d.rows;
    const uint8_t *planes = reinterpret_cast<const uint8_t *>(d.planes);
    const uint32_t pitch = (uint32_t)d.pitch, wpitch = (uint32_t)d.W * 16u;
    // this thread's cooperative piece: waves 0-3 fetch row bytes of a / b, waves 4-5 plane records
    const uint8_t *sh_base = tid < 256 ? rows : planes;                       // wave-uniform
    const uint32_t sh_step = tid < 256 ? (uint32_t)TILE : (uint32_t)(WAVE * 16);
    const uint32_t sh_lane = tid < 256 ? (uint32_t)(tid & 127) * 16u : (uint32_t)((tid - 256) & 63) * 16u;
    const bool sh_is_b = tid < 256 ? (tid >= 128) : (tid >= 320);
    __syncthreads();

    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        // leader = first quartet of the block; its (a,b) is what the workgroup shares
        const int64_t it0 = blk * NW;
        const int64_t lqi = order ? (int64_t)order[it0] : it0;
        const uint4 lq = reinterpret_cast<const uint4 *>(quartets)[lqi];
        uint32_t la = __builtin_amdgcn_readfirstlane(lq.x), lb = __builtin_amdgcn_readfirstlane(lq.y);
        const bool leader_ok = (la < T) & (lb < T);
        if (!leader_ok) la = lb = 0;
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
        oo.c = qc * pitch + (uint32_t)lane * 16u;
        oo.d = qd * pitch + (uint32_t)lane * 16u;
        oo.pc = qc * wpitch + (uint32_t)lane * 16u;
        oo.pd = qd * wpitch + (uint32_t)lane * 16u;
        const uint32_t sh_off = (sh_is_b ? lb : la) * (tid < 256 ? pitch : wpitch) + sh_lane;

        // prologue: step 0 into buffer 0
        const int slot = tid;     // rows are stored in panels already (row_offset), so piece p is slot p
        if (tid < SHARED_PIECES) shared_ab[0][slot] = ld16(sh_base, sh_off);
        OwnRegs A, B;
        load_own(A, rows, planes, oo, 0);
        uint32_t tile_carry = 0;
        __syncthreads();

        auto step = [&](const OwnRegs &own, int t) {
            TileRegs r;
            if (shares) {
                const uint4 *buf = shared_ab[t & 1];
                r.a0 = buf[lane];
                r.a1 = buf[64 + lane];
                r.b0 = buf[128 + lane];
                r.b1 = buf[192 + lane];
                r.pa = buf[256 + lane];
                r.pb = buf[320 + lane];
            } else if (work) {                               // group boundary: private rows a and b
                const int64_t boff = (int64_t)t * TILE + lane * 16;
                const uint4 *pa = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[0] * d.pitch + boff);
                const uint4 *pb = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[1] * d.pitch + boff);
                r.a0 = pa[0]; r.a1 = pa[64];
                r.b0 = pb[0]; r.b1 = pb[64];
                const int64_t woff = (int64_t)t * WAVE + lane;
                r.pa = d.planes[(int64_t)q[0] * d.W + woff];
                r.pb = d.planes[(int64_t)q[1] * d.W + woff];
            }
            if (work) {
                r.c0 = own.c0; r.c1 = own.c1;
                r.d0 = own.d0; r.d1 = own.d1;
                r.pc = own.pc; r.pd = own.pd;
                process_tile<1, SUB, METHOD>(r, lane, tile_carry, hist, park);
            }
        };

        for (int t = 0; t < d.ntiles; t += 2) {
            // ---- even step: prefetch t+1 (own -> B, shared -> registers), process A ----
            {
                const int tn = min(t + 1, last);
                uint4 sh = make_uint4(0, 0, 0, 0);
                if (tid < SHARED_PIECES) sh = ld16(sh_base, sh_off + (uint32_t)tn * sh_step);
                load_own(B, rows, planes, oo, tn);
                __builtin_amdgcn_sched_barrier(0);
                step(A, t);
                __builtin_amdgcn_sched_barrier(0);
                if (tid < SHARED_PIECES) shared_ab[(t + 1) & 1][slot] = sh;
                __syncthreads();
            }
            if (t + 1 >= d.ntiles) break;
            // ---- odd step ----
            {
                const int tn = min(t + 2, last);
                uint4 sh = make_uint4(0, 0, 0, 0);
                if (tid < SHARED_PIECES) sh = ld16(sh_base, sh_off + (uint32_t)tn * sh_step);
                load_own(A, rows, planes, oo, tn);
                __builtin_amdgcn_sched_barrier(0);
                step(B, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (tid < SHARED_PIECES) shared_ab[t & 1][slot] = sh;
                __syncthreads();
            }
        }
        // store the 256 counts of this wave's quartet and clear its histogram
        if (have) {
            uint32_t *out = cm + qi * 256;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int bin = lane + WAVE * k;
                out[bin] = work ? hist[bin] : 0u;
                hist[bin] = 0;
            }
        }
        __syncthreads();
    }
}

