// tetrad_hip.hip -- MI355X (gfx950 / CDNA4) quartet-invariant engine.
//
// Hand-written HIP for the per-quartet hot path of eaton-lab/tetrad
// (reference: tetrad/src/resolve_quartets.py:191-265 and the count kernels
// :42-104).  Not a translation: the reference is an interpreted per-quartet
// loop around a serial site scan and six LAPACK calls; here one 64-lane
// wavefront owns a quartet during the site scan and one 16-lane group owns a
// quartet during the singular-value stage.
//
// Data layout in HBM (built once per replicate by tq_set_data):
//   rows   u8   [T][Sp]  base code 0..3 per site, missing/pad -> 0    (Sp = S rounded up to 2048;
//                        inside each 2048-site step the bytes are stored in two 1 KiB panels, row_offset())
//   planes u32x4[T][W]   per 32 sites: {missing bits, base bit 0, base bit 1, run-begin bits}
//                        (W = Sp/32; run-begin = site starts a new locus run, same for every row,
//                        replicated so that one 16-byte load brings everything a lane needs)
//
// Kernel 1, tq_scan_kernel: one wavefront per quartet, persistent grid.  The wave scans 2048
//   sites per step; lane l owns 32 consecutive sites: 2 x dwordx4 of each of the 4 rows
//   (coalesced 2 KiB per row per step) + one 16-byte plane record per row, prefetched one step
//   ahead.  Which sites count is pure bit logic on 32-site words: U = variable-among-the-4 &
//   ~missing (resolve_quartets.py:216-218); in subsample mode additionally "first unmasked site
//   of its locus run" (:58-64) via an adder carry chain, cross-lane carries resolved on the
//   scalar unit from two ballots.  The 8-bit site pattern (a<<6|b<<4|c<<2|d) is built 4 sites
//   per VALU op (SWAR) and counted with EXEC-masked ds_add_u32 into an LDS histogram with NREP
//   lane-interleaved replicas; the folded 256 counts go to a scratch slab cm[Q][256] (1 KiB per
//   quartet, L2/MALL resident between the two kernels).
// Kernel 2, tq_svd_kernel: each 16-lane group takes one quartet; lane j holds column j (16 f64)
//   of a flattening and the group runs a one-sided (Hestenes) Jacobi SVD with the XOR-partner
//   parallel ordering (15 rounds per sweep, partner = lane ^ m, exchanged with DPP moves).  Rank
//   rule, minrank, tail-norm scores and argmin follow resolve_quartets.py:241-251.
//
// Bounds: the scan is L2 / LDS-atomic / VALU work on a <= 40 MB resident matrix (HBM only on
// first touch), the SVD is f64 VALU.  Algorithmic bytes per quartet: 4*S + 48.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/tetrad_hip.h"

namespace {

constexpr int WAVE = 64;
constexpr int SITES_PER_LANE = 32;
constexpr int TILE = WAVE * SITES_PER_LANE;        // 2048 sites per wave step
constexpr int QPW = 4;                              // quartets per wave pass of the SVD kernel
constexpr double F64_EPS = 2.220446049250313e-16;
constexpr double JTOL2 = 7.888609052210118e-31;     // (2^-50)^2 : rotate while g^2 > JTOL2*a*b
constexpr double JEARLY2 = 1e-10;                   // (1e-5)^2 : see jacobi16
constexpr int MAX_SWEEPS = 30;
constexpr double DEGENERATE_REL_GAP = 1e-9;

struct DevData {
    const uint8_t *rows;
    const uint4 *planes;
    int64_t pitch;      // bytes per row (Sp)
    int64_t W;          // plane records per row (Sp/32)
    int32_t T;
    int32_t ntiles;     // Sp / TILE
};

// Byte offset of site s inside a row.  A 2048-site step is stored as two 1 KiB panels: panel 0
// holds sites 0-15 of every lane's 32-site group, panel 1 holds sites 16-31, so each of the two
// 16-byte loads a lane issues per row is part of one fully contiguous 1 KiB wave access.
__host__ __device__ __forceinline__ int64_t row_offset(int64_t s)
{
    const int64_t tile = s >> 11, r = s & 2047, lane = r >> 5, k = r & 31;
    return (tile << 11) + ((k >> 4) << 10) + (lane << 4) + (k & 15);
}

struct OutPtrs {
    uint32_t *rstat;    // [Q,2]
    double *rscor;      // [Q,3]
    uint8_t *flags;     // [Q] or null
    uint32_t *cmats;    // [Q,3,16,16] or null (debug)
    double *svds;       // [Q,3,16] or null (debug)
    int32_t *ranks;     // [Q,3] or null (debug)
};

// ------------------------------------------------------------------------------------
// data preparation kernel: one thread per 32-site word of one taxon row
// ------------------------------------------------------------------------------------
__global__ void tq_prepare_rows(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ locus,
                                int64_t S, int64_t Sp, int64_t W, int32_t T, uint8_t *__restrict__ rows,
                                uint4 *__restrict__ planes)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T * W) return;
    int64_t t = gid / W, w = gid - t * W;
    const uint8_t *src = raw + t * S + w * 32;
    uint8_t *dst = rows + t * Sp;
    uint32_t mm = 0, b0 = 0, b1 = 0, rb = 0;
    for (int i = 0; i < 32; ++i) {
        int64_t s = w * 32 + i;
        uint8_t v = (s < S) ? src[i] : (uint8_t)0xFF;
        bool missing = v > 3;
        uint8_t code = missing ? (uint8_t)0 : v;
        dst[row_offset(s)] = code;
        mm |= (uint32_t)missing << i;
        b0 |= (uint32_t)(code & 1) << i;
        b1 |= (uint32_t)((code >> 1) & 1) << i;
        if (s < S) {
            bool beg = (s == 0) || (locus[s] != locus[s - 1]);
            rb |= (uint32_t)beg << i;
        }
    }
    planes[t * W + w] = make_uint4(mm, b0, b1, rb);
}

// ------------------------------------------------------------------------------------
// lexicographic unranking of 4-combinations (combinations.py:94-106)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t choose_k(uint64_t n, int k)
{
    switch (k) {
    case 0: return 1;
    case 1: return n;
    case 2: return n < 2 ? 0 : n * (n - 1) / 2;
    default: return n < 3 ? 0 : n * (n - 1) / 2 * (n - 2) / 3;
    }
}

__global__ void tq_unrank_kernel(const uint64_t *__restrict__ ranks, uint64_t first_rank, int64_t Q,
                                 int32_t T, uint32_t *__restrict__ quartets)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    uint64_t index = ranks ? ranks[i] : first_rank + (uint64_t)i;
    uint32_t out[4] = {0, 0, 0, 0};
    int nsel = 0;
    for (int t = 0; t < T && nsel < 4; ++t) {
        uint64_t c = choose_k((uint64_t)(T - t - 1), 4 - nsel - 1);
        if (c > index) {
            out[nsel++] = (uint32_t)t;
        } else {
            index -= c;
        }
    }
    uint4 v = make_uint4(out[0], out[1], out[2], out[3]);
    reinterpret_cast<uint4 *>(quartets)[i] = v;
}

// sort key of a quartet: its first two taxa (quartets sharing them share two of their four rows)
__global__ void tq_key_kernel(const uint32_t *__restrict__ quartets, int64_t Q, uint32_t T,
                              uint32_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const uint4 q = reinterpret_cast<const uint4 *>(quartets)[i];
    const uint32_t a = q.x < T ? q.x : T - 1, b = q.y < T ? q.y : T - 1;
    keys[i] = a * T + b;
    idx[i] = (uint32_t)i;
}

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

// ------------------------------------------------------------------------------------
// kernel 2: singular values of a 16x16 matrix, one column per lane of a 16-lane group
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double shx(double v, int m) { return __shfl_xor(v, m, WAVE); }

// XOR-partner exchange inside a 16-lane row with DPP moves (VALU) instead of ds_bpermute_b32:
// the LDS crossbar is one unit per CU and a bpermute holds it for 4 cycles, which made the SVD
// stage LDS-issue bound (profiles/r01_v1_baseline).  gfx9 DPP offers the involutions
// quad_perm (lane^1, ^2, ^3), row_half_mirror (lane^7), row_ror:8 (lane^8), row_mirror (lane^15);
// every other XOR mask is a product of two of them.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}

template <int Q>
__device__ __forceinline__ int dpp_quad_xor(int v)
{
    static_assert(Q >= 1 && Q <= 3, "quad xor");
    return Q == 1 ? dpp_mov<0xB1>(v) : Q == 2 ? dpp_mov<0x4E>(v) : dpp_mov<0x1B>(v);
}

template <int M>
__device__ __forceinline__ int dpp_xor16(int v)
{
    static_assert(M >= 1 && M <= 15, "xor mask within a 16-lane row");
    constexpr int DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_ROR8 = 0x128;
    if constexpr (M == 15) return dpp_mov<DPP_ROW_MIRROR>(v);
    else if constexpr (M == 7) return dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    else if constexpr (M == 8) return dpp_mov<DPP_ROW_ROR8>(v);
    else if constexpr (M >= 12) return dpp_quad_xor<(M & 3) ^ 3>(dpp_mov<DPP_ROW_MIRROR>(v));
    else if constexpr (M >= 9) return dpp_quad_xor<M & 3>(dpp_mov<DPP_ROW_ROR8>(v));
    else if constexpr (M >= 4) return dpp_quad_xor<(M & 3) ^ 3>(dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    else return dpp_quad_xor<M>(v);
}

template <int M>
__device__ __forceinline__ double dpx(double v)
{
    const int lo = dpp_xor16<M>(__double2loint(v));
    const int hi = dpp_xor16<M>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int M>
__device__ __forceinline__ void exchange_col(const double (&a)[16], double nrm, double (&b)[16], double &nb)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r] = dpx<M>(a[r]);
    nb = dpx<M>(nrm);
}

__device__ __forceinline__ double group_max(double v)
{
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v = fmax(v, shx(v, m));
    return v;
}

__device__ __forceinline__ double group_sum(double v)
{
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v += shx(v, m);
    return v;
}

struct SvResult {
    double sigma;   // this lane's singular value
    int pos;        // its 0-based position in descending order
    int rank;       // numpy.linalg.matrix_rank rule on the group's 16 values
    double smax;
};

// f64 reciprocal / reciprocal square root from the hardware estimate (v_rcp_f64 / v_rsq_f64)
// plus Newton steps.  NR = 1 gives >= ~2^-45 (enough for the rotation tangent, whose error only
// affects convergence speed), NR = 2 gives full f64 precision (needed for the cosine, which
// scales the columns and therefore the singular values).  tools/probe_math.hip measures both.
template <int NR>
__device__ __forceinline__ double rcp_nr(double v)
{
    double r = __builtin_amdgcn_rcp(v);
#pragma unroll
    for (int i = 0; i < NR; ++i) r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

template <int NR>
__device__ __forceinline__ double rsq_nr(double v)
{
    double y = __builtin_amdgcn_rsq(v);
#pragma unroll
    for (int i = 0; i < NR; ++i) y = fma(0.5 * y, fma(-v * y, y, 1.0), y);
    return y;
}

// One-sided Jacobi, XOR-partner ordering.  Mirrors tests/jacobi_model.py step for step.
//   rotation of the pair (p,q), p < q, alpha = |a_p|^2, beta = |a_q|^2, g = a_p.a_q:
//     d = beta - alpha, h = 2g, t = sign(d) * h / (|d| + sqrt(d^2 + h^2))   (smaller root)
//     c = 1/sqrt(1 + t^2), s = c*t ;  a_p <- c*a_p - s*a_q ;  a_q <- s*a_p + c*a_q
//   a pair is rotated while g^2 > JTOL2*alpha*beta; a sweep in which no pair exceeded
//   JEARLY2 before its rotation is the last one (quadratic convergence squares the
//   remaining off-diagonal, (1e-5)^2 << 2^-50, so the verification sweep is skipped).
__device__ __forceinline__ SvResult jacobi16(double (&a)[16], int j, int lane)
{
    // a 16-lane group stops rotating when ITS matrix has converged, whatever the other three
    // groups of the wave still do: results do not depend on which quartets share a wave
    bool active = true;
    for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
        double nrm = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) nrm = fma(a[r], a[r], nrm);
        // columns below eps * (largest column norm) are numerically zero: frozen, not rotated
        const double zthr = (F64_EPS * F64_EPS) * group_max(nrm);
        bool again = false;
#pragma unroll 1
        for (int m = 1; m < 16; ++m) {
            double b[16];
            double nb;
            switch (m) {                       // wave-uniform: one scalar branch per round
            case 1: exchange_col<1>(a, nrm, b, nb); break;
            case 2: exchange_col<2>(a, nrm, b, nb); break;
            case 3: exchange_col<3>(a, nrm, b, nb); break;
            case 4: exchange_col<4>(a, nrm, b, nb); break;
            case 5: exchange_col<5>(a, nrm, b, nb); break;
            case 6: exchange_col<6>(a, nrm, b, nb); break;
            case 7: exchange_col<7>(a, nrm, b, nb); break;
            case 8: exchange_col<8>(a, nrm, b, nb); break;
            case 9: exchange_col<9>(a, nrm, b, nb); break;
            case 10: exchange_col<10>(a, nrm, b, nb); break;
            case 11: exchange_col<11>(a, nrm, b, nb); break;
            case 12: exchange_col<12>(a, nrm, b, nb); break;
            case 13: exchange_col<13>(a, nrm, b, nb); break;
            case 14: exchange_col<14>(a, nrm, b, nb); break;
            default: exchange_col<15>(a, nrm, b, nb); break;
            }
            double g0 = 0.0, g1 = 0.0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                g0 = fma(a[r], b[r], g0);
                g1 = fma(a[r + 1], b[r + 1], g1);
            }
            const double g = g0 + g1;
            const bool lo = j < (j ^ m);
            const double alpha = lo ? nrm : nb;
            const double beta = lo ? nb : nrm;
            const double ab = alpha * beta;
            const double gg = g * g;
            const bool live = fmin(alpha, beta) > zthr;
            const bool doit = active && live && (gg > JTOL2 * ab);
            again |= live && (gg > JEARLY2 * ab);
            if (__any(doit)) {
                const double d = beta - alpha;
                const double h = doit ? g + g : 1.0;
                const double x = fma(d, d, h * h);
                const double rr = x * rsq_nr<1>(x);                    // sqrt(d^2 + h^2)
                const double tt = h * rcp_nr<1>(fabs(d) + rr);
                const double t = (d < 0.0) ? -tt : tt;
                double c = rsq_nr<2>(fma(t, t, 1.0));
                double sg = lo ? -(c * t) : (c * t);
                c = doit ? c : 1.0;
                sg = doit ? sg : 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = fma(sg, b[r], c * a[r]);
                const double tg = doit ? t * g : 0.0;
                nrm = fmax(lo ? nrm - tg : nrm + tg, 0.0);
            }
        }
        active = active && (((__ballot(again) >> (lane & 48)) & 0xFFFFull) != 0);
        if (!__any(active)) break;
    }
    double nrm = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) nrm = fma(a[r], a[r], nrm);
    SvResult o;
    o.sigma = sqrt(nrm);
    o.smax = group_max(o.sigma);
    int pos = 0;
#pragma unroll 1
    for (int m = 1; m < 16; ++m) {
        const double other = shx(o.sigma, m);
        const int k = j ^ m;
        pos += (other > o.sigma) || (other == o.sigma && k < j);
    }
    o.pos = pos;
    // numpy.linalg.matrix_rank: count(S > S.max() * max(M,N) * eps)
    const double thr = o.smax * 16.0 * F64_EPS;
    const uint64_t bal = __ballot(o.sigma > thr);
    o.rank = __popcll((bal >> (lane & 48)) & 0xFFFFull);
    return o;
}

// bin of element (row r, column j) of flattening t:
//   t=0: rows (i0,i1) cols (i2,i3);  t=1: rows (i0,i2) cols (i1,i3);  t=2: rows (i0,i3) cols (i1,i2)
__device__ __forceinline__ int flat_bin(int t, int r, int j)
{
    if (t == 0) return 16 * r + j;
    const int hi = 64 * (r >> 2) + 16 * (j >> 2);
    if (t == 1) return hi + 4 * (r & 3) + (j & 3);
    return hi + 4 * (j & 3) + (r & 3);
}

// kernel 2: count matrices -> singular values, rank, scores, topology
template <bool DEBUG>
__global__ void __launch_bounds__(WAVE)
tq_svd_kernel(const uint32_t *__restrict__ cm, const uint32_t *__restrict__ quartets, int64_t Q, int32_t T,
              OutPtrs out)
{
    __shared__ uint32_t lds[QPW * 256];
    const int lane = threadIdx.x;
    const int grp = lane >> 4;      // 16-lane group = quartet slot
    const int j = lane & 15;        // column owned

    const int64_t npass = (Q + QPW - 1) / QPW;
    for (int64_t wg = blockIdx.x; wg < npass; wg += gridDim.x) {
        // stage the four 1 KiB count slabs of this pass through LDS (coalesced 16-byte loads)
        {
            const int64_t q0 = wg * QPW;
            const uint4 *src = reinterpret_cast<const uint4 *>(cm + q0 * 256);
            uint4 *dst = reinterpret_cast<uint4 *>(lds);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = lane + WAVE * k;               // 256 uint4 = 4 quartets x 64
                const bool ok = (q0 + (idx >> 6)) < Q;
                dst[idx] = ok ? src[idx] : make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();

        const int64_t myq = wg * QPW + grp;
        const uint32_t *cmq = lds + 256 * grp;
        double sig[3];
        int pos[3], rnk[3];
        double smax_all = 0.0;
        uint32_t my_nsnps = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            double a[16];
            uint32_t colsum = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t v = cmq[flat_bin(t, r, j)];
                a[r] = (double)v;
                colsum += v;
                if (DEBUG) {
                    if (out.cmats && myq < Q) out.cmats[((myq * 3 + t) * 16 + r) * 16 + j] = v;
                }
            }
            if (t == 0) {                                        // resolve_quartets.py:226 cmats[0].sum()
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) colsum += __shfl_xor(colsum, m, WAVE);
                my_nsnps = colsum;
            }
            const SvResult sv = jacobi16(a, j, lane);
            sig[t] = sv.sigma;
            pos[t] = sv.pos;
            rnk[t] = sv.rank;
            smax_all = fmax(smax_all, sv.smax);
            if (DEBUG) {
                if (out.svds && myq < Q) out.svds[(myq * 3 + t) * 16 + sv.pos] = sv.sigma;
                if (out.ranks && myq < Q && j == 0) out.ranks[myq * 3 + t] = sv.rank;
            }
        }
        __syncthreads();   // all reads of the staged slabs done before the next pass overwrites them

        // resolve_quartets.py:246-251
        const int minrank = min(10, min(rnk[0], min(rnk[1], rnk[2])));
        double sc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const double v = (pos[t] >= minrank) ? sig[t] * sig[t] : 0.0;
            sc[t] = sqrt(group_sum(v));
        }
        if (j == 0 && myq < Q) {
            int topo = 0;
            if (sc[1] < sc[topo]) topo = 1;
            if (sc[2] < sc[topo]) topo = 2;
            // gap between the two lowest scores relative to the largest singular value
            const double lo1 = sc[topo];
            const double lo2 = (topo == 0) ? fmin(sc[1], sc[2]) : (topo == 1) ? fmin(sc[0], sc[2]) : fmin(sc[0], sc[1]);
            uint32_t fl = 0;
            if ((lo2 - lo1) <= DEGENERATE_REL_GAP * smax_all) fl |= TQ_FLAG_DEGENERATE;
            if (my_nsnps == 0) {                 // resolve_quartets.py:230-232
                topo = 0;
                sc[0] = sc[1] = sc[2] = 0.001;
                fl = TQ_FLAG_ZERO_DATA;
            }
            const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[myq];
            const uint32_t Tu = (uint32_t)T;
            if ((qv.x >= Tu) | (qv.y >= Tu) | (qv.z >= Tu) | (qv.w >= Tu)) fl |= TQ_FLAG_BAD_INDEX;
            out.rstat[myq * 2 + 0] = (uint32_t)topo;
            out.rstat[myq * 2 + 1] = my_nsnps;
            out.rscor[myq * 3 + 0] = sc[0];
            out.rscor[myq * 3 + 1] = sc[1];
            out.rscor[myq * 3 + 2] = sc[2];
            if (out.flags) out.flags[myq] = (uint8_t)fl;
        }
    }
}

// ====================================================================================
// Alternative singular-value path ("HQR"): Householder bidiagonalisation + implicit-shift QR
// on the bidiagonal -- the same algorithm class as the LAPACK routine the reference calls
// (resolve_quartets.py:242 -> dgesdd -> dgebrd + bidiagonal QR), ~8x fewer f64 operations
// than Jacobi.  Three kernels:
//   tq_bidiag_kernel : 4 lanes per matrix (lane c of a quad holds columns c, c+4, c+8, c+12 as
//                      64 f64 registers), 16 quartets per wave pass; Householder vectors are
//                      shared inside the quad with DPP quad_perm broadcasts, row sums with two
//                      quad_perm butterflies -- no LDS traffic after the count slabs are staged.
//   tq_bdsqr_kernel  : 1 lane per matrix, diagonal/superdiagonal parked in LDS (lane-major, so
//                      dynamically indexed accesses are bank-conflict free); Golub-Kahan
//                      implicit-shift QR sweeps with deflation.
//   tq_score_kernel  : 1 lane per quartet: sort, rank rule, minrank, tail norms, argmin, flags.
// The Jacobi kernel above stays selectable (tq_set_option "svd_method" 0) and is the
// cross-check for this path in the GPU tests.
// ====================================================================================
template <int K>
struct IC {
    static constexpr int value = K;
};

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

template <int SRC>
__device__ __forceinline__ double quad_bcast(double v)
{
    constexpr int CTRL = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);
    const int lo = dpp_mov<CTRL>(__double2loint(v));
    const int hi = dpp_mov<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double quad_sum(double v)
{
    v += dpx<1>(v);
    v += dpx<2>(v);
    return v;
}

__device__ __forceinline__ double sqrt_nr(double x)     // x >= 0, full precision, sqrt(0) = 0
{
    const double y = rsq_nr<2>(x);
    const double s = x * y;
    const double r = fma(fma(-s, s, x), 0.5 * y, s);     // one more correction on the root itself
    return x > 0.0 ? r : 0.0;
}

// de layout: f64 [3*Q][32]: d[0..15] then e[0..15] with e[0] = 0 (e[i] couples columns i-1, i)
template <bool DEBUG>
__global__ void __launch_bounds__(WAVE, 2)
tq_bidiag_kernel(const uint32_t *__restrict__ cm, int64_t Q, double *__restrict__ de,
                 uint32_t *__restrict__ nsnps_out, uint32_t *__restrict__ cmats_dbg)
{
    constexpr int QP = 16;                       // quartets per wave pass (one per quad)
    __shared__ uint32_t lds[QP * 256];
    const int lane = threadIdx.x;
    const int quad = lane >> 2;
    const int c = lane & 3;

    const int64_t npass = (Q + QP - 1) / QP;
    for (int64_t wg = blockIdx.x; wg < npass; wg += gridDim.x) {
        const int64_t q0 = wg * QP;
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(cm + q0 * 256);
            uint4 *dst = reinterpret_cast<uint4 *>(lds);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int idx = lane + WAVE * k;               // 1024 uint4 = 16 quartets x 64
                const bool ok = (q0 + (idx >> 6)) < Q;
                dst[idx] = ok ? src[idx] : make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
        const int64_t myq = q0 + quad;
        const uint32_t *cmq = lds + 256 * quad;
        // resolve_quartets.py:226: number of counted sites = sum of the count tensor
        {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < 64; ++k) s += cmq[4 * k + c];
            s += __shfl_xor(s, 1, WAVE);
            s += __shfl_xor(s, 2, WAVE);
            if (c == 0 && myq < Q) nsnps_out[myq] = s;
        }
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            double a[4][16];                                    // a[s][r] = M_t[r][4s + c]
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t v = cmq[flat_bin(t, r, 4 * s + c)];
                    a[s][r] = (double)v;
                    if (DEBUG) {
                        if (cmats_dbg && myq < Q) cmats_dbg[((myq * 3 + t) * 16 + r) * 16 + 4 * s + c] = v;
                    }
                }
            }
            // d[K] / e[K] are stored as soon as they are known (keeping 32 more f64 live would cost
            // the kernel its second wave per SIMD)
            double *dout = de + (myq * 3 + t) * 32;
            const bool writer = (c == 0) && (myq < Q);
            if (writer) dout[16] = 0.0;
            static_for<0, 16>([&](auto kc) {
                constexpr int K = decltype(kc)::value;
                // ---- left reflector: zero column K below the diagonal ----
                {
                    constexpr int so = K >> 2, co = K & 3;
                    double v[16];
#pragma unroll
                    for (int r = K; r < 16; ++r) v[r] = quad_bcast<co>(a[so][r]);
                    double n2 = 0.0;
#pragma unroll
                    for (int r = K; r < 16; ++r) n2 = fma(v[r], v[r], n2);
                    const double nrm = sqrt_nr(n2);
                    const double x0 = v[K];
                    const double alpha = (x0 < 0.0) ? nrm : -nrm;
                    v[K] = x0 - alpha;
                    const double den = fma(-alpha, x0, n2);      // = |v|^2 / 2 > 0 unless the column is zero
                    const double beta = n2 > 0.0 ? rcp_nr<2>(den) : 0.0;
                    if (writer) dout[K] = alpha;
#pragma unroll
                    for (int s = so; s < 4; ++s) {
                        double w = 0.0;
#pragma unroll
                        for (int r = K; r < 16; ++r) w = fma(v[r], a[s][r], w);
                        w *= beta;
                        if (s == so) w = (c > co) ? w : 0.0;     // columns <= K of this slot are finished
#pragma unroll
                        for (int r = K; r < 16; ++r) a[s][r] = fma(-w, v[r], a[s][r]);
                    }
                }
                // ---- right reflector: zero row K right of the superdiagonal ----
                if constexpr (K <= 13) {
                    constexpr int K1 = K + 1, s1 = K1 >> 2, c1 = K1 & 3, sb = K1 >> 2;
                    double y[4];
                    double p = 0.0;
#pragma unroll
                    for (int s = sb; s < 4; ++s) {
                        y[s] = (4 * s + c > K) ? a[s][K] : 0.0;
                        p = fma(y[s], y[s], p);
                    }
                    const double n2 = quad_sum(p);
                    const double x0 = quad_bcast<c1>(y[s1]);
                    const double nrm = sqrt_nr(n2);
                    const double alpha = (x0 < 0.0) ? nrm : -nrm;
                    const double den = fma(-alpha, x0, n2);
                    const double beta = n2 > 0.0 ? rcp_nr<2>(den) : 0.0;
                    if (c == c1) y[s1] = x0 - alpha;
                    if (writer) dout[16 + K1] = alpha;
#pragma unroll
                    for (int i = K1; i < 16; ++i) {
                        double q = 0.0;
#pragma unroll
                        for (int s = sb; s < 4; ++s) q = fma(y[s], a[s][i], q);
                        const double tt = beta * quad_sum(q);
#pragma unroll
                        for (int s = sb; s < 4; ++s) a[s][i] = fma(-tt, y[s], a[s][i]);
                    }
                } else if constexpr (K == 14) {
                    const double e15 = quad_bcast<3>(a[3][14]);  // column 15 lives in slot 3 of lane 3
                    if (writer) dout[31] = e15;
                }
            });
        }
        __syncthreads();
    }
}

// Golub-Kahan implicit-shift QR on one 16x16 bidiagonal per lane (Golub & Reinsch 1970, the
// diagonalisation half of their SVD procedure, singular values only).  w = diagonal, e =
// superdiagonal (e[0] unused), both parked lane-major in LDS.  sv out: f64 [nmat][16], unsorted, >= 0.
#define W_(i) wl[(i) * WAVE]
#define E_(i) el[(i) * WAVE]
__device__ __forceinline__ double hypot_nr(double a, double b) { return sqrt_nr(fma(a, a, b * b)); }

__global__ void __launch_bounds__(WAVE)
tq_bdsqr_kernel(const double *__restrict__ de, int64_t nmat, double *__restrict__ sv)
{
    __shared__ double lds[32 * WAVE];
    const int lane = threadIdx.x;
    double *wl = lds + lane;
    double *el = lds + 16 * WAVE + lane;
    const int64_t npass = (nmat + WAVE - 1) / WAVE;
    for (int64_t wg = blockIdx.x; wg < npass; wg += gridDim.x) {
        const int64_t m = wg * WAVE + lane;
        const bool live = m < nmat;
        double anorm = 0.0;
        double dv[16], ev[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dv[i] = live ? de[m * 32 + i] : 0.0;
            ev[i] = live ? de[m * 32 + 16 + i] : 0.0;
            W_(i) = dv[i];
            E_(i) = ev[i];
            anorm = fmax(anorm, fabs(dv[i]) + fabs(ev[i]));
        }
        // negligible(x): |x| + anorm == anorm, i.e. |x| <= ~eps/2 * anorm
        const double tiny = anorm * (0.5 * F64_EPS);
        // bit i of negE / negW: e[i] / w[i] is negligible.  Kept in registers and updated on every
        // store, so the split search is a few bit operations instead of a dependent chain of LDS reads.
        uint32_t negE = 0, negW = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            negE |= (uint32_t)(fabs(ev[i]) <= tiny) << i;
            negW |= (uint32_t)(fabs(dv[i]) <= tiny) << i;
        }
#define SET_E(i, v) do { const double v_ = (v); E_(i) = v_; negE = (negE & ~(1u << (i))) | ((uint32_t)(fabs(v_) <= tiny) << (i)); } while (0)
#define SET_W(i, v) do { const double v_ = (v); W_(i) = v_; negW = (negW & ~(1u << (i))) | ((uint32_t)(fabs(v_) <= tiny) << (i)); } while (0)
        // Every lane walks its own deflation index k: a lane whose current singular value has
        // converged moves on at once instead of waiting for the slowest lane of the wave at that k.
        int k = 15, its = 0;
        while (k >= 0) {
            // split point l = largest l <= k with e[l] negligible (or l == 0), unless a negligible
            // w[l-1] is met first (then e[l] has to be chased out of the block: "cancel")
            const uint32_t stopE = negE | 1u, stopW = negW << 1;
            const uint32_t stops = (stopE | stopW) & ((2u << k) - 1u);
            const int l = 31 - __builtin_clz(stops);
            const bool cancel = ((stopE >> l) & 1u) == 0;
            if (cancel) {
                double cc = 0.0, ss = 1.0;
                for (int i = l; i <= k; ++i) {
                    const double ei = E_(i);
                    const double f = ss * ei;
                    SET_E(i, cc * ei);
                    if (fabs(f) <= tiny) break;
                    const double g = W_(i);
                    const double h = hypot_nr(f, g);
                    SET_W(i, h);
                    const double hi = rcp_nr<2>(h);
                    cc = g * hi;
                    ss = -f * hi;
                }
            }
            double z = W_(k);
            if (l == k || its >= 60) {          // converged (or iteration cap: keep what we have)
                W_(k) = fabs(z);
                --k;
                its = 0;
                continue;
            }
            ++its;
            // shift from the bottom 2x2 minor
            double x = W_(l);
            const int nm = k - 1;
            double y = W_(nm);
            double g = E_(nm);
            double h = E_(k);
            double f = ((y - z) * (y + z) + (g - h) * (g + h)) * rcp_nr<2>(2.0 * h * y);
            g = hypot_nr(f, 1.0);
            f = ((x - z) * (x + z) + h * (y * rcp_nr<2>(f + copysign(g, f)) - h)) * rcp_nr<2>(x);
            double cc = 1.0, ss = 1.0;
            // one QR sweep over the block [l,k]; the LDS reads of the next step are issued
            // before the current step's arithmetic, and each hypot shares one rsq with the
            // reciprocal its rotation needs
            double gn = E_(l + 1), yn = W_(l + 1);
            for (int jj = l; jj <= nm; ++jj) {
                g = gn;
                y = yn;
                const int i2 = min(jj + 2, 15);
                gn = E_(i2);
                yn = W_(i2);
                h = ss * g;
                g = cc * g;
                double zz = fma(f, f, h * h);
                double rz = zz > 0.0 ? rsq_nr<2>(zz) : 0.0;
                SET_E(jj, zz * rz);
                cc = f * rz;
                ss = h * rz;
                f = fma(x, cc, g * ss);
                g = fma(g, cc, -(x * ss));
                h = y * ss;
                y *= cc;
                zz = fma(f, f, h * h);
                rz = zz > 0.0 ? rsq_nr<2>(zz) : 0.0;
                SET_W(jj, zz * rz);
                if (zz > 0.0) {
                    cc = f * rz;
                    ss = h * rz;
                }
                f = fma(cc, g, ss * y);
                x = fma(cc, y, -(ss * g));
            }
            SET_E(l, 0.0);
            SET_E(k, f);
            SET_W(k, x);
        }
#undef SET_E
#undef SET_W
        if (live) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[m * 16 + i] = fabs(W_(i));
        }
    }
}
#undef W_
#undef E_

// one lane per quartet: resolve_quartets.py:243-251 on the three sets of singular values
template <bool DEBUG>
__global__ void __launch_bounds__(256)
tq_score_kernel(const double *__restrict__ sv, const uint32_t *__restrict__ nsnps_in,
                const uint32_t *__restrict__ quartets, int64_t Q, int32_t T, OutPtrs out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= Q) return;
    double sc2[3][16];
    int rnk[3];
    double smax_all = 0.0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        double s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sv[(q * 3 + t) * 16 + i];
        // bitonic sorting network, descending (static indices only)
#pragma unroll
        for (int k = 2; k <= 16; k <<= 1) {
#pragma unroll
            for (int jj = k >> 1; jj > 0; jj >>= 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int l = i ^ jj;
                    if (l > i) {
                        const bool desc = (i & k) == 0;
                        const double lo = fmin(s[i], s[l]), hi = fmax(s[i], s[l]);
                        s[i] = desc ? hi : lo;
                        s[l] = desc ? lo : hi;
                    }
                }
            }
        }
        const double thr = s[0] * 16.0 * F64_EPS;        // numpy.linalg.matrix_rank rule
        int r = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            r += s[i] > thr;
            sc2[t][i] = s[i] * s[i];
            if (DEBUG) {
                if (out.svds) out.svds[(q * 3 + t) * 16 + i] = s[i];
            }
        }
        rnk[t] = r;
        smax_all = fmax(smax_all, s[0]);
        if (DEBUG) {
            if (out.ranks) out.ranks[q * 3 + t] = r;
        }
    }
    const int minrank = min(10, min(rnk[0], min(rnk[1], rnk[2])));
    double sc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        double acc = 0.0;
#pragma unroll
        for (int i = 15; i >= 0; --i) acc += (i >= minrank) ? sc2[t][i] : 0.0;
        sc[t] = sqrt(acc);
    }
    int topo = 0;
    if (sc[1] < sc[topo]) topo = 1;
    if (sc[2] < sc[topo]) topo = 2;
    const double lo1 = sc[topo];
    const double lo2 = (topo == 0) ? fmin(sc[1], sc[2]) : (topo == 1) ? fmin(sc[0], sc[2]) : fmin(sc[0], sc[1]);
    uint32_t fl = 0;
    if ((lo2 - lo1) <= DEGENERATE_REL_GAP * smax_all) fl |= TQ_FLAG_DEGENERATE;
    const uint32_t nsn = nsnps_in[q];
    if (nsn == 0) {
        topo = 0;
        sc[0] = sc[1] = sc[2] = 0.001;
        fl = TQ_FLAG_ZERO_DATA;
    }
    const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[q];
    const uint32_t Tu = (uint32_t)T;
    if ((qv.x >= Tu) | (qv.y >= Tu) | (qv.z >= Tu) | (qv.w >= Tu)) fl |= TQ_FLAG_BAD_INDEX;
    out.rstat[q * 2 + 0] = (uint32_t)topo;
    out.rstat[q * 2 + 1] = nsn;
    out.rscor[q * 3 + 0] = sc[0];
    out.rscor[q * 3 + 1] = sc[1];
    out.rscor[q * 3 + 2] = sc[2];
    if (out.flags) out.flags[q] = (uint8_t)fl;
}

// ====================================================================================
// Bootstrap replicate built on the device.
// Reference: resample_tmp_database (tetrad/src/run_inference.py:99-143) = jit_resample
// (tetrad/jit/resample.py:20-64: loci resampled with replacement, columns shuffled inside each
// locus, locus column = ordinal of the resampled locus) + jit_resolve_ambigs
// (tetrad/jit/resolve_ambigs.py:12-36: every IUPAC two-base code resolved to one of its two
// bases with probability 1/2, per cell) + the ACGT -> 0..3 recode (:133-136).  Here the three
// steps and the layout build are fused: the replicate never exists on the host and nothing is
// written back to HDF5.  Random streams: the reference uses numba's Mersenne twister seeded from
// the project Generator; this engine uses counter-based hashes of (seed, position).  Only the
// distribution can match (RNG-stream parity is unpinned); the host keeps the
// reference's draw order on the project Generator (tetrad_amd/bootstrap.py).
// ====================================================================================
__device__ __forceinline__ uint64_t mix64(uint64_t x)     // splitmix64 finaliser
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// widths[i] = spans[lidx[i]][1] - spans[lidx[i]][0]
__global__ void tq_boot_width_kernel(const int64_t *__restrict__ spans, const int64_t *__restrict__ lidxs, int64_t n,
                                     int64_t nloci, uint32_t *__restrict__ widths)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t l = lidxs[i];
    if (l < 0 || l >= nloci) l = 0;
    widths[i] = (uint32_t)(spans[2 * l + 1] - spans[2 * l]);
}

// one thread per resampled locus: Fisher-Yates shuffle of its columns (resample.py:49-50);
// src_col[s] = source column of output site s, site_locus[s] = ordinal of its resampled locus (:58)
__global__ void tq_boot_perm_kernel(const int64_t *__restrict__ spans, const int64_t *__restrict__ lidxs,
                                    const uint32_t *__restrict__ offsets, int64_t n, int64_t nloci, uint64_t seed,
                                    uint32_t *__restrict__ src_col, uint32_t *__restrict__ site_locus)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t l = lidxs[i];
    if (l < 0 || l >= nloci) l = 0;
    const uint32_t start = (uint32_t)spans[2 * l];
    const uint32_t w = (uint32_t)(spans[2 * l + 1] - spans[2 * l]);
    uint32_t *p = src_col + offsets[i];
    uint32_t *loc = site_locus + offsets[i];
    for (uint32_t j = 0; j < w; ++j) {
        p[j] = start + j;
        loc[j] = (uint32_t)i;
    }
    uint64_t state = mix64(seed ^ ((uint64_t)i * 0xD1342543DE82EF95ull));
    for (uint32_t j = w; j > 1; --j) {
        state = mix64(state);
        // unbiased enough for j << 2^32: multiply-high of a 32-bit draw
        const uint32_t r = (uint32_t)(((state >> 32) * (uint64_t)j) >> 32);
        const uint32_t tmp = p[j - 1];
        p[j - 1] = p[r];
        p[r] = tmp;
    }
}

// one thread per 32-site word of one taxon row: gather + ambiguity resolution + recode + layout
__global__ void tq_boot_build_kernel(const uint8_t *__restrict__ seqarr, int64_t S0,
                                     const uint32_t *__restrict__ src_col, const uint32_t *__restrict__ site_locus,
                                     int64_t S, int64_t Sp, int64_t W, int32_t T, uint64_t seed,
                                     uint8_t *__restrict__ rows, uint4 *__restrict__ planes)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T * W) return;
    const int64_t t = gid / W, w = gid - t * W;
    uint8_t *dst = rows + t * Sp;
    uint32_t mm = 0, b0 = 0, b1 = 0, rb = 0;
    for (int i = 0; i < 32; ++i) {
        const int64_t s = w * 32 + i;
        uint8_t code = 0;
        bool missing = true;
        if (s < S) {
            uint8_t v = seqarr[t * S0 + src_col[s]];
            // IUPAC two-base codes (utils.py:14-21): R->G/A K->G/T S->G/C Y->T/C W->T/A M->C/A
            const bool coin = (mix64(seed ^ ((uint64_t)t * 0x9E3779B97F4A7C15ull) ^ (uint64_t)s * 0xC2B2AE3D27D4EB4Full) >> 63) != 0;
            switch (v) {
            case 82: v = coin ? 71 : 65; break;
            case 75: v = coin ? 71 : 84; break;
            case 83: v = coin ? 71 : 67; break;
            case 89: v = coin ? 84 : 67; break;
            case 87: v = coin ? 84 : 65; break;
            case 77: v = coin ? 67 : 65; break;
            default: break;
            }
            // run_inference.py:133-136: A,C,G,T -> 0,1,2,3 ; everything else stays a byte > 3 (missing)
            if (v == 65) { code = 0; missing = false; }
            else if (v == 67) { code = 1; missing = false; }
            else if (v == 71) { code = 2; missing = false; }
            else if (v == 84) { code = 3; missing = false; }
            else if (v <= 3) { code = v; missing = false; }       // already recoded input
            const bool beg = (s == 0) || (site_locus[s] != site_locus[s - 1]);
            rb |= (uint32_t)beg << i;
        }
        dst[row_offset(s)] = code;
        mm |= (uint32_t)missing << i;
        b0 |= (uint32_t)(code & 1) << i;
        b1 |= (uint32_t)((code >> 1) & 1) << i;
    }
    planes[t * W + w] = make_uint4(mm, b0, b1, rb);
}

// replicate currently on the device -> the reference's tmparr (0..3, 78) / tmpmap layout
__global__ void tq_export_kernel(const uint8_t *__restrict__ rows, const uint4 *__restrict__ planes, int64_t S,
                                 int64_t Sp, int64_t W, int32_t T, uint8_t *__restrict__ tmparr,
                                 uint32_t *__restrict__ tmpmap)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T * S) return;
    const int64_t t = gid / S, s = gid - t * S;
    const uint32_t miss = planes[t * W + (s >> 5)].x;
    tmparr[gid] = ((miss >> (s & 31)) & 1u) ? (uint8_t)78 : rows[t * Sp + row_offset(s)];
    if (t == 0) tmpmap[2 * s + 1] = (uint32_t)s;
}

// locus ordinals from the run-begin bits (inclusive prefix count - 1), one thread per 32-site word
__global__ void tq_export_runcount_kernel(const uint4 *__restrict__ planes, int64_t W, uint32_t *__restrict__ cnt)
{
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < W) cnt[w] = (uint32_t)__popc(planes[w].w);
}

__global__ void tq_export_locus_kernel(const uint4 *__restrict__ planes, const uint32_t *__restrict__ base, int64_t S,
                                       int64_t W, uint32_t *__restrict__ tmpmap)
{
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const uint32_t rb = planes[w].w;
    uint32_t ord = base[w];                  // run-begins before this word
    for (int i = 0; i < 32; ++i) {
        const int64_t s = w * 32 + i;
        if (s >= S) break;
        ord += (rb >> i) & 1u;
        tmpmap[2 * s] = ord - 1u;
    }
}

}  // namespace

// ======================================================================================
// host side: context + C ABI
// ======================================================================================
struct tq_ctx {
    int device = 0;
    std::string err;
    hipDeviceProp_t prop{};
    // replicate data
    int64_t T = 0, S = 0, Sp = 0, W = 0;
    uint8_t *d_rows = nullptr;
    uint4 *d_planes = nullptr;      // [T][W] {miss, p0, p1, runbeg}
    bool have_data = false;
    bool locus_runs_ok = false;
    int64_t data_capacity = 0;      // allocated Sp (rows/planes are re-used by bootstrap replicates)
    // bootstrap source (tq_set_source): ASCII seqarr [T][S0], spans i64 [nloci][2]
    uint8_t *d_seqarr = nullptr;
    int64_t *d_spans = nullptr;
    int64_t src_T = 0, src_S0 = 0, nloci = 0, max_width = 0;
    int64_t *d_lidxs = nullptr;     // [nloci]
    uint32_t *d_boot = nullptr;     // widths/offsets [nloci+1] | src_col [cap] | site_locus [cap]
    int64_t boot_cap = 0;
    void *d_boot_tmp = nullptr;
    size_t boot_tmp_bytes = 0;
    // scratch for the host-buffer API
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // count-matrix slab between the two kernels: u32 [batch][256]
    uint32_t *d_cm = nullptr;
    int64_t cm_quartets = 0;
    // (a,b)-sorted processing order of the current batch: keys/idx in, keys/idx out, cub temp
    uint32_t *d_sort = nullptr;     // 4 arrays of cm_quartets u32
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int order = 1;                  // 1 = process quartets in (a,b)-sorted order
    // HQR path scratch: de f64[3*batch][32], sv f64[3*batch][16], nsnps u32[batch]
    double *d_de = nullptr, *d_sv = nullptr;
    uint32_t *d_nsnps = nullptr;
    int svd_method = 1;             // 0 = one-sided Jacobi (tq_svd_kernel), 1 = Householder + bidiagonal QR
    int scan_wg = 8;                // waves per workgroup of the cooperative scan kernel (0 = one wave per quartet)
    // software pipeline across sub-batches: scan of sub-batch i+1 runs on a second stream beside the
    // singular-value stage of sub-batch i (0 = off: one stage after the other on the caller's stream)
    int64_t overlap = 0;            // sub-batch size in quartets
    int ov_scan_wgs = 1;            // scan workgroups per CU while overlapping
    int ov_svd_waves = 6;           // singular-value-stage waves per CU while overlapping
    hipStream_t sA = nullptr, sB = nullptr;
    hipEvent_t evIn = nullptr, evA[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr}, evEndA = nullptr, evEndB = nullptr;
    int wpc_override = 0;           // transient: grid_for() uses this instead of waves_per_cu when > 0
    // options
    int nrep = 1;
    int waves_per_cu = 0;           // 0 = from the occupancy query
    int phases = 3;                 // diagnostics only: 1 = scan kernel only, 2 = SVD kernel only
    int scan_method = -1;           // 0 = EXEC-masked slot per site, 1 = set-bit walk, -1 = 1 if subsample else 0
    int64_t batch = 1 << 20;        // quartets per scan->svd batch (1 GiB slab)
    // timing: per resolve call one (start, mid, stop) triple per batch
    bool timing = false;
    struct Ev { hipEvent_t e0, e1, e2; };
    std::vector<Ev> events;
    size_t events_used = 0;
    int64_t timed_calls = 0;
};

namespace {

std::string g_create_err;

int fail(tq_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

#define TQ_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, e_ == hipErrorOutOfMemory ? TQ_ERR_OOM : TQ_ERR_HIP, "%s failed: %s", \
                        #call, hipGetErrorString(e_));                                        \
    } while (0)

void free_data(tq_ctx *ctx)
{
    if (ctx->d_rows) (void)hipFree(ctx->d_rows);
    if (ctx->d_planes) (void)hipFree(ctx->d_planes);
    ctx->d_rows = nullptr;
    ctx->d_planes = nullptr;
    ctx->have_data = false;
    ctx->data_capacity = 0;
}

void free_source(tq_ctx *ctx)
{
    if (ctx->d_seqarr) (void)hipFree(ctx->d_seqarr);
    if (ctx->d_spans) (void)hipFree(ctx->d_spans);
    if (ctx->d_lidxs) (void)hipFree(ctx->d_lidxs);
    if (ctx->d_boot) (void)hipFree(ctx->d_boot);
    if (ctx->d_boot_tmp) (void)hipFree(ctx->d_boot_tmp);
    ctx->d_seqarr = nullptr;
    ctx->d_spans = nullptr;
    ctx->d_lidxs = nullptr;
    ctx->d_boot = nullptr;
    ctx->d_boot_tmp = nullptr;
    ctx->boot_cap = 0;
    ctx->nloci = 0;
}

int ensure_scratch(tq_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return TQ_OK;
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    TQ_HIP(ctx, hipMalloc(&ctx->d_scratch, bytes));
    ctx->scratch_bytes = bytes;
    return TQ_OK;
}

int ensure_cm(tq_ctx *ctx, int64_t quartets)
{
    if (quartets <= ctx->cm_quartets) return TQ_OK;
    if (ctx->d_cm) (void)hipFree(ctx->d_cm);
    if (ctx->d_sort) (void)hipFree(ctx->d_sort);
    if (ctx->d_sort_tmp) (void)hipFree(ctx->d_sort_tmp);
    if (ctx->d_de) (void)hipFree(ctx->d_de);
    if (ctx->d_sv) (void)hipFree(ctx->d_sv);
    if (ctx->d_nsnps) (void)hipFree(ctx->d_nsnps);
    ctx->d_cm = nullptr;
    ctx->d_sort = nullptr;
    ctx->d_sort_tmp = nullptr;
    ctx->d_de = ctx->d_sv = nullptr;
    ctx->d_nsnps = nullptr;
    ctx->cm_quartets = 0;
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_cm, (size_t)quartets * 1024 * 2));   // two slabs (overlap mode)
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_de, (size_t)quartets * 3 * 32 * sizeof(double)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_sv, (size_t)quartets * 3 * 16 * sizeof(double)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_nsnps, (size_t)quartets * sizeof(uint32_t)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_sort, (size_t)quartets * 16));
    size_t tmp = 0;
    uint32_t *k = ctx->d_sort;
    TQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, k, k, k, k, (int)quartets));
    TQ_HIP(ctx, hipMalloc(&ctx->d_sort_tmp, tmp ? tmp : 16));
    ctx->sort_tmp_bytes = tmp;
    ctx->cm_quartets = quartets;
    return TQ_OK;
}

// order[] for one batch: indices sorted by (first taxon, second taxon); nullptr = natural order
int make_order(tq_ctx *ctx, const uint32_t *dq, int64_t n, hipStream_t stream, const uint32_t **order)
{
    *order = nullptr;
    if (!ctx->order || n < 1024 || ctx->T > 65535) return TQ_OK;
    uint32_t *keys_in = ctx->d_sort, *idx_in = keys_in + ctx->cm_quartets;
    uint32_t *keys_out = idx_in + ctx->cm_quartets, *idx_out = keys_out + ctx->cm_quartets;
    hipLaunchKernelGGL(tq_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dq, n,
                       (uint32_t)ctx->T, keys_in, idx_in);
    TQ_HIP(ctx, hipGetLastError());
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (uint64_t)ctx->T * (uint64_t)ctx->T) ++bits;
    size_t tmp = ctx->sort_tmp_bytes;
    TQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->d_sort_tmp, tmp, keys_in, keys_out, idx_in, idx_out, (int)n,
                                                   0, bits, stream));
    *order = idx_out;
    return TQ_OK;
}

DevData dev_data(const tq_ctx *ctx)
{
    DevData d;
    d.rows = ctx->d_rows;
    d.planes = ctx->d_planes;
    d.pitch = ctx->Sp;
    d.W = ctx->W;
    d.T = (int32_t)ctx->T;
    d.ntiles = (int32_t)(ctx->Sp / TILE);
    return d;
}

template <typename K>
int grid_for(tq_ctx *ctx, K kern, int64_t items, int64_t *grid)
{
    int wpc = ctx->wpc_override > 0 ? ctx->wpc_override : ctx->waves_per_cu;
    if (wpc <= 0) {
        int nb = 0;
        TQ_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, WAVE, 0));
        wpc = nb > 0 ? nb : 8;
    }
    int64_t g = (int64_t)ctx->prop.multiProcessorCount * wpc;
    if (g > items) g = items;
    if (g < 1) g = 1;
    *grid = g;
    return TQ_OK;
}

template <int NREP, bool SUB, int METHOD>
int launch_scan(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, hipStream_t stream)
{
    auto kern = tq_scan_kernel<NREP, SUB, METHOD>;
    int64_t grid;
    int rc = grid_for(ctx, kern, Q, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVE), 0, stream, dev_data(ctx), dq, order, Q, ctx->d_cm);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

template <bool SUB, int METHOD, int NW>
int launch_scan_wg(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, hipStream_t stream)
{
    auto kern = tq_scan_wg_kernel<SUB, METHOD, NW>;
    int wgs = ctx->wpc_override > 0 ? (ctx->wpc_override + NW - 1) / NW
              : ctx->waves_per_cu > 0 ? (ctx->waves_per_cu + NW - 1) / NW : 0;
    if (wgs <= 0) {
        int nb = 0;
        TQ_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NW * WAVE, 0));
        wgs = nb > 0 ? nb : 1;
    }
    int64_t grid = (int64_t)ctx->prop.multiProcessorCount * wgs;
    const int64_t nblk = (Q + NW - 1) / NW;
    if (grid > nblk) grid = nblk;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * WAVE), 0, stream, dev_data(ctx), dq, order, Q,
                       ctx->d_cm);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

int launch_scan_n(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, int subsample,
                  hipStream_t stream)
{
    if (ctx->scan_wg == 8 && Q >= 64 && (uint64_t)ctx->T * (uint64_t)ctx->Sp < 0xFFFF0000ull) {
        const int m = ctx->scan_method < 0 ? (subsample ? 1 : 0) : ctx->scan_method;
        if (m == 2) return launch_scan_wg<true, 2, 8>(ctx, dq, order, Q, stream);   // diagnostic
        if (subsample)
            return m ? launch_scan_wg<true, 1, 8>(ctx, dq, order, Q, stream)
                     : launch_scan_wg<true, 0, 8>(ctx, dq, order, Q, stream);
        return m ? launch_scan_wg<false, 1, 8>(ctx, dq, order, Q, stream)
                 : launch_scan_wg<false, 0, 8>(ctx, dq, order, Q, stream);
    }
#define TQ_SCAN_CASE(N)                                                                              \
    case N:                                                                                          \
        if (method == 0)                                                                             \
            return subsample ? launch_scan<N, true, 0>(ctx, dq, order, Q, stream)                    \
                             : launch_scan<N, false, 0>(ctx, dq, order, Q, stream);                  \
        return subsample ? launch_scan<N, true, 1>(ctx, dq, order, Q, stream)                        \
                         : launch_scan<N, false, 1>(ctx, dq, order, Q, stream)
    const int method = ctx->scan_method < 0 ? (subsample ? 1 : 0) : ctx->scan_method;
    switch (ctx->nrep) {
        TQ_SCAN_CASE(2);
        TQ_SCAN_CASE(4);
        TQ_SCAN_CASE(8);
        TQ_SCAN_CASE(16);
        TQ_SCAN_CASE(32);
    default:
        TQ_SCAN_CASE(1);
    }
#undef TQ_SCAN_CASE
}

template <bool DEBUG>
int launch_svd(tq_ctx *ctx, const uint32_t *dq, int64_t Q, const OutPtrs &out, hipStream_t stream)
{
    auto kern = tq_svd_kernel<DEBUG>;
    int64_t grid;
    int rc = grid_for(ctx, kern, (Q + QPW - 1) / QPW, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVE), 0, stream, (const uint32_t *)ctx->d_cm, dq, Q,
                       (int32_t)ctx->T, out);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

template <bool DEBUG>
int launch_hqr(tq_ctx *ctx, const uint32_t *dq, int64_t Q, const OutPtrs &out, hipStream_t stream)
{
    int64_t grid;
    auto k1 = tq_bidiag_kernel<DEBUG>;
    int rc = grid_for(ctx, k1, (Q + 15) / 16, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(k1, dim3((unsigned)grid), dim3(WAVE), 0, stream, (const uint32_t *)ctx->d_cm, Q, ctx->d_de,
                       ctx->d_nsnps, out.cmats);
    TQ_HIP(ctx, hipGetLastError());
    const int64_t nmat = 3 * Q;
    rc = grid_for(ctx, tq_bdsqr_kernel, (nmat + WAVE - 1) / WAVE, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(tq_bdsqr_kernel, dim3((unsigned)grid), dim3(WAVE), 0, stream, (const double *)ctx->d_de, nmat,
                       ctx->d_sv);
    TQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(tq_score_kernel<DEBUG>, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, stream,
                       (const double *)ctx->d_sv, (const uint32_t *)ctx->d_nsnps, dq, Q, (int32_t)ctx->T, out);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

OutPtrs offset_out(const OutPtrs &o, int64_t q0)
{
    OutPtrs r = o;
    r.rstat = o.rstat + q0 * 2;
    r.rscor = o.rscor + q0 * 3;
    r.flags = o.flags ? o.flags + q0 : nullptr;
    r.cmats = o.cmats ? o.cmats + q0 * 768 : nullptr;
    r.svds = o.svds ? o.svds + q0 * 48 : nullptr;
    r.ranks = o.ranks ? o.ranks + q0 * 3 : nullptr;
    return r;
}

int launch_overlapped(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, const OutPtrs &out,
                      hipStream_t stream);

// scan kernel -> cm slab -> SVD kernel, in batches so that the slab stays <= batch KiB
int launch(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, bool debug, const OutPtrs &out,
           hipStream_t stream)
{
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (subsample && !ctx->locus_runs_ok)
        return fail(ctx, TQ_ERR_LOCUS_ORDER,
                    "subsample mode needs each locus id in one contiguous run of sites (and no id 0xFFFFFFFF)");
    if (Q == 0) return TQ_OK;
    if (ctx->overlap > 0 && !debug && ctx->phases == 3 && Q > ctx->overlap)
        return launch_overlapped(ctx, dq, Q, subsample, out, stream);
    const int64_t batch = Q < ctx->batch ? Q : ctx->batch;
    int rc = ensure_cm(ctx, batch);
    if (rc) return rc;
    if (ctx->timing) ctx->timed_calls++;
    for (int64_t q0 = 0; q0 < Q; q0 += batch) {
        const int64_t n = (Q - q0) < batch ? (Q - q0) : batch;
        tq_ctx::Ev ev{};
        if (ctx->timing) {
            if (ctx->events_used == ctx->events.size()) {
                tq_ctx::Ev e{};
                TQ_HIP(ctx, hipEventCreate(&e.e0));
                TQ_HIP(ctx, hipEventCreate(&e.e1));
                TQ_HIP(ctx, hipEventCreate(&e.e2));
                ctx->events.push_back(e);
            }
            ev = ctx->events[ctx->events_used++];
            TQ_HIP(ctx, hipEventRecord(ev.e0, stream));
        }
        if (ctx->phases & 1) {
            const uint32_t *order = nullptr;
            rc = make_order(ctx, dq + q0 * 4, n, stream, &order);
            if (rc) return rc;
            rc = launch_scan_n(ctx, dq + q0 * 4, order, n, subsample, stream);
            if (rc) return rc;
        }
        if (ctx->timing) TQ_HIP(ctx, hipEventRecord(ev.e1, stream));
        if (ctx->phases & 2) {
            const OutPtrs o = offset_out(out, q0);
            if (ctx->svd_method == 0)
                rc = debug ? launch_svd<true>(ctx, dq + q0 * 4, n, o, stream)
                           : launch_svd<false>(ctx, dq + q0 * 4, n, o, stream);
            else
                rc = debug ? launch_hqr<true>(ctx, dq + q0 * 4, n, o, stream)
                           : launch_hqr<false>(ctx, dq + q0 * 4, n, o, stream);
            if (rc) return rc;
        }
        if (ctx->timing) TQ_HIP(ctx, hipEventRecord(ev.e2, stream));
    }
    return TQ_OK;
}

// Overlapped form of launch(): the batch is cut into sub-batches; scan(i+1) runs on stream A beside
// the singular-value stage of sub-batch i on stream B (two count slabs).  Both grids are sized to a
// fraction of each CU so that the two stages are co-resident: the scan is an L2/LDS/integer mix at
// ~50 % VALU, the bidiagonal QR a latency-bound f64 chain at ~40 % -- they fill each other's gaps.
int launch_overlapped(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, const OutPtrs &out,
                      hipStream_t stream)
{
    const int64_t sub = ctx->overlap;
    int rc = ensure_cm(ctx, sub);
    if (rc) return rc;
    if (!ctx->sA) {
        TQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->sA, hipStreamNonBlocking));
        TQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->sB, hipStreamNonBlocking));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evIn, hipEventDisableTiming));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evEndA, hipEventDisableTiming));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evEndB, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) {
            TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evA[i], hipEventDisableTiming));
            TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evB[i], hipEventDisableTiming));
        }
    }
    tq_ctx::Ev ev{};
    if (ctx->timing) {
        ctx->timed_calls++;
        if (ctx->events_used == ctx->events.size()) {
            tq_ctx::Ev e{};
            TQ_HIP(ctx, hipEventCreate(&e.e0));
            TQ_HIP(ctx, hipEventCreate(&e.e1));
            TQ_HIP(ctx, hipEventCreate(&e.e2));
            ctx->events.push_back(e);
        }
        ev = ctx->events[ctx->events_used++];
        TQ_HIP(ctx, hipEventRecord(ev.e0, stream));
        TQ_HIP(ctx, hipEventRecord(ev.e1, stream));       // stages overlap: only the total is meaningful
    }
    TQ_HIP(ctx, hipEventRecord(ctx->evIn, stream));
    TQ_HIP(ctx, hipStreamWaitEvent(ctx->sA, ctx->evIn, 0));
    TQ_HIP(ctx, hipStreamWaitEvent(ctx->sB, ctx->evIn, 0));
    uint32_t *cm0 = ctx->d_cm;
    int64_t i = 0;
    for (int64_t q0 = 0; q0 < Q; q0 += sub, ++i) {
        const int64_t n = (Q - q0) < sub ? (Q - q0) : sub;
        const int b = (int)(i & 1);
        // stream A: ordering + scan into slab b (after the SVD stage that last read slab b)
        if (i >= 2) TQ_HIP(ctx, hipStreamWaitEvent(ctx->sA, ctx->evB[b], 0));
        ctx->d_cm = cm0 + (size_t)b * (size_t)ctx->cm_quartets * 256;
        const uint32_t *order = nullptr;
        rc = make_order(ctx, dq + q0 * 4, n, ctx->sA, &order);
        if (!rc) {
            ctx->wpc_override = ctx->ov_scan_wgs * 8;
            rc = launch_scan_n(ctx, dq + q0 * 4, order, n, subsample, ctx->sA);
        }
        if (!rc && hipEventRecord(ctx->evA[b], ctx->sA) != hipSuccess) rc = TQ_ERR_HIP;
        // stream B: singular values of slab b
        if (!rc && hipStreamWaitEvent(ctx->sB, ctx->evA[b], 0) != hipSuccess) rc = TQ_ERR_HIP;
        if (!rc) {
            ctx->wpc_override = ctx->ov_svd_waves;
            const OutPtrs o = offset_out(out, q0);
            rc = ctx->svd_method == 0 ? launch_svd<false>(ctx, dq + q0 * 4, n, o, ctx->sB)
                                      : launch_hqr<false>(ctx, dq + q0 * 4, n, o, ctx->sB);
        }
        if (!rc && hipEventRecord(ctx->evB[b], ctx->sB) != hipSuccess) rc = TQ_ERR_HIP;
        if (rc) break;
    }
    ctx->wpc_override = 0;
    ctx->d_cm = cm0;
    if (rc) return rc > 0 ? TQ_ERR_HIP : rc;
    TQ_HIP(ctx, hipEventRecord(ctx->evEndA, ctx->sA));
    TQ_HIP(ctx, hipEventRecord(ctx->evEndB, ctx->sB));
    TQ_HIP(ctx, hipStreamWaitEvent(stream, ctx->evEndA, 0));
    TQ_HIP(ctx, hipStreamWaitEvent(stream, ctx->evEndB, 0));
    if (ctx->timing) TQ_HIP(ctx, hipEventRecord(ev.e2, stream));
    return TQ_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int tq_create(tq_ctx **out, int device_id)
{
    if (!out) return fail(nullptr, TQ_ERR_INVALID_ARG, "tq_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, TQ_ERR_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, TQ_ERR_INVALID_ARG, "device_id %d out of range (0..%d)", device_id, n - 1);
    tq_ctx *ctx = new (std::nothrow) tq_ctx();
    if (!ctx) return fail(nullptr, TQ_ERR_OOM, "out of host memory");
    ctx->device = device_id;
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device_id);
    if (e != hipSuccess) {
        int rc = fail(nullptr, TQ_ERR_HIP, "device %d not usable: %s", device_id, hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    *out = ctx;
    return TQ_OK;
}

void tq_destroy(tq_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_data(ctx);
    free_source(ctx);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_cm) (void)hipFree(ctx->d_cm);
    if (ctx->d_sort) (void)hipFree(ctx->d_sort);
    if (ctx->d_sort_tmp) (void)hipFree(ctx->d_sort_tmp);
    if (ctx->d_de) (void)hipFree(ctx->d_de);
    if (ctx->d_sv) (void)hipFree(ctx->d_sv);
    if (ctx->d_nsnps) (void)hipFree(ctx->d_nsnps);
    if (ctx->sA) {
        (void)hipStreamDestroy(ctx->sA);
        (void)hipStreamDestroy(ctx->sB);
        (void)hipEventDestroy(ctx->evIn);
        (void)hipEventDestroy(ctx->evEndA);
        (void)hipEventDestroy(ctx->evEndB);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy(ctx->evA[i]);
            (void)hipEventDestroy(ctx->evB[i]);
        }
    }
    for (auto &e : ctx->events) {
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
        (void)hipEventDestroy(e.e2);
    }
    delete ctx;
}

const char *tq_last_error(const tq_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int tq_set_data(tq_ctx *ctx, const uint8_t *tmparr, int64_t T, int64_t S, const uint32_t *locus,
                int64_t locus_stride)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!tmparr || !locus) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_data: NULL pointer");
    if (T < 1 || S < 1 || locus_stride < 1 || T > 0x7FFFFFFF)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_data: bad shape T=%lld S=%lld stride=%lld", (long long)T,
                    (long long)S, (long long)locus_stride);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    free_data(ctx);

    // contiguous copy of the locus column + the run-contiguity check subsample mode relies on
    std::vector<uint32_t> loc((size_t)S);
    bool ok = true, sorted = true;
    for (int64_t i = 0; i < S; ++i) {
        loc[(size_t)i] = locus[i * locus_stride];
        if (loc[(size_t)i] == 0xFFFFFFFFu) ok = false;
        if (i && loc[(size_t)i] < loc[(size_t)i - 1]) sorted = false;
    }
    if (ok && !sorted) {
        std::unordered_set<uint32_t> seen;
        seen.insert(loc[0]);
        for (int64_t i = 1; i < S && ok; ++i)
            if (loc[(size_t)i] != loc[(size_t)i - 1] && !seen.insert(loc[(size_t)i]).second) ok = false;
    }
    ctx->locus_runs_ok = ok;

    const int64_t Sp = (int64_t)align_up((size_t)S, TILE);
    const int64_t W = Sp / 32;
    ctx->T = T; ctx->S = S; ctx->Sp = Sp; ctx->W = W;
    ctx->data_capacity = Sp;
    uint8_t *d_raw = nullptr;
    uint32_t *d_loc = nullptr;
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)(T * Sp)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)(T * W) * sizeof(uint4)));
    TQ_HIP(ctx, hipMalloc((void **)&d_raw, (size_t)(T * S)));
    hipError_t e = hipMalloc((void **)&d_loc, (size_t)S * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(d_raw);
        return fail(ctx, TQ_ERR_OOM, "hipMalloc(locus) failed: %s", hipGetErrorString(e));
    }
    e = hipMemcpy(d_raw, tmparr, (size_t)(T * S), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_loc, loc.data(), (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int64_t n = T * W;
        hipLaunchKernelGGL(tq_prepare_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_raw, d_loc, S, Sp,
                           W, (int32_t)T, ctx->d_rows, ctx->d_planes);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    (void)hipFree(d_raw);
    (void)hipFree(d_loc);
    if (e != hipSuccess) {
        free_data(ctx);
        return fail(ctx, TQ_ERR_HIP, "tq_set_data: %s", hipGetErrorString(e));
    }
    ctx->have_data = true;
    return TQ_OK;
}

int tq_resolve_dev(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, int subsample, uint32_t *d_rstat,
                   double *d_rscor, uint8_t *d_flags, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (Q < 0 || (Q > 0 && (!d_quartets || !d_rstat || !d_rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve_dev: NULL pointer or negative Q");
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    OutPtrs out{d_rstat, d_rscor, d_flags, nullptr, nullptr, nullptr};
    return launch(ctx, d_quartets, Q, subsample, false, out, (hipStream_t)stream);
}

int tq_unrank_dev(tq_ctx *ctx, const uint64_t *d_ranks, int64_t Q, uint32_t *d_quartets, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (Q < 0 || (Q > 0 && (!d_ranks || !d_quartets)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_unrank_dev: NULL pointer or negative Q");
    if (Q == 0) return TQ_OK;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(tq_unrank_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_ranks, (uint64_t)0, Q, (int32_t)ctx->T, d_quartets);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

int tq_resolve_range_dev(tq_ctx *ctx, uint64_t first_rank, int64_t Q, int subsample, uint32_t *d_quartets,
                         uint32_t *d_rstat, double *d_rscor, uint8_t *d_flags, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (Q < 0 || (Q > 0 && (!d_rstat || !d_rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve_range_dev: NULL pointer or negative Q");
    if (Q == 0) return TQ_OK;
    const uint64_t T = (uint64_t)ctx->T;
    const uint64_t total = T < 4 ? 0 : T * (T - 1) / 2 * (T - 2) / 3 * (T - 3) / 4;
    if (first_rank + (uint64_t)Q > total)
        return fail(ctx, TQ_ERR_INVALID_ARG, "rank range [%llu,+%lld) exceeds C(%lld,4)=%llu",
                    (unsigned long long)first_rank, (long long)Q, (long long)ctx->T, (unsigned long long)total);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *dq = d_quartets;
    if (!dq) {
        int rc = ensure_scratch(ctx, (size_t)Q * 16);
        if (rc) return rc;
        dq = (uint32_t *)ctx->d_scratch;
    }
    hipLaunchKernelGGL(tq_unrank_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint64_t *)nullptr, first_rank, Q, (int32_t)ctx->T, dq);
    TQ_HIP(ctx, hipGetLastError());
    OutPtrs out{d_rstat, d_rscor, d_flags, nullptr, nullptr, nullptr};
    return launch(ctx, dq, Q, subsample, false, out, (hipStream_t)stream);
}

int tq_resolve_debug(tq_ctx *ctx, const uint32_t *quartets, int64_t Q, int subsample, uint32_t *rstat,
                     double *rscor, uint8_t *flags, uint32_t *cmats, double *svds, int32_t *ranks)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (Q < 0 || (Q > 0 && (!quartets || !rstat || !rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve: NULL pointer or negative Q");
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (Q == 0) return TQ_OK;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    // taxon indices are checked on the host here; the kernel re-checks and flags them
    for (int64_t i = 0; i < Q * 4; ++i)
        if (quartets[i] >= (uint64_t)ctx->T)
            return fail(ctx, TQ_ERR_INVALID_ARG, "quartet %lld has taxon index %u >= T=%lld", (long long)(i / 4),
                        quartets[i], (long long)ctx->T);
    const bool debug = cmats || svds || ranks;
    const size_t o_q = 0;
    const size_t o_rstat = align_up(o_q + (size_t)Q * 16, 256);
    const size_t o_rscor = align_up(o_rstat + (size_t)Q * 8, 256);
    const size_t o_flags = align_up(o_rscor + (size_t)Q * 24, 256);
    const size_t o_cm = align_up(o_flags + (size_t)Q, 256);
    const size_t o_sv = align_up(o_cm + (cmats ? (size_t)Q * 3072 : 0), 256);
    const size_t o_rk = align_up(o_sv + (svds ? (size_t)Q * 384 : 0), 256);
    const size_t total = align_up(o_rk + (ranks ? (size_t)Q * 12 : 0), 256);
    int rc = ensure_scratch(ctx, total);
    if (rc) return rc;
    char *base = (char *)ctx->d_scratch;
    TQ_HIP(ctx, hipMemcpy(base + o_q, quartets, (size_t)Q * 16, hipMemcpyHostToDevice));
    OutPtrs out;
    out.rstat = (uint32_t *)(base + o_rstat);
    out.rscor = (double *)(base + o_rscor);
    out.flags = (uint8_t *)(base + o_flags);
    out.cmats = cmats ? (uint32_t *)(base + o_cm) : nullptr;
    out.svds = svds ? (double *)(base + o_sv) : nullptr;
    out.ranks = ranks ? (int32_t *)(base + o_rk) : nullptr;
    rc = launch(ctx, (const uint32_t *)(base + o_q), Q, subsample, debug, out, nullptr);
    if (rc) return rc;
    TQ_HIP(ctx, hipDeviceSynchronize());
    TQ_HIP(ctx, hipMemcpy(rstat, out.rstat, (size_t)Q * 8, hipMemcpyDeviceToHost));
    TQ_HIP(ctx, hipMemcpy(rscor, out.rscor, (size_t)Q * 24, hipMemcpyDeviceToHost));
    if (flags) TQ_HIP(ctx, hipMemcpy(flags, out.flags, (size_t)Q, hipMemcpyDeviceToHost));
    if (cmats) TQ_HIP(ctx, hipMemcpy(cmats, out.cmats, (size_t)Q * 3072, hipMemcpyDeviceToHost));
    if (svds) TQ_HIP(ctx, hipMemcpy(svds, out.svds, (size_t)Q * 384, hipMemcpyDeviceToHost));
    if (ranks) TQ_HIP(ctx, hipMemcpy(ranks, out.ranks, (size_t)Q * 12, hipMemcpyDeviceToHost));
    return TQ_OK;
}

int tq_resolve(tq_ctx *ctx, const uint32_t *quartets, int64_t Q, int subsample, uint32_t *rstat, double *rscor,
               uint8_t *flags)
{
    return tq_resolve_debug(ctx, quartets, Q, subsample, rstat, rscor, flags, nullptr, nullptr, nullptr);
}

int tq_timing_enable(tq_ctx *ctx, int on)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    ctx->timing = on != 0;
    return TQ_OK;
}

int tq_timing_read(tq_ctx *ctx, double *kernel_ms, int64_t *launches)
{
    return tq_timing_read_split(ctx, kernel_ms, nullptr, nullptr, launches);
}

int tq_timing_read_split(tq_ctx *ctx, double *total_ms, double *scan_ms, double *svd_ms, int64_t *calls)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    double t_scan = 0.0, t_svd = 0.0;
    for (size_t i = 0; i < ctx->events_used; ++i) {
        TQ_HIP(ctx, hipEventSynchronize(ctx->events[i].e2));
        float a = 0.f, b = 0.f;
        TQ_HIP(ctx, hipEventElapsedTime(&a, ctx->events[i].e0, ctx->events[i].e1));
        TQ_HIP(ctx, hipEventElapsedTime(&b, ctx->events[i].e1, ctx->events[i].e2));
        t_scan += a;
        t_svd += b;
    }
    if (total_ms) *total_ms = t_scan + t_svd;
    if (scan_ms) *scan_ms = t_scan;
    if (svd_ms) *svd_ms = t_svd;
    if (calls) *calls = ctx->timed_calls;
    ctx->events_used = 0;
    ctx->timed_calls = 0;
    return TQ_OK;
}

int tq_set_option(tq_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return TQ_ERR_INVALID_ARG;
    if (!strcmp(name, "nrep")) {
        if (value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || value == 32) ctx->nrep = (int)value;
        else if (value != 0) return fail(ctx, TQ_ERR_INVALID_ARG, "nrep must be 1,2,4,8,16 or 32");
        else ctx->nrep = 1;
        return TQ_OK;
    }
    if (!strcmp(name, "waves_per_cu")) {
        if (value < 0 || value > 32) return fail(ctx, TQ_ERR_INVALID_ARG, "waves_per_cu must be 0..32");
        ctx->waves_per_cu = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "overlap")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "overlap must be >= 0");
        ctx->overlap = value;
        return TQ_OK;
    }
    if (!strcmp(name, "ov_scan_wgs")) {
        if (value < 1 || value > 4) return fail(ctx, TQ_ERR_INVALID_ARG, "ov_scan_wgs must be 1..4");
        ctx->ov_scan_wgs = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "ov_svd_waves")) {
        if (value < 1 || value > 32) return fail(ctx, TQ_ERR_INVALID_ARG, "ov_svd_waves must be 1..32");
        ctx->ov_svd_waves = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_wg")) {
        if (value != 0 && value != 8) return fail(ctx, TQ_ERR_INVALID_ARG, "scan_wg must be 0 or 8");
        ctx->scan_wg = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "svd_method")) {
        if (value != 0 && value != 1) return fail(ctx, TQ_ERR_INVALID_ARG, "svd_method must be 0 (Jacobi) or 1 (HQR)");
        ctx->svd_method = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "order")) {
        if (value != 0 && value != 1) return fail(ctx, TQ_ERR_INVALID_ARG, "order must be 0 or 1");
        ctx->order = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_method")) {
        if (value != 0 && value != 1 && value != -1 && value != 2)
            return fail(ctx, TQ_ERR_INVALID_ARG, "scan_method must be -1 (auto), 0, 1 (or 2: timing diagnostic)");
        ctx->scan_method = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "batch")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "batch must be >= 0");
        ctx->batch = value ? value : (1 << 20);
        return TQ_OK;
    }
    if (!strcmp(name, "phases")) {
        if (value != 0 && value != 1 && value != 2 && value != 3)
            return fail(ctx, TQ_ERR_INVALID_ARG, "phases must be 1, 2 or 3");
        ctx->phases = value ? (int)value : 3;
        return TQ_OK;
    }
    return fail(ctx, TQ_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int tq_set_source(tq_ctx *ctx, const uint8_t *seqarr, int64_t T, int64_t S0, const int64_t *spans, int64_t nloci)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!seqarr || !spans) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_source: NULL pointer");
    if (T < 1 || S0 < 1 || nloci < 1 || T > 0x7FFFFFFF)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_source: bad shape T=%lld S0=%lld nloci=%lld", (long long)T,
                    (long long)S0, (long long)nloci);
    int64_t maxw = 0;
    for (int64_t i = 0; i < nloci; ++i) {
        const int64_t a = spans[2 * i], b = spans[2 * i + 1];
        if (a < 0 || b <= a || b > S0)
            return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_source: span %lld = [%lld,%lld) outside [0,%lld)",
                        (long long)i, (long long)a, (long long)b, (long long)S0);
        if (b - a > maxw) maxw = b - a;
    }
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    free_source(ctx);
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_seqarr, (size_t)(T * S0)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_spans, (size_t)nloci * 16));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_lidxs, (size_t)nloci * 8));
    TQ_HIP(ctx, hipMemcpy(ctx->d_seqarr, seqarr, (size_t)(T * S0), hipMemcpyHostToDevice));
    TQ_HIP(ctx, hipMemcpy(ctx->d_spans, spans, (size_t)nloci * 16, hipMemcpyHostToDevice));
    ctx->src_T = T;
    ctx->src_S0 = S0;
    ctx->nloci = nloci;
    ctx->max_width = maxw;
    return TQ_OK;
}

int tq_bootstrap(tq_ctx *ctx, const int64_t *lidxs, int64_t n, uint64_t seed_shuffle, uint64_t seed_ambig,
                 int64_t *out_S)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->d_seqarr) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_source has not been called");
    if (!lidxs || n != ctx->nloci)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_bootstrap: lidxs must hold nloci=%lld locus indices", (long long)ctx->nloci);
    int64_t S = 0;
    for (int64_t i = 0; i < n; ++i)
        if (lidxs[i] < 0 || lidxs[i] >= ctx->nloci)
            return fail(ctx, TQ_ERR_INVALID_ARG, "tq_bootstrap: locus index %lld out of range", (long long)lidxs[i]);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->src_T;
    // worst-case replicate length; buffers grow only
    const int64_t cap = n * ctx->max_width;
    if (cap > ctx->boot_cap) {
        if (ctx->d_boot) (void)hipFree(ctx->d_boot);
        if (ctx->d_boot_tmp) (void)hipFree(ctx->d_boot_tmp);
        ctx->d_boot = nullptr;
        ctx->d_boot_tmp = nullptr;
        ctx->boot_cap = 0;
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_boot, (size_t)(2 * (n + 1) + 2 * cap) * sizeof(uint32_t)));
        size_t tmp = 0;
        uint32_t *u = ctx->d_boot;
        TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, u, u, (int)(n + 1)));
        size_t tmp2 = 0;
        TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, u, u, (int)(cap / 32 + 2)));
        if (tmp2 > tmp) tmp = tmp2;
        TQ_HIP(ctx, hipMalloc(&ctx->d_boot_tmp, tmp ? tmp : 16));
        ctx->boot_tmp_bytes = tmp;
        ctx->boot_cap = cap;
    }
    uint32_t *widths = ctx->d_boot, *offsets = widths + (n + 1);
    uint32_t *src_col = offsets + (n + 1), *site_locus = src_col + ctx->boot_cap;
    TQ_HIP(ctx, hipMemcpy(ctx->d_lidxs, lidxs, (size_t)n * 8, hipMemcpyHostToDevice));
    TQ_HIP(ctx, hipMemset(widths + n, 0, sizeof(uint32_t)));
    hipLaunchKernelGGL(tq_boot_width_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ctx->d_spans,
                       ctx->d_lidxs, n, ctx->nloci, widths);
    size_t tmp = ctx->boot_tmp_bytes;
    TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(ctx->d_boot_tmp, tmp, widths, offsets, (int)(n + 1)));
    uint32_t total = 0;
    TQ_HIP(ctx, hipMemcpy(&total, offsets + n, sizeof(uint32_t), hipMemcpyDeviceToHost));
    S = (int64_t)total;
    if (S < 1 || S > ctx->boot_cap) return fail(ctx, TQ_ERR_HIP, "tq_bootstrap: inconsistent replicate length %lld", (long long)S);
    const int64_t Sp = (int64_t)align_up((size_t)S, TILE);
    const int64_t W = Sp / 32;
    if (Sp > ctx->data_capacity || T != ctx->T) {
        free_data(ctx);
        const int64_t capSp = (int64_t)align_up((size_t)(Sp + Sp / 8), TILE);   // head-room: replicate lengths vary
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)(T * capSp)));
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)(T * (capSp / 32)) * sizeof(uint4)));
        ctx->data_capacity = capSp;
    }
    hipLaunchKernelGGL(tq_boot_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ctx->d_spans,
                       ctx->d_lidxs, offsets, n, ctx->nloci, seed_shuffle, src_col, site_locus);
    const int64_t nw = T * W;
    hipLaunchKernelGGL(tq_boot_build_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, ctx->d_seqarr,
                       ctx->src_S0, src_col, site_locus, S, Sp, W, (int32_t)T, seed_ambig, ctx->d_rows, ctx->d_planes);
    TQ_HIP(ctx, hipGetLastError());
    TQ_HIP(ctx, hipDeviceSynchronize());
    ctx->T = T;
    ctx->S = S;
    ctx->Sp = Sp;
    ctx->W = W;
    ctx->have_data = true;
    ctx->locus_runs_ok = true;          // locus ids are the ordinals 0..n-1, one run each
    if (out_S) *out_S = S;
    return TQ_OK;
}

int tq_get_data(tq_ctx *ctx, uint8_t *tmparr, uint32_t *tmpmap)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "no replicate on the device");
    if (!tmparr || !tmpmap) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_get_data: NULL pointer");
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->T, S = ctx->S, W = ctx->W;
    const size_t bytes = align_up((size_t)(T * S), 256) + align_up((size_t)S * 8, 256) + align_up((size_t)(W + 1) * 8, 256);
    int rc = ensure_scratch(ctx, bytes);
    if (rc) return rc;
    uint8_t *d_arr = (uint8_t *)ctx->d_scratch;
    uint32_t *d_map = (uint32_t *)((char *)ctx->d_scratch + align_up((size_t)(T * S), 256));
    uint32_t *d_cnt = (uint32_t *)((char *)d_map + align_up((size_t)S * 8, 256));
    uint32_t *d_base = d_cnt + (W + 1);
    hipLaunchKernelGGL(tq_export_kernel, dim3((unsigned)((T * S + 255) / 256)), dim3(256), 0, 0, ctx->d_rows,
                       ctx->d_planes, S, ctx->Sp, W, (int32_t)T, d_arr, d_map);
    hipLaunchKernelGGL(tq_export_runcount_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, 0, ctx->d_planes, W,
                       d_cnt);
    size_t tmp = 0;
    TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, d_cnt, d_base, (int)W));
    void *d_tmp = nullptr;
    TQ_HIP(ctx, hipMalloc(&d_tmp, tmp ? tmp : 16));
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp, d_cnt, d_base, (int)W);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tq_export_locus_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, 0, ctx->d_planes,
                           (const uint32_t *)d_base, S, W, d_map);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(tmparr, d_arr, (size_t)(T * S), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tmpmap, d_map, (size_t)S * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(ctx, TQ_ERR_HIP, "tq_get_data: %s", hipGetErrorString(e));
    return TQ_OK;
}

int tq_data_shape(tq_ctx *ctx, int64_t *T, int64_t *S)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (T) *T = ctx->have_data ? ctx->T : 0;
    if (S) *S = ctx->have_data ? ctx->S : 0;
    return TQ_OK;
}

int tq_device_info(tq_ctx *ctx, int32_t *num_cu, int32_t *waves_per_cu, int64_t *row_pitch)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (num_cu) *num_cu = ctx->prop.multiProcessorCount;
    if (waves_per_cu) *waves_per_cu = ctx->waves_per_cu;
    if (row_pitch) *row_pitch = ctx->Sp;
    return TQ_OK;
}

}  // extern "C"
