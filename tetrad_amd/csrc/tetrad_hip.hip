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
//   rows   u8  [T][Sp]   base code 0..3 per site, missing/pad -> 0   (Sp = S rounded up to 2048)
//   miss   u32 [T][W]    1 bit per site, 1 = missing or pad           (W  = Sp/32)
//   p0,p1  u32 [T][W]    bit-planes of the base code (subsample mode only)
//   runbeg u32 [W]       1 bit per site, 1 = site starts a new locus run
//
// Kernel tq_resolve_kernel, one wavefront per workgroup, persistent grid:
//   phase 1 (x4 quartets): wave scans a quartet 2048 sites per step.  Lane l owns 32
//     consecutive sites: 2 x dwordx4 per row (coalesced 2 KiB per row per step) plus one
//     bit-plane word per row.  The "count this site" mask C is pure bit logic on 32-site
//     words (missing, variable-among-4, first-unmasked-site-of-locus via an adder carry
//     chain, cross-lane carries resolved on the scalar unit from two ballots).  The 8-bit
//     site pattern (a<<6|b<<4|c<<2|d) is built 4 sites per VALU op (SWAR); uncounted sites
//     are steered to bin 0 (AAAA) -- invariant patterns can never be counted, so the four
//     invariant bins double as dump bins and are cleared afterwards.  Counts go to an LDS
//     histogram with NREP lane-interleaved replicas (ds_add_u32).
//   phase 2: each 16-lane group takes one quartet; lane j holds column j (16 f64) of a
//     flattening and the group runs a one-sided (Hestenes) Jacobi SVD with the XOR-partner
//     parallel ordering (15 rounds per sweep, partner = lane ^ m).  Rank rule, minrank,
//     tail-norm scores and argmin follow resolve_quartets.py:241-251.
//
// Bounds: the scan is L2/LDS/VALU work on a <= 26 MB matrix (HBM only on first touch), the
// SVD is f64 VALU.  Algorithmic bytes per quartet are 4*S + 48.
#include <hip/hip_runtime.h>

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
constexpr int QPW = 4;                              // quartets per wave pass (one per 16-lane group)
constexpr double F64_EPS = 2.220446049250313e-16;
constexpr double JTOL2 = 7.888609052210118e-31;     // (2^-50)^2 : rotate while g^2 > JTOL2*a*b
constexpr double JEARLY2 = 1e-10;                   // (1e-5)^2 : see jacobi16
constexpr int MAX_SWEEPS = 30;
constexpr double DEGENERATE_REL_GAP = 1e-9;

struct DevData {
    const uint8_t *rows;
    const uint32_t *miss, *p0, *p1, *runbeg;
    int64_t pitch;      // bytes per row (Sp)
    int64_t W;          // bit-plane words per row (Sp/32)
    int32_t T;
    int32_t ntiles;     // Sp / TILE
};

struct OutPtrs {
    uint32_t *rstat;    // [Q,2]
    double *rscor;      // [Q,3]
    uint8_t *flags;     // [Q] or null
    uint32_t *cmats;    // [Q,3,16,16] or null (debug)
    double *svds;       // [Q,3,16] or null (debug)
    int32_t *ranks;     // [Q,3] or null (debug)
};

// ------------------------------------------------------------------------------------
// data preparation kernels
// ------------------------------------------------------------------------------------
// one thread per 32-site word of one taxon row
__global__ void tq_prepare_rows(const uint8_t *__restrict__ raw, int64_t S, int64_t Sp, int64_t W,
                                int32_t T, uint8_t *__restrict__ rows, uint32_t *__restrict__ miss,
                                uint32_t *__restrict__ p0, uint32_t *__restrict__ p1)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T * W) return;
    int64_t t = gid / W, w = gid - t * W;
    const uint8_t *src = raw + t * S + w * 32;
    uint8_t *dst = rows + t * Sp + w * 32;
    uint32_t mm = 0, b0 = 0, b1 = 0;
    for (int i = 0; i < 32; ++i) {
        int64_t s = w * 32 + i;
        uint8_t v = (s < S) ? src[i] : (uint8_t)0xFF;
        bool missing = v > 3;
        uint8_t code = missing ? (uint8_t)0 : v;
        dst[i] = code;
        mm |= (uint32_t)missing << i;
        b0 |= (uint32_t)(code & 1) << i;
        b1 |= (uint32_t)((code >> 1) & 1) << i;
    }
    miss[t * W + w] = mm;
    p0[t * W + w] = b0;
    p1[t * W + w] = b1;
}

// one thread per 32-site word: run-begin bits of the locus column
__global__ void tq_prepare_runbeg(const uint32_t *__restrict__ locus, int64_t S, int64_t W,
                                  uint32_t *__restrict__ runbeg)
{
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    uint32_t bits = 0;
    for (int i = 0; i < 32; ++i) {
        int64_t s = w * 32 + i;
        if (s < S) {
            bool beg = (s == 0) || (locus[s] != locus[s - 1]);
            bits |= (uint32_t)beg << i;
        }
    }
    runbeg[w] = bits;
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

// ------------------------------------------------------------------------------------
// phase 1: site scan -> 256-bin pattern histogram in LDS
// ------------------------------------------------------------------------------------
struct TileRegs {
    uint4 a0, a1, b0, b1, c0, c1, d0, d1;   // 32 site bytes of each of the four rows
    uint32_t M;                              // OR of the four missing words
    uint32_t V;                              // variable-among-the-four bits (subsample only)
    uint32_t B;                              // run-begin bits (subsample only)
};

template <bool SUB>
__device__ __forceinline__ void load_tile(TileRegs &r, const DevData &d, const uint32_t (&q)[4], int tile,
                                          int lane)
{
    const int64_t boff = (int64_t)tile * TILE + lane * SITES_PER_LANE;
    const uint4 *pa = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[0] * d.pitch + boff);
    const uint4 *pb = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[1] * d.pitch + boff);
    const uint4 *pc = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[2] * d.pitch + boff);
    const uint4 *pd = reinterpret_cast<const uint4 *>(d.rows + (int64_t)q[3] * d.pitch + boff);
    r.a0 = pa[0]; r.a1 = pa[1];
    r.b0 = pb[0]; r.b1 = pb[1];
    r.c0 = pc[0]; r.c1 = pc[1];
    r.d0 = pd[0]; r.d1 = pd[1];
    const int64_t woff = (int64_t)tile * WAVE + lane;
    const int64_t wa = (int64_t)q[0] * d.W + woff, wb = (int64_t)q[1] * d.W + woff;
    const int64_t wc = (int64_t)q[2] * d.W + woff, wd = (int64_t)q[3] * d.W + woff;
    r.M = d.miss[wa] | d.miss[wb] | d.miss[wc] | d.miss[wd];
    if (SUB) {
        uint32_t a0 = d.p0[wa], a1 = d.p1[wa];
        r.V = (a0 ^ d.p0[wb]) | (a1 ^ d.p1[wb]) | (a0 ^ d.p0[wc]) | (a1 ^ d.p1[wc]) |
              (a0 ^ d.p0[wd]) | (a1 ^ d.p1[wd]);
        r.B = d.runbeg[woff];
    } else {
        r.V = 0;
        r.B = 0;
    }
}

// four sites (one dword of each row) -> four histogram increments
template <int NREP>
__device__ __forceinline__ void hist_dword(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t nib,
                                           uint32_t *hrep)
{
    // base codes are 0..3, so the per-byte pattern (a<<6|b<<4|c<<2|d) never crosses a byte
    uint32_t pat = (((((a << 2) | b) << 2) | c) << 2) | d;
    // 4 count bits -> 4 byte masks (bit k -> byte k)
    uint32_t e = __umul24(nib, 0x204081u) & 0x01010101u;
    uint32_t keep = (e << 8) - e;
    uint32_t idx = pat & keep;               // uncounted sites -> bin 0 (an invariant bin)
    __hip_atomic_fetch_add(&hrep[(idx & 0xFFu) * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&hrep[((idx >> 8) & 0xFFu) * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&hrep[((idx >> 16) & 0xFFu) * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(&hrep[(idx >> 24) * NREP], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Sites to count among this lane's 32 (bit i = site i).
//   full mode      : every non-missing site; invariant sites land in the invariant bins
//                    (resolve_quartets.py:216-218 masks them; the bins are cleared later).
//   subsample mode : unmasked sites that are the first unmasked site of their locus run
//                    (resolve_quartets.py:58-64).  seen(i) = "an unmasked site precedes i in
//                    the same run" obeys t(i) = U(i) | (P(i) & t(i-1)) with P = ~runbegin,
//                    which is the carry recurrence of the addition (U|P) + U.
template <bool SUB>
__device__ __forceinline__ uint32_t count_mask(const TileRegs &r, int lane, uint32_t &tile_carry)
{
    if (!SUB) return ~r.M;
    const uint32_t U = r.V & ~r.M;
    const uint32_t P = ~r.B;
    const uint32_t X = U | P;
    const uint32_t sum = X + U;
    const uint32_t cin0 = sum ^ X ^ U;                       // carry into each bit, lane carry-in = 0
    const uint32_t seen_local = P & cin0;
    const uint32_t gen = ((X & U) | ((X | U) & ~sum)) >> 31; // carry out of bit 31 = t(31)
    // cross-lane: T(l) = gen(l) | (allprop(l) & T(l-1)); same adder trick on 64-bit ballots (SALU)
    const uint64_t Gm = __ballot(gen != 0);
    const uint64_t Pm = __ballot(r.B == 0);
    const uint64_t Xm = Gm | Pm;
    const uint64_t s1 = Xm + Gm;
    const uint64_t s2 = s1 + (uint64_t)tile_carry;
    const uint64_t cinm = s2 ^ Xm ^ Gm;                      // carry into each lane
    const uint32_t cout = (uint32_t)((s1 < Xm) | (s2 < s1)); // carry out of lane 63
    tile_carry = cout;
    const uint32_t cin = (uint32_t)(cinm >> lane) & 1u;
    // sites before this lane's first run-begin inherit the incoming "seen" state
    const uint32_t firstseg = r.B ? ((r.B & (0u - r.B)) - 1u) : 0xFFFFFFFFu;
    const uint32_t seen = seen_local | (cin ? firstseg : 0u);
    return U & ~seen;
}

template <int NREP, bool SUB>
__device__ __forceinline__ void scan_quartet(const DevData &d, const uint32_t (&q)[4], uint32_t *hist,
                                             int lane)
{
    uint32_t *hrep = hist + (lane & (NREP - 1));
    uint32_t tile_carry = 0;
    TileRegs cur, nxt;
    load_tile<SUB>(cur, d, q, 0, lane);
    for (int t = 0; t < d.ntiles; ++t) {
        if (t + 1 < d.ntiles) load_tile<SUB>(nxt, d, q, t + 1, lane);
        const uint32_t C = count_mask<SUB>(cur, lane, tile_carry);
        hist_dword<NREP>(cur.a0.x, cur.b0.x, cur.c0.x, cur.d0.x, (C >> 0) & 15u, hrep);
        hist_dword<NREP>(cur.a0.y, cur.b0.y, cur.c0.y, cur.d0.y, (C >> 4) & 15u, hrep);
        hist_dword<NREP>(cur.a0.z, cur.b0.z, cur.c0.z, cur.d0.z, (C >> 8) & 15u, hrep);
        hist_dword<NREP>(cur.a0.w, cur.b0.w, cur.c0.w, cur.d0.w, (C >> 12) & 15u, hrep);
        hist_dword<NREP>(cur.a1.x, cur.b1.x, cur.c1.x, cur.d1.x, (C >> 16) & 15u, hrep);
        hist_dword<NREP>(cur.a1.y, cur.b1.y, cur.c1.y, cur.d1.y, (C >> 20) & 15u, hrep);
        hist_dword<NREP>(cur.a1.z, cur.b1.z, cur.c1.z, cur.d1.z, (C >> 24) & 15u, hrep);
        hist_dword<NREP>(cur.a1.w, cur.b1.w, cur.c1.w, cur.d1.w, (C >> 28) & 15u, hrep);
        cur = nxt;
    }
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}

// fold the NREP replicas into cm[256], clear the histogram for the next quartet, clear the
// four invariant/dump bins (AAAA, CCCC, GGGG, TTTT) and return the number of counted sites
template <int NREP>
__device__ __forceinline__ uint32_t fold_hist(uint32_t *hist, uint32_t *cm, int lane)
{
    uint32_t tot = 0;
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
        if (bin % 85 == 0) s = 0;            // 0, 85, 170, 255
        cm[bin] = s;
        tot += s;
    }
    return wave_sum_u32(tot);
}

// ------------------------------------------------------------------------------------
// phase 2: singular values of a 16x16 matrix, one column per lane of a 16-lane group
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

// PHASES: 3 = product kernel; 1 = scan only, 2 = singular values only (timing diagnostics, outputs
// are not meaningful: selected with tq_set_option("phases", ...), never by the product path)
template <int NREP, bool SUB, bool DEBUG, int PHASES = 3>
__global__ void __launch_bounds__(WAVE)
tq_resolve_kernel(DevData d, const uint32_t *__restrict__ quartets, int64_t Q, OutPtrs out)
{
    __shared__ uint32_t lds[256 * NREP + QPW * 256];
    uint32_t *hist = lds;
    uint32_t *cm = lds + 256 * NREP;
    const int lane = threadIdx.x;
    const int grp = lane >> 4;      // 16-lane group = quartet slot in phase 2
    const int j = lane & 15;        // column owned in phase 2

    for (int i = lane; i < 256 * NREP; i += WAVE) hist[i] = 0;
    __syncthreads();

    const int64_t ngroups = (Q + QPW - 1) / QPW;
    for (int64_t wg = blockIdx.x; wg < ngroups; wg += gridDim.x) {
        uint32_t my_nsnps = 0;
        uint32_t my_bad = 0;
        // ---------------- phase 1: four site scans ----------------
#pragma unroll 1
        for (int i = 0; i < QPW; ++i) {
            const int64_t qi = wg * QPW + i;
            uint32_t nsn = 0, bad = 0;
            uint32_t *cmi = cm + 256 * i;
            if (qi < Q) {
                const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[qi];
                uint32_t q[4];
                q[0] = __builtin_amdgcn_readfirstlane(qv.x);
                q[1] = __builtin_amdgcn_readfirstlane(qv.y);
                q[2] = __builtin_amdgcn_readfirstlane(qv.z);
                q[3] = __builtin_amdgcn_readfirstlane(qv.w);
                const uint32_t T = (uint32_t)d.T;
                bad = (q[0] >= T) | (q[1] >= T) | (q[2] >= T) | (q[3] >= T);
                if (!bad) {
                    if (PHASES & 1) {
                        scan_quartet<NREP, SUB>(d, q, hist, lane);
                        __syncthreads();
                        nsn = fold_hist<NREP>(hist, cmi, lane);
                    } else {
                        // synthetic full-rank counts so the SVD stage does representative work
                        for (int k = lane; k < 256; k += WAVE)
                            cmi[k] = (k % 85 == 0) ? 0u : ((q[0] * 131u + q[1] * 71u + q[2] * 31u + q[3] * 7u + k * 2654435761u) >> 20) % 997u;
                        nsn = 1;
                    }
                }
            }
            if ((qi >= Q) | bad) {
                for (int k = lane; k < 256; k += WAVE) cmi[k] = 0;
            }
            if (grp == i) {
                my_nsnps = nsn;
                my_bad = bad;
            }
        }
        __syncthreads();

        // ---------------- phase 2: three flattenings per 16-lane group ----------------
        const int64_t myq = wg * QPW + grp;
        const uint32_t *cmq = cm + 256 * grp;
        if (!(PHASES & 2)) {
            if (j == 0 && myq < Q) {
                out.rstat[myq * 2 + 0] = cmq[1];
                out.rstat[myq * 2 + 1] = my_nsnps;
            }
            __syncthreads();
            continue;
        }
        double sig[3];
        int pos[3], rnk[3];
        double smax_all = 0.0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            double a[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t v = cmq[flat_bin(t, r, j)];
                a[r] = (double)v;
                if (DEBUG) {
                    if (out.cmats && myq < Q) out.cmats[((myq * 3 + t) * 16 + r) * 16 + j] = v;
                }
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
        __syncthreads();   // all reads of cm done before the next pass overwrites it

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
            if (my_bad) fl |= TQ_FLAG_BAD_INDEX;
            out.rstat[myq * 2 + 0] = (uint32_t)topo;
            out.rstat[myq * 2 + 1] = my_nsnps;
            out.rscor[myq * 3 + 0] = sc[0];
            out.rscor[myq * 3 + 1] = sc[1];
            out.rscor[myq * 3 + 2] = sc[2];
            if (out.flags) out.flags[myq] = (uint8_t)fl;
        }
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
    uint32_t *d_planes = nullptr;   // miss | p0 | p1, each T*W words
    uint32_t *d_runbeg = nullptr;
    bool have_data = false;
    bool locus_runs_ok = false;
    // scratch for the host-buffer API
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // options
    int nrep = 8;
    int waves_per_cu = 0;           // 0 = from the occupancy query
    int phases = 3;                 // diagnostics only: 1 = scan only, 2 = SVD only
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
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
    if (ctx->d_runbeg) (void)hipFree(ctx->d_runbeg);
    ctx->d_rows = nullptr;
    ctx->d_planes = nullptr;
    ctx->d_runbeg = nullptr;
    ctx->have_data = false;
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

DevData dev_data(const tq_ctx *ctx)
{
    DevData d;
    d.rows = ctx->d_rows;
    d.miss = ctx->d_planes;
    d.p0 = ctx->d_planes + ctx->T * ctx->W;
    d.p1 = ctx->d_planes + 2 * ctx->T * ctx->W;
    d.runbeg = ctx->d_runbeg;
    d.pitch = ctx->Sp;
    d.W = ctx->W;
    d.T = (int32_t)ctx->T;
    d.ntiles = (int32_t)(ctx->Sp / TILE);
    return d;
}

template <int NREP, bool SUB, bool DEBUG, int PHASES = 3>
int launch_t(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, const OutPtrs &out, hipStream_t stream)
{
    auto kern = tq_resolve_kernel<NREP, SUB, DEBUG, PHASES>;
    int wpc = ctx->waves_per_cu;
    if (wpc <= 0) {
        int nb = 0;
        TQ_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, WAVE, 0));
        wpc = nb > 0 ? nb : 8;
    }
    const int64_t ngroups = (Q + QPW - 1) / QPW;
    int64_t grid = (int64_t)ctx->prop.multiProcessorCount * wpc;
    if (grid > ngroups) grid = ngroups;
    if (grid < 1) grid = 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->timing) {
        if (ctx->events_used == ctx->events.size()) {
            hipEvent_t a, b;
            TQ_HIP(ctx, hipEventCreate(&a));
            TQ_HIP(ctx, hipEventCreate(&b));
            ctx->events.emplace_back(a, b);
        }
        e0 = ctx->events[ctx->events_used].first;
        e1 = ctx->events[ctx->events_used].second;
        ctx->events_used++;
        TQ_HIP(ctx, hipEventRecord(e0, stream));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVE), 0, stream, dev_data(ctx), d_quartets, Q, out);
    TQ_HIP(ctx, hipGetLastError());
    if (ctx->timing) TQ_HIP(ctx, hipEventRecord(e1, stream));
    return TQ_OK;
}

template <int NREP>
int launch_n(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, bool debug, const OutPtrs &out,
             hipStream_t stream)
{
    if (debug) {
        return subsample ? launch_t<NREP, true, true>(ctx, dq, Q, out, stream)
                         : launch_t<NREP, false, true>(ctx, dq, Q, out, stream);
    }
    if (ctx->phases == 1)
        return subsample ? launch_t<NREP, true, false, 1>(ctx, dq, Q, out, stream)
                         : launch_t<NREP, false, false, 1>(ctx, dq, Q, out, stream);
    if (ctx->phases == 2)
        return subsample ? launch_t<NREP, true, false, 2>(ctx, dq, Q, out, stream)
                         : launch_t<NREP, false, false, 2>(ctx, dq, Q, out, stream);
    return subsample ? launch_t<NREP, true, false>(ctx, dq, Q, out, stream)
                     : launch_t<NREP, false, false>(ctx, dq, Q, out, stream);
}

int launch(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, bool debug, const OutPtrs &out,
           hipStream_t stream)
{
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (subsample && !ctx->locus_runs_ok)
        return fail(ctx, TQ_ERR_LOCUS_ORDER,
                    "subsample mode needs each locus id in one contiguous run of sites (and no id 0xFFFFFFFF)");
    if (Q == 0) return TQ_OK;
    switch (ctx->nrep) {
    case 1: return launch_n<1>(ctx, dq, Q, subsample, debug, out, stream);
    case 2: return launch_n<2>(ctx, dq, Q, subsample, debug, out, stream);
    case 4: return launch_n<4>(ctx, dq, Q, subsample, debug, out, stream);
    case 16: return launch_n<16>(ctx, dq, Q, subsample, debug, out, stream);
    case 32: return launch_n<32>(ctx, dq, Q, subsample, debug, out, stream);
    default: return launch_n<8>(ctx, dq, Q, subsample, debug, out, stream);
    }
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
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    for (auto &p : ctx->events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
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
    uint8_t *d_raw = nullptr;
    uint32_t *d_loc = nullptr;
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)(T * Sp)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)(3 * T * W) * sizeof(uint32_t)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_runbeg, (size_t)W * sizeof(uint32_t)));
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
        hipLaunchKernelGGL(tq_prepare_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_raw, S, Sp, W,
                           (int32_t)T, ctx->d_rows, ctx->d_planes, ctx->d_planes + T * W,
                           ctx->d_planes + 2 * T * W);
        hipLaunchKernelGGL(tq_prepare_runbeg, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, 0, d_loc, S, W,
                           ctx->d_runbeg);
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
    if (!ctx) return TQ_ERR_INVALID_ARG;
    double total = 0.0;
    for (size_t i = 0; i < ctx->events_used; ++i) {
        TQ_HIP(ctx, hipEventSynchronize(ctx->events[i].second));
        float ms = 0.f;
        TQ_HIP(ctx, hipEventElapsedTime(&ms, ctx->events[i].first, ctx->events[i].second));
        total += ms;
    }
    if (kernel_ms) *kernel_ms = total;
    if (launches) *launches = (int64_t)ctx->events_used;
    ctx->events_used = 0;
    return TQ_OK;
}

int tq_set_option(tq_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return TQ_ERR_INVALID_ARG;
    if (!strcmp(name, "nrep")) {
        if (value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || value == 32) ctx->nrep = (int)value;
        else if (value != 0) return fail(ctx, TQ_ERR_INVALID_ARG, "nrep must be 1,2,4,8,16 or 32");
        else ctx->nrep = 8;
        return ctx->nrep;
    }
    if (!strcmp(name, "waves_per_cu")) {
        if (value < 0 || value > 32) return fail(ctx, TQ_ERR_INVALID_ARG, "waves_per_cu must be 0..32");
        ctx->waves_per_cu = (int)value;
        return ctx->waves_per_cu;
    }
    if (!strcmp(name, "phases")) {
        if (value != 0 && value != 1 && value != 2 && value != 3)
            return fail(ctx, TQ_ERR_INVALID_ARG, "phases must be 1, 2 or 3");
        ctx->phases = value ? (int)value : 3;
        return ctx->phases;
    }
    return fail(ctx, TQ_ERR_INVALID_ARG, "unknown option '%s'", name);
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
