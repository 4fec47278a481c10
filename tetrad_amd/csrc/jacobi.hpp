// jacobi.hpp -- one-sided Jacobi singular values in registers + scores (tq_svd_kernel)
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
#pragma once

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
    bool noconv;    // the sweep cap was reached with rotations still pending
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
    o.noconv = active;      // the loop ends early (break) once no group of the wave is active
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
        bool noconv = false;
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
            noconv |= sv.noconv;
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
            if (noconv) fl |= TQ_FLAG_NO_CONVERGENCE;
            if (my_nsnps == 0) {                 // resolve_quartets.py:230-232
                topo = 0;
                sc[0] = sc[1] = sc[2] = 0.001;
                fl = TQ_FLAG_ZERO_DATA;
            }
            const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[myq];
            const uint32_t Tu = (uint32_t)T;
            if ((qv.x >= Tu) | (qv.y >= Tu) | (qv.z >= Tu) | (qv.w >= Tu)) fl |= TQ_FLAG_BAD_INDEX;
            fl |= out.flag_or;
            out.rstat[myq * 2 + 0] = (uint32_t)topo;
            out.rstat[myq * 2 + 1] = my_nsnps;
            out.rscor[myq * 3 + 0] = sc[0];
            out.rscor[myq * 3 + 1] = sc[1];
            out.rscor[myq * 3 + 2] = sc[2];
            if (out.flags) out.flags[myq] = (uint8_t)fl;
        }
    }
}

