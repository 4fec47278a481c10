// hqr.hpp -- Householder bidiagonalisation + bidiagonal QR + score kernel (default singular-value path)
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
#pragma once

// ====================================================================================
// Default singular-value path ("HQR"): Householder bidiagonalisation + implicit-shift QR
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
// The Jacobi kernel (jacobi.hpp) stays selectable (tq_set_option "svd_method" 0) and is the
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
    // v_rsq_f64 (5e-8) + one Newton step (4e-15), then one Heron step on the root itself, which squares that
    // error: full precision without a second Newton step on the reciprocal root
    const double y = rsq_nr<1>(x);
    const double s = x * y;
    const double r = fma(fma(-s, s, x), 0.5 * y, s);
    return x > 0.0 ? r : 0.0;
}

// de layout: f64 [3*Q][32]: d[0..15] then e[0..15] with e[0] = 0 (e[i] couples columns i-1, i)
template <bool DEBUG>
__global__ void __launch_bounds__(WAVE, 2)
tq_bidiag_kernel(const uint32_t *__restrict__ cm, int64_t Q, double *__restrict__ de,
                 uint32_t *__restrict__ nsnps_out, uint32_t *__restrict__ cmats_dbg, int tsplit)
{
    constexpr int QP = 16;                       // quartets per wave pass (one per quad)
    // a quartet's 256 counts sit 260 dwords apart: the 16 quads read the same bin of 16 different quartets in every
    // ds_read, and with a pitch of 256 all of them hit one bank (SQ_LDS_BANK_CONFLICT 85 % of the LDS-busy cycles);
    // with 4 dwords of padding quad q is displaced by 4q banks, which is conflict-free for flattenings 0 and 1
    // (a quad's four lanes read 4 consecutive bins) and 4-way for flattening 2 (bins 4 apart)
    constexpr int QPITCH = 260;
    __shared__ uint32_t lds[QP * QPITCH];
    const int lane = threadIdx.x;
    const int quad = lane >> 2;
    const int c = lane & 3;

    // tsplit = 1 (small batches): the three flattenings of a pass go to three blocks instead of one after the other --
    // a call of a few thousand quartets (the reference's chunk sizes, run_inference.py:73-96) does not fill the chip
    // and is bound by the latency of its kernels: 33 -> 13 us for 1 000 quartets.  Same arithmetic per matrix.
    const int64_t npass = (Q + QP - 1) / QP;
    const int64_t nitem = tsplit ? 3 * npass : npass;
    for (int64_t item = blockIdx.x; item < nitem; item += gridDim.x) {
        const int64_t wg = tsplit ? item / 3 : item;
        const int t_lo = tsplit ? (int)(item - 3 * wg) : 0, t_hi = tsplit ? t_lo + 1 : 3;
        const int64_t q0 = wg * QP;
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(cm + q0 * 256);
            uint4 *dst = reinterpret_cast<uint4 *>(lds);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int idx = lane + WAVE * k;               // 1024 uint4 = 16 quartets x 64: quartet k, piece lane
                const bool ok = (q0 + k) < Q;
                dst[k * (QPITCH / 4) + lane] = ok ? src[idx] : make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
        const int64_t myq = q0 + quad;
        const uint32_t *cmq = lds + QPITCH * quad;
        // resolve_quartets.py:226: number of counted sites = sum of the count tensor
        {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < 64; ++k) s += cmq[4 * k + c];
            s += __shfl_xor(s, 1, WAVE);
            s += __shfl_xor(s, 2, WAVE);
            if (c == 0 && myq < Q && t_lo == 0) nsnps_out[myq] = s;
        }
        double thr2 = 0.0;                                      // set at the first t: ||M_t||_F is the same for all three
#pragma unroll 1
        for (int t = t_lo; t < t_hi; ++t) {
            double a[4][16];                                    // a[s][r] = M_t[r][4s + c]
            // bin of M_t[r][4s + c], r = 4 rh + rl: 64 rh + RL rl + SS s + CS c with (RL, SS, CS) = (16, 4, 1), (4, 16, 1),
            // (1, 16, 4) for t = 0, 1, 2 (flat_bin).  t is a runtime (scalar) value, so the three strides are SGPRs: 16
            // per-lane addresses (one per (rl, s)) and the four rh of each as compile-time LDS offsets -- with flat_bin(t, ..)
            // evaluated per element every load paid two selects and two adds for its index (6 % of the kernel)
            {
                const int RL = t == 0 ? 16 : (t == 1 ? 4 : 1), SS = t == 0 ? 4 : 16, CS = t == 2 ? 4 : 1;
                const uint32_t *base = cmq + CS * c;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int rl = 0; rl < 4; ++rl) {
                        const uint32_t *pp = base + RL * rl + SS * s;
#pragma unroll
                        for (int rh = 0; rh < 4; ++rh) {
                            const int r = 4 * rh + rl;
                            const uint32_t v = pp[64 * rh];
                            a[s][r] = (double)v;
                            if (DEBUG) {
                                if (cmats_dbg && myq < Q) cmats_dbg[((myq * 3 + t) * 16 + r) * 16 + 4 * s + c] = v;
                            }
                        }
                    }
                }
            }
            // Columns / rows whose norm is below 1e-20 * ||M||_F are treated as exactly zero (no
            // reflector, d or e = 0).  Without this a zero column that picked up rounding residue
            // (~1e-17) is "reflected", the next one shrinks to ~1e-33, ... and after ten such
            // columns the squared norms reach the denormal range, 1/den overflows and the rest of
            // the matrix is destroyed (found by the sparse-data stress test against the Jacobi path).
            // (the three flattenings hold the same 256 counts, so their Frobenius norms agree -- exactly: the squares
            // are integers below 2^53, every summation order gives the same double: once per quartet)
            if (t == t_lo) {
                double f2 = 0.0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) f2 = fma(a[s][r], a[s][r], f2);
                }
                thr2 = quad_sum(f2) * 1e-40;
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
                    const bool live = n2 > thr2;
                    const double nrm = sqrt_nr(n2);
                    const double x0 = v[K];
                    const double alpha = live ? ((x0 < 0.0) ? nrm : -nrm) : 0.0;
                    v[K] = x0 - alpha;
                    const double den = fma(-alpha, x0, n2);      // = |v|^2 / 2 > 0 for a live column
                    // one Newton step (2e-15) is enough: I - beta(1+d) v v^T is the exact reflector times a scaling
                    // of ONE direction by 1+2d -- invertible, so ranks and zero singular values are untouched and the
                    // others move by a relative 4e-15
                    const double beta = live ? rcp_nr<1>(den) : 0.0;
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
                    const bool live = n2 > thr2;
                    const double x0 = quad_bcast<c1>(y[s1]);
                    const double nrm = sqrt_nr(n2);
                    const double alpha = live ? ((x0 < 0.0) ? nrm : -nrm) : 0.0;
                    const double den = fma(-alpha, x0, n2);
                    const double beta = live ? rcp_nr<1>(den) : 0.0;
                    if (c == c1) y[s1] = x0 - alpha;
                    if (writer) dout[16 + K1] = alpha;
                    // -beta * y: one multiply per slot instead of one per row, and the sign goes in here so that the
                    // row update is a plain v_fmac (the two-operand form has no negate modifier: with fma(-tt, y, a)
                    // the compiler spent a v_xor per value on sign flips, 145 per matrix)
                    double yb[4];
#pragma unroll
                    for (int s = sb; s < 4; ++s) yb[s] = -beta * y[s];
#pragma unroll
                    for (int i = K1; i < 16; ++i) {
                        double q = 0.0;
#pragma unroll
                        for (int s = sb; s < 4; ++s) q = fma(yb[s], a[s][i], q);
                        const double tt = quad_sum(q);           // = -beta * (row i . y)
#pragma unroll
                        for (int s = sb; s < 4; ++s) a[s][i] = fma(tt, y[s], a[s][i]);
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

// ------------------------------------------------------------------------------------
// tq_bidiag2_kernel: the same bidiagonalisation with the matrix dealt 2 x 2 over the quad: lane (ri, ci) = 2 ri + ci of a
// quad holds the rows of parity ri and the columns of parity ci (8 x 8 values, a[cs][rs] = M[2 rs + ri][2 cs + ci]).
// In the column layout of tq_bidiag_kernel a left reflector's vector lives in ONE lane (16 - K broadcasts, two DPP moves
// each -- 64-bit DPP does not exist) and every row of a right reflector is summed over all FOUR lanes (two butterflies);
// here both reflectors cost one broadcast per value a lane needs (8 - K/2) and one one-stage butterfly per dot product:
// ~520 instead of ~840 v_mov_b32_dpp and ~130 instead of ~270 butterfly adds per matrix.  Mod-2 interleaving keeps all
// four lanes busy while the active block shrinks.  Same reflectors, same thresholds, same outputs (d, e, nsnps).
// ------------------------------------------------------------------------------------
template <int SRC_CI>
__device__ __forceinline__ double quad_from_ci(double v)          // value of the lane with the same ri and ci = SRC_CI
{
    constexpr int CTRL = SRC_CI | (SRC_CI << 2) | ((2 + SRC_CI) << 4) | ((2 + SRC_CI) << 6);
    const int lo = dpp_mov<CTRL>(__double2loint(v));
    const int hi = dpp_mov<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <int SRC_RI>
__device__ __forceinline__ double quad_from_ri(double v)          // value of the lane with the same ci and ri = SRC_RI
{
    constexpr int CTRL = (2 * SRC_RI) | ((2 * SRC_RI + 1) << 2) | ((2 * SRC_RI) << 4) | ((2 * SRC_RI + 1) << 6);
    const int lo = dpp_mov<CTRL>(__double2loint(v));
    const int hi = dpp_mov<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

template <bool DEBUG>
__global__ void __launch_bounds__(WAVE, 2)
tq_bidiag2_kernel(const uint32_t *__restrict__ cm, int64_t Q, double *__restrict__ de,
                  uint32_t *__restrict__ nsnps_out, uint32_t *__restrict__ cmats_dbg, int tsplit)
{
    constexpr int QP = 16;
    constexpr int QPITCH = 260;
    __shared__ uint32_t lds[QP * QPITCH];
    const int lane = threadIdx.x;
    const int quad = lane >> 2;
    const int ri = (lane >> 1) & 1, ci = lane & 1;

    const int64_t npass = (Q + QP - 1) / QP;
    const int64_t nitem = tsplit ? 3 * npass : npass;
    for (int64_t item = blockIdx.x; item < nitem; item += gridDim.x) {
        const int64_t wg = tsplit ? item / 3 : item;
        const int t_lo = tsplit ? (int)(item - 3 * wg) : 0, t_hi = tsplit ? t_lo + 1 : 3;
        const int64_t q0 = wg * QP;
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(cm + q0 * 256);
            uint4 *dst = reinterpret_cast<uint4 *>(lds);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int idx = lane + WAVE * k;
                const bool ok = (q0 + k) < Q;
                dst[k * (QPITCH / 4) + lane] = ok ? src[idx] : make_uint4(0, 0, 0, 0);
            }
        }
        __syncthreads();
        const int64_t myq = q0 + quad;
        const uint32_t *cmq = lds + QPITCH * quad;
        {
            uint32_t s = 0;
            const int c4 = lane & 3;
#pragma unroll
            for (int k = 0; k < 64; ++k) s += cmq[4 * k + c4];
            s += __shfl_xor(s, 1, WAVE);
            s += __shfl_xor(s, 2, WAVE);
            if (c4 == 0 && myq < Q && t_lo == 0) nsnps_out[myq] = s;
        }
        double thr2 = 0.0;
#pragma unroll 1
        for (int t = t_lo; t < t_hi; ++t) {
            double a[8][8];                                     // a[cs][rs] = M_t[2 rs + ri][2 cs + ci]
            // bin of M_t[r][col], r = 4 rh + rl, col = 4 s + c: 64 rh + RL rl + SS s + CS c (see tq_bidiag_kernel); with
            // r = 2 rs + ri, col = 2 cs + ci: rh = rs >> 1, rl = 2 (rs & 1) + ri, s = cs >> 1, c = 2 (cs & 1) + ci
            {
                const int RL = t == 0 ? 16 : (t == 1 ? 4 : 1), SS = t == 0 ? 4 : 16, CS = t == 2 ? 4 : 1;
                const uint32_t *base = cmq + RL * ri + CS * ci;
#pragma unroll
                for (int cs = 0; cs < 8; ++cs) {
#pragma unroll
                    for (int rlo = 0; rlo < 2; ++rlo) {
                        const uint32_t *pp = base + 2 * RL * rlo + SS * (cs >> 1) + 2 * CS * (cs & 1);
#pragma unroll
                        for (int rh = 0; rh < 4; ++rh) {
                            const int rs = 2 * rh + rlo;
                            const uint32_t v = pp[64 * rh];
                            a[cs][rs] = (double)v;
                            if (DEBUG) {
                                if (cmats_dbg && myq < Q)
                                    cmats_dbg[((myq * 3 + t) * 16 + 2 * rs + ri) * 16 + 2 * cs + ci] = v;
                            }
                        }
                    }
                }
            }
            if (t == t_lo) {
                double f2 = 0.0;
#pragma unroll
                for (int cs = 0; cs < 8; ++cs) {
#pragma unroll
                    for (int rs = 0; rs < 8; ++rs) f2 = fma(a[cs][rs], a[cs][rs], f2);
                }
                thr2 = quad_sum(f2) * 1e-40;
            }
            double *dout = de + (myq * 3 + t) * 32;
            const bool writer = ((lane & 3) == 0) && (myq < Q);
            if (writer) dout[16] = 0.0;
            static_for<0, 16>([&](auto kc) {
                constexpr int K = decltype(kc)::value;
                constexpr int PK = K & 1, S0 = K >> 1;           // row / column K: parity PK, slot S0
                // ---- left reflector: zero column K below the diagonal ----
                {
                    // column K's values of this lane's rows: from the lane of the same row parity that holds column K
                    double v[8];
#pragma unroll
                    for (int rs = S0; rs < 8; ++rs) v[rs] = quad_from_ci<PK>(a[S0][rs]);
                    if constexpr (PK == 1) v[S0] = (ri == 1) ? v[S0] : 0.0;     // row 2 S0 = K - 1 is above the diagonal
                    double n2 = 0.0;
#pragma unroll
                    for (int rs = S0; rs < 8; ++rs) n2 = fma(v[rs], v[rs], n2);
                    n2 += dpx<2>(n2);                                        // + the rows of the other parity
                    const bool live = n2 > thr2;
                    const double nrm = sqrt_nr(n2);
                    const double x0 = quad_from_ri<PK>(v[S0]);               // M[K][K]
                    const double alpha = live ? ((x0 < 0.0) ? nrm : -nrm) : 0.0;
                    if (ri == PK) v[S0] = x0 - alpha;
                    const double den = fma(-alpha, x0, n2);
                    const double beta = live ? rcp_nr<1>(den) : 0.0;
                    if (writer) dout[K] = alpha;
                    // columns right of K: slot S0 holds column K + 1 for the lanes with ci = 1 when K is even
                    constexpr int CS0 = PK == 0 ? S0 : S0 + 1;
#pragma unroll
                    for (int cs = CS0; cs < 8; ++cs) {
                        double w = 0.0;
#pragma unroll
                        for (int rs = S0; rs < 8; ++rs) w = fma(v[rs], a[cs][rs], w);
                        w += dpx<2>(w);
                        w *= beta;
                        if (cs == S0) w = (ci == 1) ? w : 0.0;               // K even: column K itself is finished
#pragma unroll
                        for (int rs = S0; rs < 8; ++rs) a[cs][rs] = fma(-w, v[rs], a[cs][rs]);
                    }
                }
                // ---- right reflector: zero row K right of the superdiagonal ----
                if constexpr (K <= 13) {
                    constexpr int K1 = K + 1, P1 = K1 & 1, S1 = K1 >> 1;
                    // row K's values of this lane's columns: from the lane of the same column parity that holds row K
                    double y[8];
                    double p = 0.0;
#pragma unroll
                    for (int cs = S1; cs < 8; ++cs) y[cs] = quad_from_ri<PK>(a[cs][S0]);
                    if constexpr (P1 == 1) y[S1] = (ci == 1) ? y[S1] : 0.0;     // column 2 S1 = K is the diagonal
#pragma unroll
                    for (int cs = S1; cs < 8; ++cs) p = fma(y[cs], y[cs], p);
                    const double n2 = p + dpx<1>(p);
                    const bool live = n2 > thr2;
                    const double x0 = quad_from_ci<P1>(y[S1]);               // M[K][K + 1]
                    const double nrm = sqrt_nr(n2);
                    const double alpha = live ? ((x0 < 0.0) ? nrm : -nrm) : 0.0;
                    const double den = fma(-alpha, x0, n2);
                    const double beta = live ? rcp_nr<1>(den) : 0.0;
                    if (ci == P1) y[S1] = x0 - alpha;
                    if (writer) dout[16 + K1] = alpha;
                    double yb[8];
#pragma unroll
                    for (int cs = S1; cs < 8; ++cs) yb[cs] = -beta * y[cs];
                    // rows below K: slot S1 holds row K + 1 for the lanes with ri = 1 when K + 1 is odd (its ri = 0 row is K)
#pragma unroll
                    for (int rs = S1; rs < 8; ++rs) {
                        double q = 0.0;
#pragma unroll
                        for (int cs = S1; cs < 8; ++cs) q = fma(yb[cs], a[cs][rs], q);
                        double tt = q + dpx<1>(q);                           // = -beta * (row . y)
                        if (P1 == 1 && rs == S1) tt = (ri == 1) ? tt : 0.0;
#pragma unroll
                        for (int cs = S1; cs < 8; ++cs) a[cs][rs] = fma(tt, y[cs], a[cs][rs]);
                    }
                } else if constexpr (K == 14) {
                    const double e15 = quad_bcast<1>(a[7][7]);               // M[14][15]: row parity 0, column parity 1
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
constexpr int RSQ_NR = 1;      // Newton steps on v_rsq_f64 inside a rotation (1: 5e-15, 2: 1e-16 relative)

__global__ void __launch_bounds__(WAVE)
tq_bdsqr_kernel(const double *__restrict__ de, int64_t nmat, double *__restrict__ sv, int maxit,
                unsigned long long *__restrict__ stats, uint32_t *__restrict__ work_out = nullptr)
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
        // The wave's 64 bidiagonals are one contiguous 16 KiB tile of `de`.  It is read as sixteen fully coalesced
        // 1 KiB wave accesses (every line of the tile is touched by exactly one instruction; the lane-per-matrix reads
        // this replaces touched every line with eight, 256 bytes apart per lane, and fetched 1.5x the tile from memory)
        // into the LDS park in matrix-major order -- 16-byte piece c of matrix mm at piece slot mm*16 + (c ^ (mm & 15)),
        // so that both this store and the lane-per-matrix read-back below are bank-conflict free -- then every lane
        // takes its own matrix to registers and the park is rewritten lane-major for the sweeps.
        {
            double2 *stage = reinterpret_cast<double2 *>(lds);
            const double2 *src = reinterpret_cast<const double2 *>(de + wg * WAVE * 32);
            const int64_t pieces = (nmat - wg * WAVE) * 16;                 // valid 16-byte pieces from the tile's start
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int p = it * WAVE + lane, mm = p >> 4, c = p & 15;
                stage[mm * 16 + (c ^ (mm & 15))] = p < pieces ? src[p] : make_double2(0.0, 0.0);
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double2 v = stage[lane * 16 + (c ^ (lane & 15))];
                const double2 u = stage[lane * 16 + ((c + 8) ^ (lane & 15))];
                dv[2 * c] = v.x; dv[2 * c + 1] = v.y;
                ev[2 * c] = u.x; ev[2 * c + 1] = u.y;
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
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
#define SET_E(i, v) do { const double v_ = (v); E_(i) = v_; const uint32_t b_ = 1u << (i); negE = (negE & ~b_) | (fabs(v_) <= tiny ? b_ : 0u); } while (0)
#define SET_W(i, v) do { const double v_ = (v); W_(i) = v_; const uint32_t b_ = 1u << (i); negW = (negW & ~b_) | (fabs(v_) <= tiny ? b_ : 0u); } while (0)
        // Every lane walks its own deflation index k.  Converged values are popped in a short inner
        // loop that touches registers only, so a lane never spends a whole sweep slot of the wave on
        // "k--" while its neighbours sweep.
        int k = 15, its = 0, l = 0;
        bool noconv = false;
        uint32_t my_steps = 0, my_sweeps = 0, wave_iters = 0;          // diagnostics (stats != nullptr)
        for (;;) {
            for (;;) {
                // split point l = largest l <= k with e[l] negligible (or l == 0), unless a negligible
                // w[l-1] is met first (then e[l] has to be chased out of the block: "cancel")
                const uint32_t stopE = negE | 1u, stopW = negW << 1;
                const uint32_t stops = (stopE | stopW) & ((2u << k) - 1u);
                l = 31 - __builtin_clz(stops);
                if (((stopE >> l) & 1u) == 0) {                       // cancel (rare)
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
                if (l != k) {
                    if (its < maxit) break;         // a block of >= 2 values that still needs sweeps
                    noconv = true;                  // iteration cap: keep what we have, but say so
                }
                if (--k < 0) break;                 // converged
                its = 0;
            }
            if (k < 0) break;
            ++its;
            // shift from the bottom 2x2 minor
            const int nm = k - 1;
            double x = W_(l);
            double y = W_(nm);
            double g = E_(nm);
            double h = E_(k);
            const double z = W_(k);
            // the shift only steers convergence (any shift gives an orthogonal sweep), so its three reciprocals and
            // its square root need no more than one Newton step (~1e-15)
            double f = ((y - z) * (y + z) + (g - h) * (g + h)) * rcp_nr<1>(2.0 * h * y);
            {
                const double t2 = fma(f, f, 1.0);
                g = t2 * rsq_nr<1>(t2);
            }
            f = ((x - z) * (x + z) + h * (y * rcp_nr<1>(f + copysign(g, f)) - h)) * rcp_nr<1>(x);
            double cc = 1.0, ss = 1.0;
            // one QR sweep over the block [l,k]; the LDS reads of the next step are issued before the
            // current step's arithmetic, each hypot shares one rsq with the reciprocal its rotation
            // needs, and a zero pivot is a select, not a branch (f = h = 0 there, so c = s = 0 * rz)
            double gn = E_(l + 1), yn = W_(l + 1);
            if (work_out) {
                my_steps += (uint32_t)(nm - l + 1);
                ++my_sweeps;
            }
            if (stats) {
                my_steps += (uint32_t)(nm - l + 1);
                ++my_sweeps;
                // what the wave pays for this sweep: the longest block among its lanes
                const int len = nm - l + 1;
                int mx = len;
                for (uint64_t act = __ballot(1); act; act &= act - 1)       // only lanes that sweep now (slow; diagnostics)
                    mx = max(mx, __shfl(len, (int)__builtin_ctzll(act), WAVE));
                wave_iters += (uint32_t)mx;
            }
            for (int jj = l; jj <= nm; ++jj) {
                g = gn;
                y = yn;
                const int i2 = min(jj + 2, 15);
                gn = E_(i2);
                yn = W_(i2);
                h = ss * g;
                g = cc * g;
                double zz = fma(f, f, h * h);
                double rz = rsq_nr<RSQ_NR>(fmax(zz, 1e-290));
                SET_E(jj, zz * rz);
                cc = f * rz;
                ss = h * rz;
                f = fma(x, cc, g * ss);
                g = fma(g, cc, -(x * ss));
                h = y * ss;
                y *= cc;
                zz = fma(f, f, h * h);
                rz = rsq_nr<RSQ_NR>(fmax(zz, 1e-290));
                SET_W(jj, zz * rz);
                const bool nz = zz > 0.0;
                cc = nz ? f * rz : cc;
                ss = nz ? h * rz : ss;
                f = fma(cc, g, ss * y);
                x = fma(cc, y, -(ss * g));
            }
            SET_E(l, 0.0);
            SET_E(k, f);
            SET_W(k, x);
        }
#undef SET_E
#undef SET_W
        if (work_out && live) work_out[m] = my_steps | (my_sweeps << 16);     // test hook (tq_debug_bdsqr): this matrix's work
        if (stats) {
            // a lane takes part in every sweep of the wave until its own matrix is done, so its wave_iters is what
            // the wave had issued when it finished; the maximum over the lanes is what the wave issued in all
            uint32_t wi = wave_iters;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wi = max(wi, (uint32_t)__shfl_xor((int)wi, o, WAVE));
            if (live) {
                atomicAdd(&stats[0], 1ull);
                atomicAdd(&stats[1], (unsigned long long)my_steps);
                atomicAdd(&stats[3], (unsigned long long)my_sweeps);
                atomicAdd(&stats[4], (unsigned long long)(wi - wave_iters));     // slots after this lane's matrix was done
            }
            if (lane == 0) atomicAdd(&stats[2], 64ull * wi);
        }
        // results: 16 values per matrix = 8 KiB per wave, staged matrix-major through the park (same piece swizzle)
        // and written as eight coalesced 1 KiB wave accesses
        {
            double out[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // a matrix that hit the sweep cap hands its values over with the sign bit set (also on
                // zeros): tq_score_kernel turns that into TQ_FLAG_NO_CONVERGENCE (numpy raises LinAlgError)
                const double v = fabs(W_(i));
                out[i] = noconv ? -v : v;
            }
            __syncthreads();
            double2 *stage = reinterpret_cast<double2 *>(lds);
#pragma unroll
            for (int c = 0; c < 8; ++c) stage[lane * 8 + (c ^ (lane & 7))] = make_double2(out[2 * c], out[2 * c + 1]);
            __syncthreads();
            double2 *dst = reinterpret_cast<double2 *>(sv + wg * WAVE * 16);
            const int64_t pieces = (nmat - wg * WAVE) * 8;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = it * WAVE + lane, mm = p >> 3, c = p & 7;
                if (p < pieces) dst[p] = stage[mm * 8 + (c ^ (mm & 7))];
            }
            __syncthreads();
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
    int noconv = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        double s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double v = sv[(q * 3 + t) * 16 + i];
            noconv |= __double2hiint(v) < 0;                 // sign bit: tq_bdsqr_kernel's "not converged"
            s[i] = fabs(v);
        }
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
    if (noconv) fl |= TQ_FLAG_NO_CONVERGENCE;
    const uint32_t nsn = nsnps_in[q];
    if (nsn == 0) {
        topo = 0;
        sc[0] = sc[1] = sc[2] = 0.001;
        fl = TQ_FLAG_ZERO_DATA;
    }
    const uint4 qv = reinterpret_cast<const uint4 *>(quartets)[q];
    const uint32_t Tu = (uint32_t)T;
    if ((qv.x >= Tu) | (qv.y >= Tu) | (qv.z >= Tu) | (qv.w >= Tu)) fl |= TQ_FLAG_BAD_INDEX;
    fl |= out.flag_or;
    out.rstat[q * 2 + 0] = (uint32_t)topo;
    out.rstat[q * 2 + 1] = nsn;
    out.rscor[q * 3 + 0] = sc[0];
    out.rscor[q * 3 + 1] = sc[1];
    out.rscor[q * 3 + 2] = sc[2];
    if (out.flags) out.flags[q] = (uint8_t)fl;
}

