// common.hpp -- constants, device-side views of the resident data, output pointer bundle
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
#pragma once


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
    const uint8_t *nib;     // [T][Sp/2] two base codes per byte (see nib_offset)
    const uint8_t *nib5;    // [T][Sp/2] the same with 4 for a missing cell (rows d1, d2 of the joint-histogram scan, scan_dp.hpp)
    const uint4 *planes;    // [T][W] {miss, p0, p1, runbeg}
    const uint32_t *planes3; // [T][W][3] {miss, p0, p1}: the compact copy the cooperative scan streams
    const uint32_t *runbeg;  // [W] run-begin bits (the same for every taxon)
    int64_t pitch;      // bytes per row (Sp)
    int64_t W;          // plane records per row (Sp/32)
    int32_t T;
    int32_t ntiles;     // Sp / TILE
    uint32_t inv;       // 0, or ~0 with option "count_invariant": sites where the four bases are equal are counted too (the
                        // kernel-level mirrors of resolve_quartets.py:42-104 honour a caller's mask exactly; one-wave kernel only)
};

// Byte offset of site s inside a row.  A 2048-site step is stored as two 1 KiB panels: panel 0
// holds sites 0-15 of every lane's 32-site group, panel 1 holds sites 16-31, so each of the two
// 16-byte loads a lane issues per row is part of one fully contiguous 1 KiB wave access.
__host__ __device__ __forceinline__ int64_t row_offset(int64_t s)
{
    const int64_t tile = s >> 11, r = s & 2047, lane = r >> 5, k = r & 31;
    return (tile << 11) + ((k >> 4) << 10) + (lane << 4) + (k & 15);
}

// Nibble-packed copy of the rows, read by the cooperative scan kernel for a wave's own rows c and d
// (half the bytes through the L2 -> CU path, which is what bounds that kernel).  Eight consecutive
// sites share four bytes: byte k of a group holds site k in its low nibble and site k+4 in its high
// nibble, so `w & 0x0F0F0F0F` and `(w >> 4) & 0x0F0F0F0F` are the code bytes of sites 0-3 and 4-7 in
// site order.  A lane's 32 sites are 16 contiguous bytes; a 2048-site step is 1 KiB per row.
__host__ __device__ __forceinline__ int64_t nib_offset(int64_t s) { return ((s >> 3) << 2) + (s & 3); }
__host__ __device__ __forceinline__ int nib_shift(int64_t s) { return (int)((s >> 2) & 1) * 4; }

struct OutPtrs {
    uint32_t *rstat;    // [Q,2]
    double *rscor;      // [Q,3]
    uint8_t *flags;     // [Q] or null
    uint32_t *cmats;    // [Q,3,16,16] or null (debug)
    double *svds;       // [Q,3,16] or null (debug)
    int32_t *ranks;     // [Q,3] or null (debug)
    uint32_t flag_or;   // OR-ed into every row's flags (TQ_FLAG_INVALID_DIAGNOSTIC while a timing-diagnostic mode is set)
};

