// bootstrap.hpp -- bootstrap replicate built on the device, replicate export
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
#pragma once

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
                                     uint8_t *__restrict__ rows, uint8_t *__restrict__ nib, uint8_t *__restrict__ nib5,
                                     uint4 *__restrict__ planes, uint32_t *__restrict__ planes3,
                                     uint32_t *__restrict__ runbeg)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T * W) return;
    const int64_t t = gid / W, w = gid - t * W;
    uint8_t *dst = rows + t * Sp;
    uint32_t mm = 0, b0 = 0, b1 = 0, rb = 0;
    uint32_t nw[4] = {0, 0, 0, 0}, n5[4] = {0, 0, 0, 0};
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
        nw[i >> 3] |= (uint32_t)code << (8 * (i & 3) + 4 * ((i >> 2) & 1));
        n5[i >> 3] |= (uint32_t)(missing ? 4 : code) << (8 * (i & 3) + 4 * ((i >> 2) & 1));
        mm |= (uint32_t)missing << i;
        b0 |= (uint32_t)(code & 1) << i;
        b1 |= (uint32_t)((code >> 1) & 1) << i;
    }
    planes[t * W + w] = make_uint4(mm, b0, b1, rb);
    store_planes3(planes3, runbeg, t, W, w, mm, b0, b1, rb);
    reinterpret_cast<uint4 *>(nib + t * (Sp / 2))[w] = make_uint4(nw[0], nw[1], nw[2], nw[3]);
    reinterpret_cast<uint4 *>(nib5 + t * (Sp / 2))[w] = make_uint4(n5[0], n5[1], n5[2], n5[3]);
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

