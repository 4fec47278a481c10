// prepare.hpp -- layout build, lexicographic unranking, sort keys
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
#pragma once

__device__ __forceinline__ void store_planes3(uint32_t *planes3, uint32_t *runbeg, int64_t t, int64_t W, int64_t w,
                                              uint32_t mm, uint32_t b0, uint32_t b1, uint32_t rb)
{
    uint32_t *p3 = planes3 + (t * W + w) * 3;
    p3[0] = mm;
    p3[1] = b0;
    p3[2] = b1;
    if (t == 0) runbeg[w] = rb;
}

// ------------------------------------------------------------------------------------
// data preparation kernel: one thread per 32-site word of one taxon row
// ------------------------------------------------------------------------------------
__global__ void tq_prepare_rows(const uint8_t *__restrict__ raw, const uint32_t *__restrict__ locus,
                                int64_t S, int64_t Sp, int64_t W, int32_t T, uint8_t *__restrict__ rows,
                                uint8_t *__restrict__ nib, uint8_t *__restrict__ nib5, uint4 *__restrict__ planes,
                                uint32_t *__restrict__ planes3, uint32_t *__restrict__ runbeg)
{
    int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)T * W) return;
    int64_t t = gid / W, w = gid - t * W;
    const uint8_t *src = raw + t * S + w * 32;
    uint8_t *dst = rows + t * Sp;
    uint32_t mm = 0, b0 = 0, b1 = 0, rb = 0;
    uint32_t nw[4] = {0, 0, 0, 0}, n5[4] = {0, 0, 0, 0};   // this word's 32 sites = 16 nibble bytes (and with 4 = missing)
    for (int i = 0; i < 32; ++i) {
        int64_t s = w * 32 + i;
        uint8_t v = (s < S) ? src[i] : (uint8_t)0xFF;
        bool missing = v > 3;
        uint8_t code = missing ? (uint8_t)0 : v;
        dst[row_offset(s)] = code;
        nw[i >> 3] |= (uint32_t)code << (8 * (i & 3) + 4 * ((i >> 2) & 1));
        n5[i >> 3] |= (uint32_t)(missing ? 4 : code) << (8 * (i & 3) + 4 * ((i >> 2) & 1));
        mm |= (uint32_t)missing << i;
        b0 |= (uint32_t)(code & 1) << i;
        b1 |= (uint32_t)((code >> 1) & 1) << i;
        if (s < S) {
            bool beg = (s == 0) || (locus[s] != locus[s - 1]);
            rb |= (uint32_t)beg << i;
        }
    }
    planes[t * W + w] = make_uint4(mm, b0, b1, rb);
    store_planes3(planes3, runbeg, t, W, w, mm, b0, b1, rb);
    reinterpret_cast<uint4 *>(nib + t * (Sp / 2))[w] = make_uint4(nw[0], nw[1], nw[2], nw[3]);
    reinterpret_cast<uint4 *>(nib5 + t * (Sp / 2))[w] = make_uint4(n5[0], n5[1], n5[2], n5[3]);
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

// sort key of a quartet: its first two taxa (quartets sharing them share two of their four rows in
// LDS), then -- when it fits 32 bits -- the third (neighbours then also find row c in the caches)
__global__ void tq_key_kernel(const uint32_t *__restrict__ quartets, int64_t Q, uint32_t T, int with_c,
                              uint32_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const uint4 q = reinterpret_cast<const uint4 *>(quartets)[i];
    const uint32_t a = q.x < T ? q.x : T - 1, b = q.y < T ? q.y : T - 1, c = q.z < T ? q.z : T - 1;
    const uint32_t ab = a * T + b;
    keys[i] = with_c ? ab * T + c : ab;
    idx[i] = (uint32_t)i;
}


// ------------------------------------------------------------------------------------
// Random quartet sample on the device (opt-in): Q distinct lexicographic
// ranks drawn uniformly from C(T,4), in random order -- the distribution of
// `rng.choice(C(T,4), size=Q, replace=False)` (combinations.py:109-114) -- without the host's 30 ms draw.
// Rank i of the sample is pi(i), pi = a keyed pseudo-random permutation of [0, N): a 6-round Feistel
// network on the smallest even-split bit width >= log2 N, cycle-walked into range (a permutation of
// [0, 2^b) restricted to [0, N) by iterating until the value lands below N is a permutation of [0, N)).
// The first Q values of a random permutation are a uniform ordered sample without replacement.  Not the
// project Generator's stream: statistically equivalent, documented as such (like the device bootstrap).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t feistel_round(uint32_t x, uint32_t key)
{
    x ^= key;
    x *= 0x7FEB352Du;              // lowbias32 (two multiply-xorshift rounds)
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// six independent 32-bit round keys from the 64-bit seed: a splitmix64 chain (Steele, Lea & Flood 2014), one output
// per round (shifted bytes of one seed plus a round constant -- the first version -- made rounds r and r+4 differ by a
// constant only)
struct FeistelKeys {
    uint32_t k[6];
};

__device__ __forceinline__ FeistelKeys feistel_keys(uint64_t seed)
{
    FeistelKeys ks;
    uint64_t x = seed;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        ks.k[r] = (uint32_t)(z >> 16);
    }
    return ks;
}

__device__ __forceinline__ uint64_t feistel_permute(uint64_t v, int half_bits, const FeistelKeys &ks)
{
    const uint32_t mask = half_bits >= 32 ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
    uint32_t L = (uint32_t)(v >> half_bits) & mask, R = (uint32_t)v & mask;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const uint32_t t = L ^ (feistel_round(R, ks.k[r]) & mask);
        L = R;
        R = t;
    }
    return ((uint64_t)L << half_bits) | (uint64_t)R;
}

__global__ void tq_sample_kernel(uint64_t seed, uint64_t N, int half_bits, int64_t Q, int32_t T,
                                 uint64_t *__restrict__ ranks_out, uint32_t *__restrict__ quartets)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Q) return;
    const FeistelKeys ks = feistel_keys(seed);
    uint64_t v = (uint64_t)i;
    do {                                    // expected < 4 rounds: 2^(2*half_bits) < 4N
        v = feistel_permute(v, half_bits, ks);
    } while (v >= N);
    if (ranks_out) ranks_out[i] = v;
    uint64_t index = v;
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
    reinterpret_cast<uint4 *>(quartets)[i] = make_uint4(out[0], out[1], out[2], out[3]);
}
