// format.hpp -- host-side text producers for the consumers right after the hot path
// (SURVEY.md section 8 row f3): the 9-column quartets TSV (run_inference.py:233-234) and the wQMC
// input lines (run_inference.py:254-305).  No device code; part of the single translation unit
// tetrad_hip.hip (included inside its anonymous namespace).
#pragma once

// Decimal text of x with `dec` (<= 9) digits after the point, identical to printf("%.<dec>f").
// Fast path: round x * 10^dec to the nearest integer; when the product is so close to a tie that its
// own rounding error could decide the direction, or x is outside the comfortable range, snprintf
// (exact decimal expansion of the binary value) takes over.
inline char *put_fixed(char *p, double x, int dec)
{
    static const double P10[10] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
    if (x >= 0.0 && x < 4.0e9 && dec >= 0 && dec <= 9) {
        const double scaled = x * P10[dec];                 // relative error 2^-53: below 5e-7 absolute
        const double fl = std::floor(scaled);               // while scaled < 2^32
        const double frac = scaled - fl;
        if (std::fabs(frac - 0.5) > 2e-6 && scaled < 4.0e9) {
            uint64_t v = (uint64_t)fl + (frac > 0.5 ? 1u : 0u);
            char tmp[40];
            int n = 0;
            for (int i = 0; i < dec; ++i) {
                tmp[n++] = (char)('0' + v % 10);
                v /= 10;
            }
            if (dec) tmp[n++] = '.';
            do {
                tmp[n++] = (char)('0' + v % 10);
                v /= 10;
            } while (v);
            while (n) *p++ = tmp[--n];
            return p;
        }
    }
    return p + snprintf(p, 400, "%.*f", dec, x);
}

inline char *put_u32(char *p, uint32_t v)
{
    char tmp[12];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

constexpr int64_t TSV_MAX_LINE = 4 * 11 + 3 * 330 + 2 * 11;     // generous upper bound per row (f64 max ~ 1e308)
constexpr int64_t QMC_MAX_LINE = 4 * 11 + 330;

// rows "a\tb\tc\td\tscore0\tscore1\tscore2\ttopo\tnsnps\n"; returns bytes written, or -(bytes the
// buffer must hold) when `cap` is too small for the worst case
inline int64_t format_tsv(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q, char *out,
                          int64_t cap)
{
    char *p = out;
    char *const end = out + cap;
    for (int64_t i = 0; i < Q; ++i) {
        if (end - p < TSV_MAX_LINE) {
            // slow exact sizing of what is left, so that the caller can retry with the right size
            int64_t need = p - out;
            char line[TSV_MAX_LINE];
            for (int64_t j = i; j < Q; ++j) {
                char *q = line;
                for (int k = 0; k < 4; ++k) { q = put_u32(q, quartets[j * 4 + k]); *q++ = '\t'; }
                for (int k = 0; k < 3; ++k) { q = put_fixed(q, rscor[j * 3 + k], 6); *q++ = '\t'; }
                q = put_u32(q, rstat[j * 2]); *q++ = '\t';
                q = put_u32(q, rstat[j * 2 + 1]); *q++ = '\n';
                need += q - line;
            }
            return -(need + TSV_MAX_LINE);
        }
        for (int k = 0; k < 4; ++k) { p = put_u32(p, quartets[i * 4 + k]); *p++ = '\t'; }
        for (int k = 0; k < 3; ++k) { p = put_fixed(p, rscor[i * 3 + k], 6); *p++ = '\t'; }
        p = put_u32(p, rstat[i * 2]); *p++ = '\t';
        p = put_u32(p, rstat[i * 2 + 1]); *p++ = '\n';
    }
    return p - out;
}

// The double a reader gets back from the text of x with `dec` decimals ("%.<dec>f" then float()): away from a
// rounding tie that is the integer nearest to x * 10^dec, divided by 10^dec -- one correctly rounded division of two
// exactly representable numbers, hence the double nearest to the decimal string, which is what strtod returns; near
// a tie (or out of range) the text is actually produced and parsed.
inline double reread_dec(double x, int dec)
{
    static const double P10[10] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
    if (x >= 0.0 && x < 4.0e9 && dec >= 0 && dec <= 9) {
        const double scaled = x * P10[dec];
        const double fl = std::floor(scaled);
        const double frac = scaled - fl;
        if (std::fabs(frac - 0.5) > 2e-6 && scaled < 4.0e9) return (fl + (frac > 0.5 ? 1.0 : 0.0)) / P10[dec];
    }
    char buf[400];
    char *e = put_fixed(buf, x, dec);
    *e = 0;
    return strtod(buf, nullptr);
}

// the value the reference reads back from the TSV: the score rounded to 6 decimals as text
inline double reread6(double x) { return reread_dec(x, 6); }

// One row of the quartets table -> does it pass min_snps / min_ratio (run_inference.py:258,275,300), which split does
// its topology mean (:264-270) and what is its weight (:280-297, from the scores as they read back from the TSV).
inline bool qmc_row(const uint32_t *q, uint32_t order, uint32_t nsnps, const double *sc, int weights, int64_t min_snps,
                    double min_ratio, uint32_t (&split)[4], double &weight)
{
    if ((int64_t)nsnps < min_snps) return false;                      // :275
    weight = 1.0;
    double ratio = 1.0;
    if (weights) {                                                    // :284-297
        double s[3] = {reread6(sc[0]), reread6(sc[1]), reread6(sc[2])};
        if (s[0] > s[1]) std::swap(s[0], s[1]);
        if (s[1] > s[2]) std::swap(s[1], s[2]);
        if (s[0] > s[1]) std::swap(s[0], s[1]);
        const double smean = (s[1] + s[2]) / 2.0;                     // numpy mean of two values
        const double smin = s[0];
        ratio = smin == 0.0 ? 1.0 : smean / smin;
        if (weights == 1) weight = smean;
        else if (weights == 2) weight = ratio;
        else weight = 1.0 - smin / ((s[0] + s[1]) + s[2]);            // numpy sum, left to right
    }
    if (ratio < min_ratio) return false;                              // :300
    split[0] = q[0]; split[1] = q[1]; split[2] = q[2]; split[3] = q[3];
    if (order == 1) { split[1] = q[2]; split[2] = q[1]; }             // a,c|b,d
    else if (order == 2) { split[1] = q[3]; split[2] = q[1]; split[3] = q[2]; }   // a,d|b,c
    return true;
}

// the same rows as arrays: splits u32[n,4] = a,b|c,d and the weights as the wQMC input file would carry them
// ("%.5f" text read back); returns the number of rows kept
inline int64_t qmc_splits(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q, int weights,
                          int64_t min_snps, double min_ratio, uint32_t *splits, double *wout)
{
    if (min_snps < 1) min_snps = 1;                                   // :258
    int64_t n = 0;
    for (int64_t i = 0; i < Q; ++i) {
        uint32_t sp[4];
        double w;
        if (!qmc_row(quartets + i * 4, rstat[i * 2], rstat[i * 2 + 1], rscor + i * 3, weights, min_snps, min_ratio, sp, w))
            continue;
        memcpy(splits + 4 * n, sp, 16);
        wout[n] = reread_dec(w, 5);                                   // :305 "{:.5f}"
        ++n;
    }
    return n;
}

// "a,b|c,d:weight\n" lines (run_inference.py:264-305) for the rows that pass the two filters;
// returns bytes written (or -needed), *n_lines = lines produced
inline int64_t format_qmc(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q, int weights,
                          int64_t min_snps, double min_ratio, char *out, int64_t cap, int64_t *n_lines)
{
    if (min_snps < 1) min_snps = 1;                                   // :258
    char *p = out;
    char *const end = out + cap;
    int64_t lines = 0;
    for (int64_t i = 0; i < Q; ++i) {
        if (end - p < QMC_MAX_LINE) return -((p - out) + (Q - i) * QMC_MAX_LINE);
        uint32_t sp[4];
        double weight;
        if (!qmc_row(quartets + i * 4, rstat[i * 2], rstat[i * 2 + 1], rscor + i * 3, weights, min_snps, min_ratio, sp, weight))
            continue;
        const uint32_t a = sp[0], b = sp[1], c = sp[2], d = sp[3];
        p = put_u32(p, a); *p++ = ',';
        p = put_u32(p, b); *p++ = '|';
        p = put_u32(p, c); *p++ = ',';
        p = put_u32(p, d); *p++ = ':';
        p = put_fixed(p, weight, 5); *p++ = '\n';
        ++lines;
    }
    if (n_lines) *n_lines = lines;
    return p - out;
}

// lexicographic unranking of 4-combinations on the host (combinations.py:94-106, `_index_to_combination`, one
// O(T) Python loop per quartet in the reference): rank -> (a,b,c,d), a<b<c<d, by walking the same
// "does the block of combinations that start with t contain the rank?" comparisons with closed-form binomials
inline uint64_t host_choose(uint64_t n, int k)
{
    switch (k) {
    case 0: return 1;
    case 1: return n;
    case 2: return n < 2 ? 0 : n * (n - 1) / 2;
    default: return n < 3 ? 0 : n * (n - 1) / 2 * (n - 2) / 3;
    }
}

inline uint64_t host_choose4(uint64_t n) { return n < 4 ? 0 : n * (n - 1) / 2 * (n - 2) / 3 * (n - 3) / 4; }

inline void unrank_host(const uint64_t *ranks, uint64_t first_rank, int64_t Q, int32_t T, uint32_t *quartets)
{
    // With k picks left and candidates p..T-1, the combinations whose next pick is < t number
    // C(T-p, k) - C(T-t, k); the pick is the largest t for which that is <= the remaining index: a binary
    // search per level (about 4 x log2 T steps instead of the reference's walk over all T candidates).
    auto choose = [](uint64_t n, int k) { return k == 4 ? host_choose4(n) : host_choose(n, k); };
    for (int64_t i = 0; i < Q; ++i) {
        uint64_t index = ranks ? ranks[i] : first_rank + (uint64_t)i;
        uint32_t out[4];
        uint32_t p = 0;
        for (int k = 4; k >= 1; --k) {
            const uint64_t all = choose((uint64_t)(T - p), k);
            uint32_t lo = p, hi = (uint32_t)(T - k);                  // the pick lies in [lo, hi]
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo + 1) / 2;
                if (all - choose((uint64_t)(T - mid), k) <= index) lo = mid;
                else hi = mid - 1;
            }
            index -= all - choose((uint64_t)(T - lo), k);
            out[4 - k] = lo;
            p = lo + 1;
        }
        memcpy(quartets + 4 * i, out, 16);
    }
}

// ------------------------------------------------------------------------------------------------------
// The reference draws its quartet sample with `rng.choice(C(T,4), size, replace=False)` on the project's NumPy
// Generator (combinations.py:113): 31 ms for 1e6 of 10.7e6 -- more than a bootstrap replicate's kernels -- and
// its stream is what a reproducible run is defined by.  For samples larger than 1/50 of the population NumPy
// (numpy/random/_generator.pyx, `Generator.choice`; third-party dependency of the reference, unpinned by it,
// 2.2.6 here) shuffles the TAIL of arange(pop): for i = pop-1 down to pop-size it swaps position i with a
// position j drawn uniformly from [0, i] by Lemire's multiply-shift rejection on 32-bit draws, then returns
// positions pop-size .. pop-1.  This function makes THE SAME draws from the SAME bit generator (through the
// function pointers NumPy publishes as `bit_generator.ctypes`) and applies the same swaps -- on a sparse map of
// the positions touched instead of an 8*pop-byte array, with the map slots of the next draws prefetched -- so
// it returns NumPy's sample and leaves the Generator in NumPy's state.  tetrad_amd/combinations.py checks it
// against `Generator.choice` itself before trusting it (and tests/test_combinations.py for many shapes).
// ------------------------------------------------------------------------------------------------------
struct NpBitgen {                 // numpy/random/bitgen.h: bitgen_t
    void *state;
    uint64_t (*next_uint64)(void *);
    uint32_t (*next_uint32)(void *);
    double (*next_double)(void *);
    uint64_t (*next_raw)(void *);
};

inline uint32_t np_bounded_lemire_u32(NpBitgen *bg, uint32_t rng)        // uniform on [0, rng], rng < 2^32 - 1
{
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)bg->next_uint32(bg->state) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)bg->next_uint32(bg->state) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

// out[k] = value that ends at position (pop - size + k) of the tail-shuffled arange(pop); pop <= 2^32 - 2.
// Positions >= pop - size (the part that is returned) live in `out` itself; positions below it are only ever the
// target `j` of a swap and live in a sparse open-addressing map (one 8-byte slot = position+1 in the high half,
// value in the low half).  The draws of the next 32 swaps are made ahead and their slots prefetched.
inline int numpy_choice_tail(NpBitgen *bg, uint64_t pop, int64_t size, int64_t *out)
{
    const uint64_t base = pop - (uint64_t)size;                                     // first returned position
    const uint64_t first = base > 1 ? base : 1;                                     // max(pop - size, 1)
    const uint64_t count = pop - first;                                             // swaps to make
    for (int64_t k = 0; k < size; ++k) out[k] = (int64_t)(base + (uint64_t)k);      // arange, tail part
    uint64_t cap = 1024;
    while (cap < 2 * count) cap <<= 1;
    const uint64_t mask = cap - 1;
    static thread_local std::vector<uint64_t> head;
    try {
        head.assign((size_t)cap, 0ull);
    } catch (const std::bad_alloc &) {
        return TQ_ERR_OOM;
    }
    uint64_t *slots = head.data();
    auto home = [mask](uint64_t pos) { return ((pos * 0x9E3779B97F4A7C15ull) >> 24) & mask; };
    constexpr int B = 32;
    uint32_t jb[B];
    uint64_t i = pop - 1, left = count;
    while (left) {
        const int nb = left < (uint64_t)B ? (int)left : B;
        for (int b = 0; b < nb; ++b) {
            const uint32_t j = np_bounded_lemire_u32(bg, (uint32_t)(i - (uint64_t)b));
            jb[b] = j;
            if (j >= base) __builtin_prefetch(&out[j - base], 1);
            else __builtin_prefetch(&slots[home(j)], 1);
        }
        for (int b = 0; b < nb; ++b, --i) {
            const uint64_t j = jb[b];
            const uint64_t vi = (uint64_t)out[i - base];          // i >= first >= base always
            uint64_t vj;
            if (j >= base) {                                      // both in the returned part
                vj = (uint64_t)out[j - base];
                out[j - base] = (int64_t)vi;
            } else {
                uint64_t h = home(j);
                const uint64_t key = (j + 1) << 32;
                while (slots[h] != 0 && (slots[h] & 0xFFFFFFFF00000000ull) != key) h = (h + 1) & mask;
                vj = slots[h] ? (slots[h] & 0xFFFFFFFFull) : j;
                slots[h] = key | vi;                              // data[j] = data[i]
            }
            out[i - base] = (int64_t)vj;                          // data[i] = data[j]
        }
        left -= (uint64_t)nb;
    }
    // size == pop: base = 0, position 0 is part of `out` and was handled as a `j` target; nothing left to do
    if (head.capacity() > ((size_t)1 << 25)) std::vector<uint64_t>().swap(head);   // do not sit on > 256 MB
    return TQ_OK;
}
