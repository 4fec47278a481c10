/*
 * oracle/count_matrices.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the integer part of tetrad's per-quartet hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; tetrad_amd/ never does.
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference checkout, tetrad/src/resolve_quartets.py unless noted).  The
 * numba-jitted loops in the reference are serial scalar loops, so a serial C
 * loop is the closest CPU stand-in for them.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against
 * tests/golden/*.npz, which were produced by running the reference's own
 * functions (tests/golden/make_golden.py).
 */
#include <stdint.h>
#include <string.h>

/* resolve_quartets.py:212-218  --  seqs = tmparr[sidx,:];
 *   nmask0 = sum(seqs >= 78, axis=0); nmask1 = sum(seqs == seqs[0], axis=0) == 4
 *   mask = nmask0 + nmask1  (non-zero => site is skipped, :60 / :93)
 * seqs is u8[4,S] row-major; mask out is u8[S] (0 keep, non-zero skip).
 * Values 4..77 would index outside the 16x16 matrix in the reference
 * (undefined under numba); this restatement masks them (documented deviation
 * for inputs that violate the reference's precondition). */
void oracle_quartet_mask(const uint8_t *seqs, int64_t S, uint8_t *mask)
{
    const uint8_t *r0 = seqs, *r1 = seqs + S, *r2 = seqs + 2 * S, *r3 = seqs + 3 * S;
    for (int64_t i = 0; i < S; i++) {
        int nmask0 = (r0[i] >= 78) + (r1[i] >= 78) + (r2[i] >= 78) + (r3[i] >= 78);
        int nmask1 = (r1[i] == r0[i]) && (r2[i] == r0[i]) && (r3[i] == r0[i]);
        int bad = (r0[i] > 3) | (r1[i] > 3) | (r2[i] > 3) | (r3[i] > 3);
        mask[i] = (uint8_t)((nmask0 + nmask1 + bad) != 0);
    }
}

/* resolve_quartets.py:66-72 and :97-103 -- "fill the alternates":
 *   x runs over the 16 rows of mats[0]; block (y,z) of mats[1] gets
 *   mats[0][x].reshape(4,4), block (y,z) of mats[2] gets its transpose. */
static void fill_alternates(uint32_t *mats)
{
    uint32_t *m0 = mats, *m1 = mats + 256, *m2 = mats + 512;
    int x = 0;
    for (int y = 0; y < 16; y += 4) {
        for (int z = 0; z < 16; z += 4) {
            for (int a = 0; a < 4; a++) {
                for (int b = 0; b < 4; b++) {
                    m1[(y + a) * 16 + (z + b)] = m0[x * 16 + 4 * a + b];
                    m2[(y + a) * 16 + (z + b)] = m0[x * 16 + 4 * b + a];
                }
            }
            x++;
        }
    }
}

/* resolve_quartets.py:76-104 full_chunk_to_matrices(tmparr=seqs, tmpmap=locus, mask)
 * mats is u32[3,16,16]. */
void oracle_full_chunk_to_matrices(const uint8_t *seqs, int64_t S,
                                   const uint32_t *locus, const uint8_t *mask,
                                   uint32_t *mats)
{
    (void)locus;
    memset(mats, 0, 3 * 256 * sizeof(uint32_t));
    const uint8_t *r0 = seqs, *r1 = seqs + S, *r2 = seqs + 2 * S, *r3 = seqs + 3 * S;
    for (int64_t idx = 0; idx < S; idx++) {          /* :92 */
        if (!mask[idx]) {                             /* :93 */
            mats[(4 * r0[idx] + r1[idx]) * 16 + (4 * r2[idx] + r3[idx])] += 1; /* :95 */
        }
    }
    fill_alternates(mats);
}

/* resolve_quartets.py:42-73 subsample_chunk_to_matrices.
 * last_loc starts at uint32(-1) = 4294967295 (:58, numba wrap-around);
 * a site is counted iff unmasked and locus[idx] != last_loc (:60-61), and
 * last_loc is updated only when a site is counted (:64). */
void oracle_subsample_chunk_to_matrices(const uint8_t *seqs, int64_t S,
                                        const uint32_t *locus, const uint8_t *mask,
                                        uint32_t *mats)
{
    memset(mats, 0, 3 * 256 * sizeof(uint32_t));
    const uint8_t *r0 = seqs, *r1 = seqs + S, *r2 = seqs + 2 * S, *r3 = seqs + 3 * S;
    uint32_t last_loc = 0xFFFFFFFFu;                  /* :58 */
    for (int64_t idx = 0; idx < S; idx++) {           /* :59 */
        if (!mask[idx]) {                             /* :60 */
            if (locus[idx] != last_loc) {             /* :61 */
                mats[(4 * r0[idx] + r1[idx]) * 16 + (4 * r2[idx] + r3[idx])] += 1; /* :63 */
                last_loc = locus[idx];                /* :64 */
            }
        }
    }
    fill_alternates(mats);
}

/* Convenience for the timed CPU baseline: :212-223 for one quartet without
 * the intermediate NumPy temporaries (row gather + masks + count loop).
 * tmparr is u8[T,S] row-major, quartet is 4 row indices, scratch is u8[5*S]. */
void oracle_quartet_to_matrices(const uint8_t *tmparr, int64_t S,
                                const uint32_t *locus, const uint32_t *quartet,
                                int subsample, uint8_t *scratch, uint32_t *mats)
{
    uint8_t *seqs = scratch, *mask = scratch + 4 * S;
    for (int r = 0; r < 4; r++)                       /* :212 */
        memcpy(seqs + (int64_t)r * S, tmparr + (int64_t)quartet[r] * S, (size_t)S);
    oracle_quartet_mask(seqs, S, mask);               /* :216-218 */
    if (subsample)
        oracle_subsample_chunk_to_matrices(seqs, S, locus, mask, mats);
    else
        oracle_full_chunk_to_matrices(seqs, S, locus, mask, mats);
}
