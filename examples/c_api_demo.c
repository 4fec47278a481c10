/* Minimal C host for the engine's C ABI (include/tetrad_hip.h): no Python, no torch.
 *   gcc -O2 examples/c_api_demo.c -Iinclude -Ltetrad_amd/csrc -ltetrad_hip -Wl,-rpath,$PWD/tetrad_amd/csrc -o c_api_demo
 * Prints one line per quartet: a b c d score0 score1 score2 topo nsnps  (the reference's TSV row,
 * tetrad/src/run_inference.py:233-234).                                                              */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "tetrad_hip.h"

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(int argc, char **argv)
{
    const int64_t T = 9, S = 3000;
    const int subsample = argc > 1 ? atoi(argv[1]) : 1;
    uint8_t *tmparr = malloc((size_t)(T * S));
    uint32_t *tmpmap = malloc((size_t)S * 2 * sizeof(uint32_t));
    uint32_t seed = 12345u, locus = 0;
    for (int64_t s = 0; s < S; ++s) {
        uint8_t anc = (uint8_t)(lcg(&seed) & 3);
        for (int64_t t = 0; t < T; ++t) {
            uint32_t r = lcg(&seed) % 100;
            tmparr[t * S + s] = r < 8 ? 78 : (r < 40 ? (uint8_t)(lcg(&seed) & 3) : anc);   /* 78 = N */
        }
        if (s && lcg(&seed) % 4 == 0) ++locus;
        tmpmap[2 * s] = locus;
        tmpmap[2 * s + 1] = (uint32_t)s;
    }
    int64_t Q = 0;
    uint32_t *quartets = malloc(126 * 4 * sizeof(uint32_t));
    for (uint32_t a = 0; a < T; ++a)
        for (uint32_t b = a + 1; b < T; ++b)
            for (uint32_t c = b + 1; c < T; ++c)
                for (uint32_t d = c + 1; d < T; ++d) {
                    quartets[4 * Q] = a; quartets[4 * Q + 1] = b; quartets[4 * Q + 2] = c; quartets[4 * Q + 3] = d;
                    ++Q;
                }
    tq_ctx *ctx = NULL;
    int rc = tq_create(&ctx, 0);
    if (rc) { fprintf(stderr, "tq_create: %d %s\n", rc, tq_last_error(NULL)); return 1; }
    rc = tq_set_data(ctx, tmparr, T, S, tmpmap, 2);          /* tmpmap[:,0] read with stride 2 */
    if (rc) { fprintf(stderr, "tq_set_data: %s\n", tq_last_error(ctx)); return 1; }
    uint32_t *rstat = malloc((size_t)Q * 2 * sizeof(uint32_t));
    double *rscor = malloc((size_t)Q * 3 * sizeof(double));
    uint8_t *flags = malloc((size_t)Q);
    rc = tq_resolve(ctx, quartets, Q, subsample, rstat, rscor, flags);
    if (rc) { fprintf(stderr, "tq_resolve: %s\n", tq_last_error(ctx)); return 1; }
    for (int64_t i = 0; i < Q; ++i)
        printf("%u\t%u\t%u\t%u\t%.6f\t%.6f\t%.6f\t%u\t%u\n", quartets[4 * i], quartets[4 * i + 1], quartets[4 * i + 2],
               quartets[4 * i + 3], rscor[3 * i], rscor[3 * i + 1], rscor[3 * i + 2], rstat[2 * i], rstat[2 * i + 1]);
    /* error path: a taxon index out of range must be refused with a message, not crash */
    quartets[3] = 99;
    rc = tq_resolve(ctx, quartets, 1, subsample, rstat, rscor, flags);
    fprintf(stderr, "bad index -> rc=%d (%s)\n", rc, tq_last_error(ctx));
    tq_destroy(ctx);
    return rc == TQ_ERR_INVALID_ARG ? 0 : 2;
}
