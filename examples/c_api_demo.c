/* Minimal C host for the engine's C ABI (include/tetrad_hip.h): no Python, no torch.
 *   gcc -O2 examples/c_api_demo.c -Iinclude -Ltetrad_amd/csrc -ltetrad_hip -Wl,-rpath,$PWD/tetrad_amd/csrc -o c_api_demo
 * Prints one line per quartet: a b c d score0 score1 score2 topo nsnps  (the reference's TSV row,
 * tetrad/src/run_inference.py:233-234).                                                              */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    /* the same call with page-locked result arrays from the library's pool (written by the copy engine directly) */
    uint32_t *p_rstat = NULL;
    double *p_rscor = NULL;
    if (tq_host_alloc(Q * 8, (void **)&p_rstat) || tq_host_alloc(Q * 24, (void **)&p_rscor)) return 3;
    rc = tq_resolve(ctx, quartets, Q, subsample, p_rstat, p_rscor, NULL);
    if (rc || memcmp(p_rstat, rstat, (size_t)Q * 8) || memcmp(p_rscor, rscor, (size_t)Q * 24)) {
        fprintf(stderr, "pinned result arrays differ (rc=%d)\n", rc);
        return 4;
    }
    tq_host_free(p_rstat);
    tq_host_free(p_rscor);
    /* the consumers after the hot path, natively: wQMC lines (weights strategy 1) and the quartet supertree */
    {
        int64_t written = 0, nlines = 0, cap = 64 * Q + 4096;
        char *text = malloc((size_t)cap);
        rc = tq_format_qmc(quartets, rstat, rscor, Q, 1, 0, 1.0, text, cap, &written, &nlines);
        if (rc) return 5;
        uint32_t *splits = malloc((size_t)nlines * 16);
        double *w = malloc((size_t)nlines * 8);
        const char *c = text;
        for (int64_t i = 0; i < nlines; ++i) {                    /* "a,b|c,d:weight\n" */
            unsigned a, b, cc, d;
            double wt;
            int used = 0;
            if (sscanf(c, "%u,%u|%u,%u:%lf%n", &a, &b, &cc, &d, &wt, &used) != 5) return 6;
            splits[4 * i] = a; splits[4 * i + 1] = b; splits[4 * i + 2] = cc; splits[4 * i + 3] = d;
            w[i] = wt;
            c += used + 1;
        }
        char nwk[1024];
        rc = tq_qmc_tree(splits, w, nlines, T, 7, nwk, sizeof nwk - 1, &written);
        if (rc) return 7;
        nwk[written] = 0;
        fprintf(stderr, "tree: %s (%lld weighted quartets)\n", nwk, (long long)nlines);
        free(text); free(splits); free(w);
    }
    /* error path: a taxon index out of range must be refused with a message, not crash */
    quartets[3] = 99;
    rc = tq_resolve(ctx, quartets, 1, subsample, rstat, rscor, flags);
    fprintf(stderr, "bad index -> rc=%d (%s)\n", rc, tq_last_error(ctx));
    tq_destroy(ctx);
    return rc == TQ_ERR_INVALID_ARG ? 0 : 2;
}
