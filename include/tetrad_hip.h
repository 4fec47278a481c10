/*
 * tetrad_hip.h -- C ABI of the MI355X (gfx950) quartet-invariant engine.
 *
 * This is the drop-in boundary for tetrad's per-chunk worker.  Each entry point
 * cites the reference interface it replaces (paths relative to the reference
 * checkout).  The reference is pure Python, so "the FFI a maintainer would
 * bind" is ctypes; the binding is shown in INTEGRATION.md and shipped as
 * tetrad_amd/_lib.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.
 *   - every function returns TQ_OK (0) or a negative TQ_ERR_* code; it never
 *     throws and never aborts.  tq_last_error() returns a human-readable
 *     message for the last failure on that context (or the last failure of
 *     tq_create when ctx is NULL).
 *   - one context per (process, device).  A context is not thread-safe;
 *     distinct contexts are independent.  The library keeps no host pointer
 *     after a call returns.
 *   - host-buffer entry points (tq_set_data, tq_resolve, tq_resolve_to_host,
 *     tq_resolve_debug) are synchronous: results are in the caller's arrays on
 *     return.  Inside, kernels and copies run on the context's own streams and the
 *     result D2H of one piece overlaps the kernels of the next.  *_dev entry points
 *     take device pointers, enqueue on the given HIP stream and return without
 *     synchronising.
 *   - stream rule: the *_dev entry points of one context share its device scratch (count slab,
 *     ordering and singular-value scratch, the replicate's layout).  The library orders them in CALL
 *     order whatever their streams: a call on another stream than the previous *_dev call first makes
 *     its stream wait for that call's work, and a host-buffer call waits for every *_dev call made
 *     before it.  What the library cannot see is the caller's own use of the OUTPUT arrays of a *_dev
 *     call: read them on the stream they were produced on, or after synchronising with it.
 */
#ifndef TETRAD_HIP_H
#define TETRAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tq_ctx tq_ctx;

enum {
    TQ_OK = 0,
    TQ_ERR_INVALID_ARG = -1,   /* NULL pointer, negative size, taxon index >= T ... */
    TQ_ERR_NO_DEVICE = -2,     /* no HIP device / device is not usable             */
    TQ_ERR_HIP = -3,           /* a HIP runtime call failed (see tq_last_error)    */
    TQ_ERR_NO_DATA = -4,       /* tq_resolve* before tq_set_data                   */
    TQ_ERR_LOCUS_ORDER = -5,   /* subsample requested but a locus id re-appears
                                  after its run ended (see tq_set_data)            */
    TQ_ERR_OOM = -6
};

/* per-quartet flag bits written by tq_resolve* */
enum {
    TQ_FLAG_ZERO_DATA = 1,     /* no site counted: scores are 0.001, topology 0.
                                  (reference: unseeded np.random.randint(3),
                                  resolve_quartets.py:230-232)                      */
    TQ_FLAG_DEGENERATE = 2,    /* two lowest scores within 1e-9 * sigma_max: argmin is
                                  decided by SVD rounding noise in any implementation */
    TQ_FLAG_BAD_INDEX = 4,     /* a taxon index was >= T; row treated as zero-data   */
    TQ_FLAG_NO_CONVERGENCE = 8, /* a singular-value iteration hit its sweep cap; the row holds the
                                  unconverged values (reference: np.linalg.svd raises LinAlgError,
                                  resolve_quartets.py:242; the Python mirror raises it too)   */
    TQ_FLAG_INVALID_DIAGNOSTIC = 16 /* the row was made while a timing-diagnostic mode was set (tq_set_option
                                  "scan_method" 2..5 or "phases" 1 / 2): it is NOT a result.  Set on every row of
                                  such a call; a call without a flags array fails while such a mode is set; the
                                  Python mirror raises.                                        */
};

/* Create / destroy a context bound to HIP device `device_id`.
 * Replaces: the per-engine process state of the reference (an ipyparallel
 * engine that re-opens the HDF5 file on every call, resolve_quartets.py:33-35). */
int tq_create(tq_ctx **out, int device_id);
void tq_destroy(tq_ctx *ctx);
const char *tq_last_error(const tq_ctx *ctx);

/* Upload one replicate's genotype matrix and locus column (H2D once per
 * replicate, not once per chunk).
 * Replaces: `tmparr = io5["tmparr"][:]; tmpmap = io5["tmpmap"][:]`
 *           (resolve_quartets.py:33-35) and the `tmpmap[:, 0]` argument of
 *           resolve_quartets.py:221,223.
 *   tmparr  u8 [T,S] row-major; 0..3 = A,C,G,T; anything > 3 (78 = N) is missing
 *   locus   u32, element i at locus[i * locus_stride]  (pass tmpmap and stride 2
 *           to use tmpmap[:,0] without a copy)
 * Subsample mode requires every locus id to occupy one contiguous run of sites
 * (always true for the reference's writers: write_database.py:138-149,
 * jit/resample.py:58) and no id equal to 0xFFFFFFFF; otherwise tq_resolve*
 * with subsample != 0 returns TQ_ERR_LOCUS_ORDER.                                  */
int tq_set_data(tq_ctx *ctx, const uint8_t *tmparr, int64_t T, int64_t S,
                const uint32_t *locus, int64_t locus_stride);

/* Bootstrap replicates built on the device.
 * tq_set_source uploads, once per project, what the reference keeps in its HDF5 database for
 * resampling: `seqarr` (u8 [T,S0], ASCII bases incl. IUPAC two-base codes, N/'-' = missing;
 * write_database.py:157-159) and `spans` (i64 [nloci,2], [start,end) columns of each locus;
 * jit/get_spans.py:11-48).
 * tq_bootstrap replaces resample_tmp_database (run_inference.py:99-143): loci `lidxs` (host,
 * i64 [nloci], values in [0,nloci); the reference draws them with rng.choice(nloci, nloci,
 * replace=True) at :117) are concatenated with their columns shuffled (jit/resample.py:20-64),
 * IUPAC codes are resolved at random per cell (jit/resolve_ambigs.py:12-36), bases are recoded
 * 0..3 (:133-136) and the device layout is rebuilt -- the result is the resident replicate, as if
 * tq_set_data had been called with the resampled tmparr/tmpmap.  The two seeds take the place of
 * the two rng.integers(2**31) draws (:120,:123); the streams behind them are this engine's own
 * (counter-based), so replicates are distributed like the reference's, not identical to them.
 * *out_S receives the replicate's number of sites.
 * tq_get_data copies the resident replicate back in the reference's layout (tmparr u8 [T,S] with
 * 0..3 and 78 = missing; tmpmap u32 [S,2] = {locus ordinal, site index}); tq_data_shape gives T,S. */
int tq_set_source(tq_ctx *ctx, const uint8_t *seqarr, int64_t T, int64_t S0,
                  const int64_t *spans, int64_t nloci);
int tq_bootstrap(tq_ctx *ctx, const int64_t *lidxs, int64_t n, uint64_t seed_shuffle,
                 uint64_t seed_ambig, int64_t *out_S);
/* Same, enqueued on `stream` without waiting (the replicate length is computed on the host from the
 * spans): the replicate of iteration k+1 is built right behind the kernels of iteration k.  Resolve
 * calls for the new replicate must be ordered after it (same stream, or an event).              */
int tq_bootstrap_async(tq_ctx *ctx, const int64_t *lidxs, int64_t n, uint64_t seed_shuffle,
                       uint64_t seed_ambig, int64_t *out_S, void *stream);
int tq_get_data(tq_ctx *ctx, uint8_t *tmparr, uint32_t *tmpmap);
int tq_data_shape(tq_ctx *ctx, int64_t *T, int64_t *S);

/* Resolve Q quartets (host buffers, synchronous).
 * Replaces: new_infer_resolved_quartets(tmparr, tmpmap, quartets, subsample_snps)
 *           (resolve_quartets.py:191-265), which returns (quartets, rstat, rscor).
 *   quartets u32 [Q,4]  taxon indices (returned unchanged by the reference, :265)
 *   rstat    u32 [Q,2]  out: [:,0] topology 0/1/2 (argmin of scores, :251),
 *                            [:,1] number of counted sites (:226,:264)
 *   rscor    f64 [Q,3]  out: invariant score of the three flattenings (:246-248)
 *   flags    u8  [Q]    out, may be NULL: TQ_FLAG_* bits                           */
int tq_resolve(tq_ctx *ctx, const uint32_t *quartets, int64_t Q, int subsample,
               uint32_t *rstat, double *rscor, uint8_t *flags);

/* Page-locked host memory from a process-wide pool (blocks are recycled: pinning pages costs about
 * as much as a resolve call).  Result arrays (and quartet arrays) that live in such a block -- or in
 * any other page-locked memory -- are read / written by the copy engine directly, asynchronously under
 * the kernels; pageable arrays work too, through pinned staging pieces and one host memcpy per piece.
 * Blocks are independent of any context and may outlive it.  No reference counterpart (the reference
 * returns fresh NumPy arrays, resolve_quartets.py:253-265; the ctypes binding wraps such blocks as NumPy
 * arrays).                                                                                       */
int tq_host_alloc(int64_t bytes, void **out);
int tq_host_free(void *p);

/* Quartets already on the device, results to host arrays (synchronous): what bench.py times as one
 * step -- inputs resident in HBM, result D2H inside the step.
 * Replaces: the same function as tq_resolve.                                                     */
int tq_resolve_to_host(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, int subsample,
                       uint32_t *rstat, double *rscor, uint8_t *flags);

/* Same, device pointers in and out, asynchronous on `stream` (a hipStream_t; NULL = the
 * default stream).                                                                        */
int tq_resolve_dev(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, int subsample,
                   uint32_t *d_rstat, double *d_rscor, uint8_t *d_flags, void *stream);

/* Full-mode quartet generation on the device: resolves the quartets whose
 * lexicographic ranks are [first_rank, first_rank+Q) among C(T,4), i.e. what
 * islice(combinations(range(T),4), start, end) yields (combinations.py:40-55),
 * without any quartet H2D.  d_quartets (u32[Q,4], may be NULL) receives them.     */
int tq_resolve_range_dev(tq_ctx *ctx, uint64_t first_rank, int64_t Q, int subsample,
                         uint32_t *d_quartets, uint32_t *d_rstat, double *d_rscor,
                         uint8_t *d_flags, void *stream);

/* The two stages of tq_resolve_dev as separate calls, for callers that ship results in pieces (the
 * multi-GPU path gathers and copies piece i while piece i+1 is computed):
 *   tq_scan_dev  orders and scans quartets [0,Q) into the context's count slab (Q <= option "batch");
 *   tq_svd_dev   turns rows [q0, q0+n) of that scanned batch into results; d_rstat / d_rscor / d_flags
 *                point at the outputs OF ROW q0 (so pieces can live in separate slabs).
 * Both enqueue on `stream`.  Replaces: the two halves of new_infer_resolved_quartets
 * (resolve_quartets.py:208-226 and :236-251).                                                    */
int tq_scan_dev(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, int subsample, void *stream);
int tq_svd_dev(tq_ctx *ctx, int64_t q0, int64_t n, uint32_t *d_rstat, double *d_rscor,
               uint8_t *d_flags, void *stream);

/* Random-mode quartet generation on the device: unranks host-sampled
 * lexicographic ranks (combinations.py:94-114) into d_quartets.                    */
int tq_unrank_dev(tq_ctx *ctx, const uint64_t *d_ranks, int64_t Q, uint32_t *d_quartets,
                  void *stream);

/* Opt-in quartet sampler on the device: Q DISTINCT quartets drawn uniformly from the C(T,4) of the resident
 * (or source) matrix, in random order, written to d_quartets (u32[Q,4]) and -- if d_ranks is not NULL -- their
 * lexicographic ranks to d_ranks (u64[Q]).  Same distribution as
 * random_combination_sample_via_index (combinations.py:109-114: rng.choice(C(T,4), size, replace=False)
 * + unranking) but NOT the project Generator's stream (a keyed permutation of the rank space,
 * counter-based): for runs that do not need to reproduce a reference run's sample.               */
int tq_sample_quartets_dev(tq_ctx *ctx, uint64_t seed, int64_t Q, uint64_t *d_ranks,
                           uint32_t *d_quartets, void *stream);

/* Kernel-level outputs for parity tests (host buffers, synchronous); any of the
 * debug pointers may be NULL.
 * Replaces: subsample_chunk_to_matrices / full_chunk_to_matrices
 *           (resolve_quartets.py:42-104) -> cmats u32[Q,3,16,16];
 *           np.linalg.svd(...)[1] (:242) -> svds f64[Q,3,16] (descending);
 *           np.linalg.matrix_rank (:243) -> ranks i32[Q,3].                        */
int tq_resolve_debug(tq_ctx *ctx, const uint32_t *quartets, int64_t Q, int subsample,
                     uint32_t *rstat, double *rscor, uint8_t *flags,
                     uint32_t *cmats, double *svds, int32_t *ranks);

/* HIP-event timing of the two resolve kernels, on the stream they were launched on.
 * tq_timing_enable(ctx, 1) makes every subsequent resolve call record HIP events
 * around its kernels; tq_timing_read synchronises those events, returns the summed
 * kernel milliseconds and the number of resolve calls since the last reset, and resets. */
int tq_timing_enable(tq_ctx *ctx, int on);
int tq_timing_read(tq_ctx *ctx, double *kernel_ms, int64_t *launches);
/* Same, split by kernel: total = scan + svd milliseconds summed over the `calls` resolve calls
 * made since the last read (a resolve call launches one scan + one SVD kernel per batch).     */
int tq_timing_read_split(tq_ctx *ctx, double *total_ms, double *scan_ms, double *svd_ms, int64_t *calls);
/* Same, per kernel: ms[0] ordering (key + radix sort), ms[1] site scan, ms[2] bidiagonalisation (or the
 * whole Jacobi kernel with svd_method 0), ms[3] bidiagonal QR, ms[4] scores; entries beyond n_ms are
 * not written, entries beyond 4 are set to 0.                                                    */
int tq_timing_read_kernels(tq_ctx *ctx, double *ms, int n_ms, int64_t *calls);

/* Tuning / diagnostic knobs (value 0 = library default unless noted).  Returns TQ_OK or an error.
 * Names: nrep, waves_per_cu, wg_min_quartets (batches below it -- default 4096 -- are scanned by the one-wave-per-quartet kernel:
 * small calls are latency-bound), batch (quartets per internal batch, default 2^23; device scratch is about
 * 3.2 KB per quartet of the largest batch resolved so far), order, scan_wg (waves per scan workgroup: 1, 2, 4, 8, 16), scan_method (-1 auto), svd_method (0 Jacobi,
 * 1 Householder+QR), bidiag_layout (1, default: the bidiagonalisation deals a matrix 2 x 2 over four lanes; 0: four column
 * groups), xcd_remap (1: scan workgroups of one XCD take a contiguous part of the sorted order),
 * svd_wpc (blocks per CU of the singular-value grids, 0 = one pass per block), svd_chunk (quartets per
 * pass of the singular-value stage = per result-copy piece, default 2^18), svd_streams (1 or 2: chunks alternate
 * between two streams so that one chunk's tail is filled by the next chunk; default 2), share_c (1: scan variant that also shares row c inside a
 * workgroup; measured slower, off), park_t (1, default: transposed, bank-conflict-free pattern park of the set-bit walk; 0: lane-
 * contiguous park), scan_pair (1: two quartets per wavefront; measured slower, off), count_invariant (1: sites whose four bases
 * are equal are counted too -- what the reference's count kernels do when their caller's mask leaves such a site open,
 * resolve_quartets.py:59-64; one-wave kernel), bdsqr_maxit (QR sweeps per singular value before TQ_FLAG_NO_CONVERGENCE,
 * default 60), bdsqr_stats (1: count rotation steps / issued lane-slots, read with tq_debug_fetch which = 3), phases
 * (timing diagnostics).  scan_method 2..5 and phases 1 / 2 are timing diagnostics whose rows are wrong: every row of a
 * call made under them carries TQ_FLAG_INVALID_DIAGNOSTIC and a call without a flags array fails.  scan_method 6 = the
 * bank-private counter kernel (scan_pb.hpp; an A/B form, slower).  batch is clamped to 2^31 - 1.
 * scan_dp (1, default: full-mode batches -- subsample = 0 -- of at least dp_min_quartets go to the joint-histogram scan,
 * scan_dp.hpp: two quartets that share their first three taxa per wavefront, one LDS atomic per site and pair; the same rows,
 * bit for bit; 0: always one quartet per wavefront), dp_min_quartets (default 32 768; smaller batches are not sorted and hold
 * few pairs), scan_f4 (-1, default: in subsample mode the cooperative scan streams a wavefront's own rows as 12-byte
 * {missing, bit 0, bit 1} records only and makes the pattern of a counted site inside the histogram walk, scan_f4.hpp =
 * SURVEY 8 row f4; 1: in full mode too (slower there); 0: the nibble-code kernel of rounds 1-3 everywhere; NOTE: 0 is a value
 * of its own for this option, the default is -1).                                                                   */
int tq_set_option(tq_ctx *ctx, const char *name, int64_t value);

/* Test hook: copy the scratch of the last resolve call to the host.  which = 0: count slab
 * u32[n][256] of the last scan batch; 1: bidiagonals f64[3m][32] (d[16], e[16]); 2: singular values
 * f64[3m][16] (unsorted, sign bit = not converged), m = quartets of the last singular-value chunk; 3: the 8 u64 counters
 * of option bdsqr_stats.  No reference counterpart.                                                                    */
int tq_debug_fetch(tq_ctx *ctx, int which, void *dst, int64_t bytes);

/* Test hook: run the bidiagonal-QR kernel (tq_bdsqr_kernel) alone on nmat bidiagonals given on the host -- de f64[nmat][32]
 * as tq_debug_fetch(which = 1) hands them out, in ANY order: lane i of wave w gets matrix 64 w + i -- and return the singular
 * values sv f64[nmat][16] (unsorted; may be NULL), per matrix its rotation steps | sweeps << 16 (work u32[nmat]; may be NULL)
 * and the mean duration of `reps` launches in ms.  Used to measure what ordering the matrices into waves is worth
 * (tools/bdsqr_order.py).  No reference counterpart.                                                                */
int tq_debug_bdsqr(tq_ctx *ctx, const double *de, int64_t nmat, double *sv, uint32_t *work, int reps, double *ms);

/* Text for the consumers right after the hot path (host code, no device involved).
 * tq_format_tsv writes the rows the reference appends to <name>.quartets_<rep>.tsv
 *   (run_inference.py:233-234: pd.concat([rqrts, rscor, rstat], axis=1).to_csv(sep="\t",
 *   float_format='%.6f', index=False, header=False)): "a\tb\tc\td\ts0\ts1\ts2\ttopo\tnsnps\n".
 * tq_format_qmc writes the wQMC input lines "a,b|c,d:weight\n" of iter_qmc_formatted
 *   (run_inference.py:254-305) for the rows that pass min_snps (:258,:275) and min_ratio (:300), with the
 *   weight strategies 0..3 (:280-297) computed, as the reference does, from the scores as they read
 *   back from the TSV (rounded to 6 decimals); *n_lines = lines written.  The reference then shuffles
 *   the file with `shuf` (:326-327); that is left to the caller.
 * Both return TQ_OK with *written = bytes produced (no terminating 0), TQ_ERR_OOM with *written = a
 * buffer size that is sufficient when `cap` was too small, TQ_ERR_INVALID_ARG otherwise.            */
int tq_format_tsv(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q,
                  char *out, int64_t cap, int64_t *written);
int tq_format_qmc(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q,
                  int weights, int64_t min_snps, double min_ratio, char *out, int64_t cap,
                  int64_t *written, int64_t *n_lines);

/* The reference's quartet sample, stream-identical and faster (host code): what
 * `Generator.choice(pop, size, replace=False)` (combinations.py:113) returns when size > pop // 50 and
 * pop > 10 000 -- NumPy's tail shuffle of arange(pop) -- drawn from the SAME NumPy bit generator, passed as the
 * address NumPy publishes in `rng.bit_generator.ctypes.bit_generator` (caller holds `bit_generator.lock`).
 * The Generator is left in the state NumPy's own call would leave it in.  pop <= 2^32 - 2.        */
int tq_numpy_choice_tail(void *np_bitgen, uint64_t pop, int64_t size, int64_t *out);

/* Lexicographic unranking on the host (no device involved): quartets[i] = the 4-combination of range(T) with
 * rank ranks[i] (or first_rank + i when ranks is NULL), i.e. _index_to_combination (combinations.py:94-106) for
 * every sampled index / islice(combinations(range(T), 4), first_rank, first_rank + Q) (combinations.py:40-55).
 * TQ_ERR_INVALID_ARG when a rank is >= C(T,4).                                                     */
int tq_unrank(const uint64_t *ranks, uint64_t first_rank, int64_t Q, int64_t T, uint32_t *quartets);

/* The rows tq_format_qmc would write, as arrays (no text round trip): splits u32[n,4] = "a,b|c,d" and weights f64[n]
 * (the value of the line's "%.5f" text), n = *n_rows <= Q; splits / weights must hold Q rows.  Input for tq_qmc_tree.  */
int tq_qmc_splits(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q, int weights,
                  int64_t min_snps, double min_ratio, uint32_t *splits, double *weights_out, int64_t *n_rows);

/* Quartet supertree (host code, no device involved): weighted Quartet MaxCut over `n` resolved quartets
 * splits u32[n,4] = "a,b|c,d" (the taxa of a wQMC input line, run_inference.py:264-305), weights f64[n] or
 * NULL (all 1), taxa 0..ntaxa-1.  Writes the unrooted tree as newick with the taxon numbers as tip labels
 * ("((0,1),(2,3),4);"), *written = its length (TQ_ERR_OOM with the needed size when cap is too small).
 * Takes the place of: `bin/max-cut-tree qrtt=.. weights=on|off otre=..` as called by run_qmc
 * (run_inference.py:146-166).  That binary ships without source and is never run or read here: this is an
 * implementation from the published method, and parity with it is UNPINNED by construction.        */
int tq_qmc_tree(const uint32_t *splits, const double *weights, int64_t n, int64_t ntaxa, uint64_t seed,
                char *out, int64_t cap, int64_t *written);

/* Device facts used by bench.py: writes CU count, wave slots used by the resolve
 * kernel per CU and the padded row pitch in bytes.                                 */
int tq_device_info(tq_ctx *ctx, int32_t *num_cu, int32_t *waves_per_cu, int64_t *row_pitch);

#ifdef __cplusplus
}
#endif
#endif /* TETRAD_HIP_H */
