// tetrad_hip.hip -- MI355X (gfx950 / CDNA4) quartet-invariant engine.
//
// Hand-written HIP for the per-quartet hot path of eaton-lab/tetrad
// (reference: tetrad/src/resolve_quartets.py:191-265 and the count kernels
// :42-104).  Not a translation: the reference is an interpreted per-quartet
// loop around a serial site scan and six LAPACK calls; here one 64-lane
// wavefront owns a quartet during the site scan (four neighbours of the sorted
// order form a workgroup that shares what depends on their common taxa a,b),
// four lanes own a matrix during bidiagonalisation and one lane owns it during
// the QR iteration.
//
// Data layout in HBM (built once per replicate by tq_set_data / tq_bootstrap):
//   rows    u8   [T][Sp]   base code 0..3 per site, missing/pad -> 0  (Sp = S rounded up to 2048;
//                          inside each 2048-site step the bytes sit in two 1 KiB panels, row_offset())
//   nib     u8   [T][Sp/2] the same codes, two per byte (nib_offset()); nib4: codes pre-multiplied by 4
//   planes  u32x4[T][W]    per 32 sites: {missing bits, base bit 0, base bit 1, run-begin bits}, W = Sp/32
//   planes3 u32x3[T][W]    compact copy {missing, bit 0, bit 1}; runbeg u32 [W] run-begin bits, stored once
//
// Kernels (each in its own header of this directory, all included below into one translation unit):
//   prepare.hpp   layout build, lexicographic unranking, sort keys
//   scan.hpp      tq_scan_wg_kernel / tq_scan_kernel: site scan -> 256 pattern counts per quartet
//   hqr.hpp       tq_bidiag_kernel + tq_bdsqr_kernel + tq_score_kernel: singular values (default)
//   jacobi.hpp    tq_svd_kernel: one-sided Jacobi singular values in registers (alternative)
//   bootstrap.hpp tq_boot_*: bootstrap replicate built on the device
// This file holds the context, the launch logic and the C ABI (include/tetrad_hip.h).
//
// Bounds: the scan is L2 / LDS-atomic / VALU work on a <= 40 MB resident matrix (HBM only on
// first touch), the singular-value stage is f64 VALU.  Algorithmic bytes per quartet: 4*S + 48
// (SURVEY.md 8d).  DESIGN.md section 4 has the per-kernel description and measurements.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <utility>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/tetrad_hip.h"

namespace {

#include "common.hpp"
#include "prepare.hpp"
#include "scan.hpp"
#include "jacobi.hpp"
#include "hqr.hpp"
#include "bootstrap.hpp"
#include "format.hpp"

}  // namespace

// ======================================================================================
// host side: context + C ABI
// ======================================================================================
struct tq_ctx {
    int device = 0;
    std::string err;
    hipDeviceProp_t prop{};
    // replicate data
    int64_t T = 0, S = 0, Sp = 0, W = 0;
    uint8_t *d_rows = nullptr;
    uint8_t *d_nib = nullptr;       // [T][Sp/2] nibble-packed copy of the rows (common.hpp: nib_offset)
    uint4 *d_planes = nullptr;      // [T][W] {miss, p0, p1, runbeg}
    uint32_t *d_planes3 = nullptr;  // [T][W][3] {miss, p0, p1}, then runbeg [W]
    bool have_data = false;
    bool locus_runs_ok = false;
    int64_t plane_cap_W = 0;        // W the planes3 / runbeg allocation was sized for
    int64_t data_capacity = 0;      // allocated Sp (rows/planes are re-used by bootstrap replicates)
    // bootstrap source (tq_set_source): ASCII seqarr [T][S0], spans i64 [nloci][2]
    uint8_t *d_seqarr = nullptr;
    int64_t *d_spans = nullptr;
    int64_t src_T = 0, src_S0 = 0, nloci = 0, max_width = 0;
    int64_t *d_lidxs = nullptr;     // [nloci]
    uint32_t *d_boot = nullptr;     // widths/offsets [nloci+1] | src_col [cap] | site_locus [cap]
    int64_t boot_cap = 0;
    void *d_boot_tmp = nullptr;
    size_t boot_tmp_bytes = 0;
    // scratch for the host-buffer API
    void *d_scratch = nullptr;
    void *h_stage = nullptr;        // pinned staging for small host-buffer calls
    size_t stage_bytes = 0;
    size_t scratch_bytes = 0;
    // count-matrix slab between the two kernels: u32 [batch][256]
    uint32_t *d_cm = nullptr;
    int64_t cm_quartets = 0;
    int cm_slabs = 0;
    // (a,b)-sorted processing order of the current batch: keys/idx in, keys/idx out, cub temp
    uint32_t *d_sort = nullptr;     // 4 arrays of cm_quartets u32
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int order = 1;                  // 1 = process quartets in (a,b)-sorted order
    // HQR path scratch: de f64[3*batch][32], sv f64[3*batch][16], nsnps u32[batch]
    double *d_de = nullptr, *d_sv = nullptr;
    uint32_t *d_nsnps = nullptr;
    int svd_method = 1;             // 0 = one-sided Jacobi (tq_svd_kernel), 1 = Householder + bidiagonal QR
    int scan_wg = 4;                // waves per workgroup of the cooperative scan kernel (1 = one wave per quartet)
    // software pipeline across sub-batches: scan of sub-batch i+1 runs on a second stream beside the
    // singular-value stage of sub-batch i (0 = off: one stage after the other on the caller's stream)
    int64_t overlap = 0;            // sub-batch size in quartets
    bool input_sorted = false;      // set by the host-buffer entry points when the quartets already are in key order
    int xcd_remap = 1;              // 1: scan workgroups of one XCD take a contiguous part of the sorted order
    int svd_wpc = 0;                // blocks per CU of the bidiag / bdsqr grids (0 = one pass per block)
    int ov_scan_wgs = 1;            // scan workgroups per CU while overlapping
    int ov_svd_waves = 6;           // singular-value-stage waves per CU while overlapping
    hipStream_t sA = nullptr, sB = nullptr;
    hipEvent_t evIn = nullptr, evA[2] = {nullptr, nullptr}, evB[2] = {nullptr, nullptr}, evEndA = nullptr, evEndB = nullptr;
    int wpc_override = 0;           // transient: grid_for() uses this instead of waves_per_cu when > 0
    // options
    int nrep = 1;
    int waves_per_cu = 0;           // 0 = from the occupancy query
    int phases = 3;                 // diagnostics only: 1 = scan kernel only, 2 = SVD kernel only
    int scan_method = -1;           // 0 = EXEC-masked slot per site, 1 = set-bit walk, -1 = 1 if subsample else 0
    int64_t batch = 1 << 23;        // quartets per scan->svd batch (<= 8 GiB count slab + 9 GiB bidiagonals / values)
    // timing: per resolve call one (start, mid, stop) triple per batch
    bool timing = false;
    struct Ev { hipEvent_t e0, e1, e2; };
    std::vector<Ev> events;
    size_t events_used = 0;
    int64_t timed_calls = 0;
};

namespace {

std::string g_create_err;

int fail(tq_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_err = buf;
    return code;
}

#define TQ_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, e_ == hipErrorOutOfMemory ? TQ_ERR_OOM : TQ_ERR_HIP, "%s failed: %s", \
                        #call, hipGetErrorString(e_));                                        \
    } while (0)

void free_data(tq_ctx *ctx)
{
    if (ctx->d_rows) (void)hipFree(ctx->d_rows);
    if (ctx->d_nib) (void)hipFree(ctx->d_nib);
    if (ctx->d_planes) (void)hipFree(ctx->d_planes);
    if (ctx->d_planes3) (void)hipFree(ctx->d_planes3);
    ctx->d_rows = nullptr;
    ctx->d_nib = nullptr;
    ctx->d_planes = nullptr;
    ctx->d_planes3 = nullptr;
    ctx->have_data = false;
    ctx->data_capacity = 0;
}

void free_source(tq_ctx *ctx)
{
    if (ctx->d_seqarr) (void)hipFree(ctx->d_seqarr);
    if (ctx->d_spans) (void)hipFree(ctx->d_spans);
    if (ctx->d_lidxs) (void)hipFree(ctx->d_lidxs);
    if (ctx->d_boot) (void)hipFree(ctx->d_boot);
    if (ctx->d_boot_tmp) (void)hipFree(ctx->d_boot_tmp);
    ctx->d_seqarr = nullptr;
    ctx->d_spans = nullptr;
    ctx->d_lidxs = nullptr;
    ctx->d_boot = nullptr;
    ctx->d_boot_tmp = nullptr;
    ctx->boot_cap = 0;
    ctx->nloci = 0;
}

int ensure_scratch(tq_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return TQ_OK;
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    TQ_HIP(ctx, hipMalloc(&ctx->d_scratch, bytes));
    ctx->scratch_bytes = bytes;
    return TQ_OK;
}

int ensure_stage(tq_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->stage_bytes) return TQ_OK;
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->stage_bytes = 0;
    TQ_HIP(ctx, hipHostMalloc(&ctx->h_stage, bytes, hipHostMallocDefault));
    ctx->stage_bytes = bytes;
    return TQ_OK;
}

int ensure_cm(tq_ctx *ctx, int64_t quartets, int slabs = 1)
{
    if (quartets <= ctx->cm_quartets && slabs <= ctx->cm_slabs) return TQ_OK;
    if (quartets < ctx->cm_quartets) quartets = ctx->cm_quartets;
    if (slabs < ctx->cm_slabs) slabs = ctx->cm_slabs;
    if (ctx->d_cm) (void)hipFree(ctx->d_cm);
    if (ctx->d_sort) (void)hipFree(ctx->d_sort);
    if (ctx->d_sort_tmp) (void)hipFree(ctx->d_sort_tmp);
    if (ctx->d_de) (void)hipFree(ctx->d_de);
    if (ctx->d_sv) (void)hipFree(ctx->d_sv);
    if (ctx->d_nsnps) (void)hipFree(ctx->d_nsnps);
    ctx->d_cm = nullptr;
    ctx->d_sort = nullptr;
    ctx->d_sort_tmp = nullptr;
    ctx->d_de = ctx->d_sv = nullptr;
    ctx->d_nsnps = nullptr;
    ctx->cm_quartets = 0;
    ctx->cm_slabs = 0;
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_cm, (size_t)quartets * 1024 * (size_t)slabs));   // second slab: overlap mode only
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_de, (size_t)quartets * 3 * 32 * sizeof(double)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_sv, (size_t)quartets * 3 * 16 * sizeof(double)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_nsnps, (size_t)quartets * sizeof(uint32_t)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_sort, (size_t)quartets * 16));
    size_t tmp = 0;
    uint32_t *k = ctx->d_sort;
    TQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, k, k, k, k, (int)quartets));
    TQ_HIP(ctx, hipMalloc(&ctx->d_sort_tmp, tmp ? tmp : 16));
    ctx->sort_tmp_bytes = tmp;
    ctx->cm_quartets = quartets;
    ctx->cm_slabs = slabs;
    return TQ_OK;
}

// order[] for one batch: indices sorted by (first taxon, second taxon); nullptr = natural order
int make_order(tq_ctx *ctx, const uint32_t *dq, int64_t n, hipStream_t stream, const uint32_t **order)
{
    *order = nullptr;
    if (!ctx->order || n < 1024 || ctx->T > 65535 || ctx->input_sorted) return TQ_OK;
    uint32_t *keys_in = ctx->d_sort, *idx_in = keys_in + ctx->cm_quartets;
    uint32_t *keys_out = idx_in + ctx->cm_quartets, *idx_out = keys_out + ctx->cm_quartets;
    const uint64_t T = (uint64_t)ctx->T;
    const int with_c = T * T * T <= 0xFFFFFFFFull;
    hipLaunchKernelGGL(tq_key_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dq, n,
                       (uint32_t)ctx->T, with_c, keys_in, idx_in);
    TQ_HIP(ctx, hipGetLastError());
    int bits = 1;
    while (bits < 32 && (1ull << bits) < (with_c ? T * T * T : T * T)) ++bits;
    size_t tmp = ctx->sort_tmp_bytes;
    TQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->d_sort_tmp, tmp, keys_in, keys_out, idx_in, idx_out, (int)n,
                                                   0, bits, stream));
    *order = idx_out;
    return TQ_OK;
}

DevData dev_data(const tq_ctx *ctx)
{
    DevData d;
    d.rows = ctx->d_rows;
    d.nib = ctx->d_nib;
    d.nib4 = ctx->d_nib + (size_t)ctx->T * (size_t)ctx->data_capacity / 2;
    d.planes = ctx->d_planes;
    d.planes3 = ctx->d_planes3;
    d.runbeg = ctx->d_planes3 + (size_t)ctx->T * (size_t)ctx->plane_cap_W * 3;
    d.pitch = ctx->Sp;
    d.W = ctx->W;
    d.T = (int32_t)ctx->T;
    d.ntiles = (int32_t)(ctx->Sp / TILE);
    return d;
}

constexpr size_t STAGE_LIMIT = 2u << 20;     // bytes of quartets + results that go through pinned staging

template <typename K>
int grid_for(tq_ctx *ctx, K kern, int64_t items, int64_t *grid, int wpc_kernel = 0)
{
    int wpc = wpc_kernel > 0 ? wpc_kernel : ctx->wpc_override > 0 ? ctx->wpc_override : ctx->waves_per_cu;
    if (wpc <= 0) {
        int nb = 0;
        TQ_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, WAVE, 0));
        wpc = nb > 0 ? nb : 8;
    }
    int64_t g = (int64_t)ctx->prop.multiProcessorCount * wpc;
    if (g > items) g = items;
    if (g < 1) g = 1;
    *grid = g;
    return TQ_OK;
}

template <int NREP, bool SUB, int METHOD>
int launch_scan(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, hipStream_t stream)
{
    auto kern = tq_scan_kernel<NREP, SUB, METHOD>;
    int64_t grid;
    int rc = grid_for(ctx, kern, Q, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVE), 0, stream, dev_data(ctx), dq, order, Q, ctx->d_cm);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

template <bool SUB, int METHOD, int NW>
int launch_scan_wg(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, hipStream_t stream)
{
    auto kern = tq_scan_wg_kernel<SUB, METHOD, NW>;
    int wgs = ctx->wpc_override > 0 ? (ctx->wpc_override + NW - 1) / NW
              : ctx->waves_per_cu > 0 ? (ctx->waves_per_cu + NW - 1) / NW : 0;
    // default: one block of NW quartets per workgroup, dispatched in sorted order.  Workgroups that
    // run at the same time are then neighbours of the (a,b) order (their shared rows are L2 hits) and
    // the dispatcher balances the load; a persistent grid-stride loop was 13 % slower.
    const int64_t nblk = (Q + NW - 1) / NW;
    int64_t grid = wgs > 0 ? (int64_t)ctx->prop.multiProcessorCount * wgs : nblk;
    int64_t xcd_chunk = 0;
    if (grid >= nblk) {
        grid = nblk;
        if (ctx->xcd_remap && nblk >= 64) {                  // one block per workgroup, XCD-contiguous
            xcd_chunk = (nblk + 7) / 8;
            grid = xcd_chunk * 8;
        }
    }
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * WAVE), 0, stream, dev_data(ctx), dq, order, Q,
                       ctx->d_cm, xcd_chunk);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

int launch_scan_n(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, int subsample,
                  hipStream_t stream)
{
    if (ctx->scan_wg >= 2 && Q >= 64 && (uint64_t)ctx->T * (uint64_t)ctx->Sp < 0xFFFF0000ull) {
        const int m = ctx->scan_method < 0 ? (subsample ? 1 : 0) : ctx->scan_method;
#define TQ_WG_CASE(NW)                                                                                   \
    if (ctx->scan_wg == NW) {                                                                            \
        if (m == 2) return launch_scan_wg<true, 2, NW>(ctx, dq, order, Q, stream);                       \
        if (m == 3) return launch_scan_wg<true, 3, NW>(ctx, dq, order, Q, stream);                       \
        if (subsample)                                                                                   \
            return m ? launch_scan_wg<true, 1, NW>(ctx, dq, order, Q, stream)                            \
                     : launch_scan_wg<true, 0, NW>(ctx, dq, order, Q, stream);                           \
        return m ? launch_scan_wg<false, 1, NW>(ctx, dq, order, Q, stream)                               \
                 : launch_scan_wg<false, 0, NW>(ctx, dq, order, Q, stream);                              \
    }
        TQ_WG_CASE(2)
        TQ_WG_CASE(4)
        TQ_WG_CASE(16)
        TQ_WG_CASE(8)
#undef TQ_WG_CASE
    }
#define TQ_SCAN_CASE(N)                                                                              \
    case N:                                                                                          \
        if (method == 0)                                                                             \
            return subsample ? launch_scan<N, true, 0>(ctx, dq, order, Q, stream)                    \
                             : launch_scan<N, false, 0>(ctx, dq, order, Q, stream);                  \
        return subsample ? launch_scan<N, true, 1>(ctx, dq, order, Q, stream)                        \
                         : launch_scan<N, false, 1>(ctx, dq, order, Q, stream)
    const int method = ctx->scan_method < 0 ? (subsample ? 1 : 0) : ctx->scan_method;
    switch (ctx->nrep) {
        TQ_SCAN_CASE(2);
        TQ_SCAN_CASE(4);
        TQ_SCAN_CASE(8);
        TQ_SCAN_CASE(16);
        TQ_SCAN_CASE(32);
    default:
        TQ_SCAN_CASE(1);
    }
#undef TQ_SCAN_CASE
}

template <bool DEBUG>
int launch_svd(tq_ctx *ctx, const uint32_t *dq, int64_t Q, const OutPtrs &out, hipStream_t stream)
{
    auto kern = tq_svd_kernel<DEBUG>;
    int64_t grid;
    int rc = grid_for(ctx, kern, (Q + QPW - 1) / QPW, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVE), 0, stream, (const uint32_t *)ctx->d_cm, dq, Q,
                       (int32_t)ctx->T, out);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

template <bool DEBUG>
int launch_hqr(tq_ctx *ctx, const uint32_t *dq, int64_t Q, const OutPtrs &out, hipStream_t stream)
{
    int64_t grid;
    auto k1 = tq_bidiag_kernel<DEBUG>;
    // one pass per block unless told otherwise: the work per pass varies (QR iterations), and the
    // hardware dispatcher balances it better than a static grid-stride loop (3.8 ms vs 5.1 ms per 1e6)
    const int svd_wpc = ctx->svd_wpc > 0 ? ctx->svd_wpc : (1 << 20);
    int rc = grid_for(ctx, k1, (Q + 15) / 16, &grid, svd_wpc);
    if (rc) return rc;
    hipLaunchKernelGGL(k1, dim3((unsigned)grid), dim3(WAVE), 0, stream, (const uint32_t *)ctx->d_cm, Q, ctx->d_de,
                       ctx->d_nsnps, out.cmats);
    TQ_HIP(ctx, hipGetLastError());
    const int64_t nmat = 3 * Q;
    rc = grid_for(ctx, tq_bdsqr_kernel, (nmat + WAVE - 1) / WAVE, &grid, svd_wpc);
    if (rc) return rc;
    hipLaunchKernelGGL(tq_bdsqr_kernel, dim3((unsigned)grid), dim3(WAVE), 0, stream, (const double *)ctx->d_de, nmat,
                       ctx->d_sv);
    TQ_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(tq_score_kernel<DEBUG>, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, stream,
                       (const double *)ctx->d_sv, (const uint32_t *)ctx->d_nsnps, dq, Q, (int32_t)ctx->T, out);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

OutPtrs offset_out(const OutPtrs &o, int64_t q0)
{
    OutPtrs r = o;
    r.rstat = o.rstat + q0 * 2;
    r.rscor = o.rscor + q0 * 3;
    r.flags = o.flags ? o.flags + q0 : nullptr;
    r.cmats = o.cmats ? o.cmats + q0 * 768 : nullptr;
    r.svds = o.svds ? o.svds + q0 * 48 : nullptr;
    r.ranks = o.ranks ? o.ranks + q0 * 3 : nullptr;
    return r;
}

int launch_overlapped(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, const OutPtrs &out,
                      hipStream_t stream);

// scan kernel -> cm slab -> SVD kernel, in batches so that the slab stays <= batch KiB
int launch(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, bool debug, const OutPtrs &out,
           hipStream_t stream)
{
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (subsample && !ctx->locus_runs_ok)
        return fail(ctx, TQ_ERR_LOCUS_ORDER,
                    "subsample mode needs each locus id in one contiguous run of sites (and no id 0xFFFFFFFF)");
    if (Q == 0) return TQ_OK;
    if (ctx->overlap > 0 && !debug && ctx->phases == 3 && Q > ctx->overlap)
        return launch_overlapped(ctx, dq, Q, subsample, out, stream);
    const int64_t batch = Q < ctx->batch ? Q : ctx->batch;
    int rc = ensure_cm(ctx, batch);
    if (rc) return rc;
    if (ctx->timing) ctx->timed_calls++;
    for (int64_t q0 = 0; q0 < Q; q0 += batch) {
        const int64_t n = (Q - q0) < batch ? (Q - q0) : batch;
        tq_ctx::Ev ev{};
        if (ctx->timing) {
            if (ctx->events_used == ctx->events.size()) {
                tq_ctx::Ev e{};
                TQ_HIP(ctx, hipEventCreate(&e.e0));
                TQ_HIP(ctx, hipEventCreate(&e.e1));
                TQ_HIP(ctx, hipEventCreate(&e.e2));
                ctx->events.push_back(e);
            }
            ev = ctx->events[ctx->events_used++];
            TQ_HIP(ctx, hipEventRecord(ev.e0, stream));
        }
        if (ctx->phases & 1) {
            const uint32_t *order = nullptr;
            rc = make_order(ctx, dq + q0 * 4, n, stream, &order);
            if (rc) return rc;
            rc = launch_scan_n(ctx, dq + q0 * 4, order, n, subsample, stream);
            if (rc) return rc;
        }
        if (ctx->timing) TQ_HIP(ctx, hipEventRecord(ev.e1, stream));
        if (ctx->phases & 2) {
            const OutPtrs o = offset_out(out, q0);
            if (ctx->svd_method == 0)
                rc = debug ? launch_svd<true>(ctx, dq + q0 * 4, n, o, stream)
                           : launch_svd<false>(ctx, dq + q0 * 4, n, o, stream);
            else
                rc = debug ? launch_hqr<true>(ctx, dq + q0 * 4, n, o, stream)
                           : launch_hqr<false>(ctx, dq + q0 * 4, n, o, stream);
            if (rc) return rc;
        }
        if (ctx->timing) TQ_HIP(ctx, hipEventRecord(ev.e2, stream));
    }
    return TQ_OK;
}

// Overlapped form of launch(): the batch is cut into sub-batches; scan(i+1) runs on stream A beside
// the singular-value stage of sub-batch i on stream B (two count slabs).  Both grids are sized to a
// fraction of each CU so that the two stages are co-resident: the scan is an L2/LDS/integer mix at
// ~50 % VALU, the bidiagonal QR a latency-bound f64 chain at ~40 % -- they fill each other's gaps.
int launch_overlapped(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, const OutPtrs &out,
                      hipStream_t stream)
{
    const int64_t sub = ctx->overlap;
    int rc = ensure_cm(ctx, sub, 2);
    if (rc) return rc;
    if (!ctx->sA) {
        TQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->sA, hipStreamNonBlocking));
        TQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->sB, hipStreamNonBlocking));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evIn, hipEventDisableTiming));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evEndA, hipEventDisableTiming));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evEndB, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) {
            TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evA[i], hipEventDisableTiming));
            TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evB[i], hipEventDisableTiming));
        }
    }
    tq_ctx::Ev ev{};
    if (ctx->timing) {
        ctx->timed_calls++;
        if (ctx->events_used == ctx->events.size()) {
            tq_ctx::Ev e{};
            TQ_HIP(ctx, hipEventCreate(&e.e0));
            TQ_HIP(ctx, hipEventCreate(&e.e1));
            TQ_HIP(ctx, hipEventCreate(&e.e2));
            ctx->events.push_back(e);
        }
        ev = ctx->events[ctx->events_used++];
        TQ_HIP(ctx, hipEventRecord(ev.e0, stream));
        TQ_HIP(ctx, hipEventRecord(ev.e1, stream));       // stages overlap: only the total is meaningful
    }
    TQ_HIP(ctx, hipEventRecord(ctx->evIn, stream));
    TQ_HIP(ctx, hipStreamWaitEvent(ctx->sA, ctx->evIn, 0));
    TQ_HIP(ctx, hipStreamWaitEvent(ctx->sB, ctx->evIn, 0));
    uint32_t *cm0 = ctx->d_cm;
    int64_t i = 0;
    for (int64_t q0 = 0; q0 < Q; q0 += sub, ++i) {
        const int64_t n = (Q - q0) < sub ? (Q - q0) : sub;
        const int b = (int)(i & 1);
        // stream A: ordering + scan into slab b (after the SVD stage that last read slab b)
        if (i >= 2) TQ_HIP(ctx, hipStreamWaitEvent(ctx->sA, ctx->evB[b], 0));
        ctx->d_cm = cm0 + (size_t)b * (size_t)ctx->cm_quartets * 256;
        const uint32_t *order = nullptr;
        rc = make_order(ctx, dq + q0 * 4, n, ctx->sA, &order);
        if (!rc) {
            ctx->wpc_override = ctx->ov_scan_wgs * (ctx->scan_wg > 1 ? ctx->scan_wg : 8);
            rc = launch_scan_n(ctx, dq + q0 * 4, order, n, subsample, ctx->sA);
        }
        if (!rc && hipEventRecord(ctx->evA[b], ctx->sA) != hipSuccess) rc = TQ_ERR_HIP;
        // stream B: singular values of slab b
        if (!rc && hipStreamWaitEvent(ctx->sB, ctx->evA[b], 0) != hipSuccess) rc = TQ_ERR_HIP;
        if (!rc) {
            ctx->wpc_override = ctx->ov_svd_waves;
            const OutPtrs o = offset_out(out, q0);
            rc = ctx->svd_method == 0 ? launch_svd<false>(ctx, dq + q0 * 4, n, o, ctx->sB)
                                      : launch_hqr<false>(ctx, dq + q0 * 4, n, o, ctx->sB);
        }
        if (!rc && hipEventRecord(ctx->evB[b], ctx->sB) != hipSuccess) rc = TQ_ERR_HIP;
        if (rc) break;
    }
    ctx->wpc_override = 0;
    ctx->d_cm = cm0;
    if (rc) return rc > 0 ? TQ_ERR_HIP : rc;
    TQ_HIP(ctx, hipEventRecord(ctx->evEndA, ctx->sA));
    TQ_HIP(ctx, hipEventRecord(ctx->evEndB, ctx->sB));
    TQ_HIP(ctx, hipStreamWaitEvent(stream, ctx->evEndA, 0));
    TQ_HIP(ctx, hipStreamWaitEvent(stream, ctx->evEndB, 0));
    if (ctx->timing) TQ_HIP(ctx, hipEventRecord(ev.e2, stream));
    return TQ_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int tq_create(tq_ctx **out, int device_id)
{
    if (!out) return fail(nullptr, TQ_ERR_INVALID_ARG, "tq_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, TQ_ERR_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, TQ_ERR_INVALID_ARG, "device_id %d out of range (0..%d)", device_id, n - 1);
    tq_ctx *ctx = new (std::nothrow) tq_ctx();
    if (!ctx) return fail(nullptr, TQ_ERR_OOM, "out of host memory");
    ctx->device = device_id;
    e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device_id);
    if (e != hipSuccess) {
        int rc = fail(nullptr, TQ_ERR_HIP, "device %d not usable: %s", device_id, hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    *out = ctx;
    return TQ_OK;
}

void tq_destroy(tq_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    free_data(ctx);
    free_source(ctx);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_cm) (void)hipFree(ctx->d_cm);
    if (ctx->d_sort) (void)hipFree(ctx->d_sort);
    if (ctx->d_sort_tmp) (void)hipFree(ctx->d_sort_tmp);
    if (ctx->d_de) (void)hipFree(ctx->d_de);
    if (ctx->d_sv) (void)hipFree(ctx->d_sv);
    if (ctx->d_nsnps) (void)hipFree(ctx->d_nsnps);
    if (ctx->sA) {
        (void)hipStreamDestroy(ctx->sA);
        (void)hipStreamDestroy(ctx->sB);
        (void)hipEventDestroy(ctx->evIn);
        (void)hipEventDestroy(ctx->evEndA);
        (void)hipEventDestroy(ctx->evEndB);
        for (int i = 0; i < 2; ++i) {
            (void)hipEventDestroy(ctx->evA[i]);
            (void)hipEventDestroy(ctx->evB[i]);
        }
    }
    for (auto &e : ctx->events) {
        (void)hipEventDestroy(e.e0);
        (void)hipEventDestroy(e.e1);
        (void)hipEventDestroy(e.e2);
    }
    delete ctx;
}

const char *tq_last_error(const tq_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int tq_set_data(tq_ctx *ctx, const uint8_t *tmparr, int64_t T, int64_t S, const uint32_t *locus,
                int64_t locus_stride)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!tmparr || !locus) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_data: NULL pointer");
    if (T < 1 || S < 1 || locus_stride < 1 || T > 0x7FFFFFFF)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_data: bad shape T=%lld S=%lld stride=%lld", (long long)T,
                    (long long)S, (long long)locus_stride);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    free_data(ctx);

    // contiguous copy of the locus column + the run-contiguity check subsample mode relies on
    std::vector<uint32_t> loc((size_t)S);
    bool ok = true, sorted = true;
    for (int64_t i = 0; i < S; ++i) {
        loc[(size_t)i] = locus[i * locus_stride];
        if (loc[(size_t)i] == 0xFFFFFFFFu) ok = false;
        if (i && loc[(size_t)i] < loc[(size_t)i - 1]) sorted = false;
    }
    if (ok && !sorted) {
        std::unordered_set<uint32_t> seen;
        seen.insert(loc[0]);
        for (int64_t i = 1; i < S && ok; ++i)
            if (loc[(size_t)i] != loc[(size_t)i - 1] && !seen.insert(loc[(size_t)i]).second) ok = false;
    }
    ctx->locus_runs_ok = ok;

    const int64_t Sp = (int64_t)align_up((size_t)S, TILE);
    const int64_t W = Sp / 32;
    ctx->T = T; ctx->S = S; ctx->Sp = Sp; ctx->W = W;
    ctx->data_capacity = Sp;
    uint8_t *d_raw = nullptr;
    uint32_t *d_loc = nullptr;
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)(T * Sp)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_nib, (size_t)(T * Sp)));             // nib, then nib4
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)(T * W) * sizeof(uint4)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes3, (size_t)(T * W * 3 + W) * sizeof(uint32_t)));
    ctx->plane_cap_W = W;
    TQ_HIP(ctx, hipMalloc((void **)&d_raw, (size_t)(T * S)));
    hipError_t e = hipMalloc((void **)&d_loc, (size_t)S * sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipFree(d_raw);
        return fail(ctx, TQ_ERR_OOM, "hipMalloc(locus) failed: %s", hipGetErrorString(e));
    }
    e = hipMemcpy(d_raw, tmparr, (size_t)(T * S), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_loc, loc.data(), (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int64_t n = T * W;
        hipLaunchKernelGGL(tq_prepare_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_raw, d_loc, S, Sp,
                           W, (int32_t)T, ctx->d_rows, ctx->d_nib, ctx->d_nib + (size_t)(T * Sp / 2), ctx->d_planes,
                           ctx->d_planes3,
                           ctx->d_planes3 + (size_t)T * (size_t)W * 3);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    (void)hipFree(d_raw);
    (void)hipFree(d_loc);
    if (e != hipSuccess) {
        free_data(ctx);
        return fail(ctx, TQ_ERR_HIP, "tq_set_data: %s", hipGetErrorString(e));
    }
    ctx->have_data = true;
    return TQ_OK;
}

int tq_resolve_dev(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, int subsample, uint32_t *d_rstat,
                   double *d_rscor, uint8_t *d_flags, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (Q < 0 || (Q > 0 && (!d_quartets || !d_rstat || !d_rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve_dev: NULL pointer or negative Q");
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    OutPtrs out{d_rstat, d_rscor, d_flags, nullptr, nullptr, nullptr};
    return launch(ctx, d_quartets, Q, subsample, false, out, (hipStream_t)stream);
}

int tq_unrank_dev(tq_ctx *ctx, const uint64_t *d_ranks, int64_t Q, uint32_t *d_quartets, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (Q < 0 || (Q > 0 && (!d_ranks || !d_quartets)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_unrank_dev: NULL pointer or negative Q");
    if (Q == 0) return TQ_OK;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(tq_unrank_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_ranks, (uint64_t)0, Q, (int32_t)ctx->T, d_quartets);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

int tq_resolve_range_dev(tq_ctx *ctx, uint64_t first_rank, int64_t Q, int subsample, uint32_t *d_quartets,
                         uint32_t *d_rstat, double *d_rscor, uint8_t *d_flags, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (Q < 0 || (Q > 0 && (!d_rstat || !d_rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve_range_dev: NULL pointer or negative Q");
    if (Q == 0) return TQ_OK;
    const uint64_t T = (uint64_t)ctx->T;
    const uint64_t total = T < 4 ? 0 : T * (T - 1) / 2 * (T - 2) / 3 * (T - 3) / 4;
    if (first_rank + (uint64_t)Q > total)
        return fail(ctx, TQ_ERR_INVALID_ARG, "rank range [%llu,+%lld) exceeds C(%lld,4)=%llu",
                    (unsigned long long)first_rank, (long long)Q, (long long)ctx->T, (unsigned long long)total);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    uint32_t *dq = d_quartets;
    if (!dq) {
        int rc = ensure_scratch(ctx, (size_t)Q * 16);
        if (rc) return rc;
        dq = (uint32_t *)ctx->d_scratch;
    }
    hipLaunchKernelGGL(tq_unrank_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint64_t *)nullptr, first_rank, Q, (int32_t)ctx->T, dq);
    TQ_HIP(ctx, hipGetLastError());
    OutPtrs out{d_rstat, d_rscor, d_flags, nullptr, nullptr, nullptr};
    ctx->input_sorted = true;            // consecutive lexicographic ranks are in (a,b,c) order already
    const int rc = launch(ctx, dq, Q, subsample, false, out, (hipStream_t)stream);
    ctx->input_sorted = false;
    return rc;
}

int tq_resolve_debug(tq_ctx *ctx, const uint32_t *quartets, int64_t Q, int subsample, uint32_t *rstat,
                     double *rscor, uint8_t *flags, uint32_t *cmats, double *svds, int32_t *ranks)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (Q < 0 || (Q > 0 && (!quartets || !rstat || !rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve: NULL pointer or negative Q");
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (Q == 0) return TQ_OK;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    // taxon indices are checked on the host here; the kernel re-checks and flags them
    // ... and the same pass notices input that already is in (a,b,c) order -- the lexicographic
    // chunks of the reference's default mode (combinations.py:40-55) are -- so that the device sort
    // (about twenty small kernels, most of the time of a chunk of a few thousand quartets) is skipped
    bool sorted = true;
    uint64_t prev_key = 0;
    for (int64_t i = 0; i < Q; ++i) {
        const uint32_t *q = quartets + i * 4;
        for (int k = 0; k < 4; ++k)
            if (q[k] >= (uint64_t)ctx->T)
                return fail(ctx, TQ_ERR_INVALID_ARG, "quartet %lld has taxon index %u >= T=%lld", (long long)i, q[k],
                            (long long)ctx->T);
        const uint64_t key = ((uint64_t)q[0] << 42) | ((uint64_t)q[1] << 21) | (uint64_t)q[2];
        sorted &= key >= prev_key;
        prev_key = key;
    }
    ctx->input_sorted = sorted && ctx->T < (1 << 21);
    const bool debug = cmats || svds || ranks;
    const size_t o_q = 0;
    const size_t o_rstat = align_up(o_q + (size_t)Q * 16, 256);
    const size_t o_rscor = align_up(o_rstat + (size_t)Q * 8, 256);
    const size_t o_flags = align_up(o_rscor + (size_t)Q * 24, 256);
    const size_t o_cm = align_up(o_flags + (size_t)Q, 256);
    const size_t o_sv = align_up(o_cm + (cmats ? (size_t)Q * 3072 : 0), 256);
    const size_t o_rk = align_up(o_sv + (svds ? (size_t)Q * 384 : 0), 256);
    const size_t total = align_up(o_rk + (ranks ? (size_t)Q * 12 : 0), 256);
    int rc = ensure_scratch(ctx, total);
    if (rc) return rc;
    char *base = (char *)ctx->d_scratch;
    // Small calls (the reference's distributor hands out chunks of a few thousand quartets,
    // run_inference.py:73-96) are dominated by the number of blocking HIP calls: they go through one
    // pinned staging buffer -- one H2D, one D2H for the three result arrays -- instead of four pageable
    // copies.  Large calls keep the direct copies (an extra host pass over 49 B/quartet costs more).
    const bool staged = !debug && o_cm <= STAGE_LIMIT;
    if (staged) {
        rc = ensure_stage(ctx, o_cm);
        if (rc) return rc;
        memcpy(ctx->h_stage, quartets, (size_t)Q * 16);
        TQ_HIP(ctx, hipMemcpyAsync(base + o_q, ctx->h_stage, (size_t)Q * 16, hipMemcpyHostToDevice, nullptr));
    } else {
        TQ_HIP(ctx, hipMemcpy(base + o_q, quartets, (size_t)Q * 16, hipMemcpyHostToDevice));
    }
    OutPtrs out;
    out.rstat = (uint32_t *)(base + o_rstat);
    out.rscor = (double *)(base + o_rscor);
    out.flags = (uint8_t *)(base + o_flags);
    out.cmats = cmats ? (uint32_t *)(base + o_cm) : nullptr;
    out.svds = svds ? (double *)(base + o_sv) : nullptr;
    out.ranks = ranks ? (int32_t *)(base + o_rk) : nullptr;
    rc = launch(ctx, (const uint32_t *)(base + o_q), Q, subsample, debug, out, nullptr);
    ctx->input_sorted = false;
    if (rc) return rc;
    if (staged) {
        TQ_HIP(ctx, hipMemcpyAsync((char *)ctx->h_stage + o_rstat, base + o_rstat, o_cm - o_rstat, hipMemcpyDeviceToHost,
                                   nullptr));
        TQ_HIP(ctx, hipStreamSynchronize(nullptr));
        memcpy(rstat, (char *)ctx->h_stage + o_rstat, (size_t)Q * 8);
        memcpy(rscor, (char *)ctx->h_stage + o_rscor, (size_t)Q * 24);
        if (flags) memcpy(flags, (char *)ctx->h_stage + o_flags, (size_t)Q);
        return TQ_OK;
    }
    TQ_HIP(ctx, hipDeviceSynchronize());
    TQ_HIP(ctx, hipMemcpy(rstat, out.rstat, (size_t)Q * 8, hipMemcpyDeviceToHost));
    TQ_HIP(ctx, hipMemcpy(rscor, out.rscor, (size_t)Q * 24, hipMemcpyDeviceToHost));
    if (flags) TQ_HIP(ctx, hipMemcpy(flags, out.flags, (size_t)Q, hipMemcpyDeviceToHost));
    if (cmats) TQ_HIP(ctx, hipMemcpy(cmats, out.cmats, (size_t)Q * 3072, hipMemcpyDeviceToHost));
    if (svds) TQ_HIP(ctx, hipMemcpy(svds, out.svds, (size_t)Q * 384, hipMemcpyDeviceToHost));
    if (ranks) TQ_HIP(ctx, hipMemcpy(ranks, out.ranks, (size_t)Q * 12, hipMemcpyDeviceToHost));
    return TQ_OK;
}

int tq_resolve(tq_ctx *ctx, const uint32_t *quartets, int64_t Q, int subsample, uint32_t *rstat, double *rscor,
               uint8_t *flags)
{
    return tq_resolve_debug(ctx, quartets, Q, subsample, rstat, rscor, flags, nullptr, nullptr, nullptr);
}

int tq_timing_enable(tq_ctx *ctx, int on)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    ctx->timing = on != 0;
    return TQ_OK;
}

int tq_timing_read(tq_ctx *ctx, double *kernel_ms, int64_t *launches)
{
    return tq_timing_read_split(ctx, kernel_ms, nullptr, nullptr, launches);
}

int tq_timing_read_split(tq_ctx *ctx, double *total_ms, double *scan_ms, double *svd_ms, int64_t *calls)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    double t_scan = 0.0, t_svd = 0.0;
    for (size_t i = 0; i < ctx->events_used; ++i) {
        TQ_HIP(ctx, hipEventSynchronize(ctx->events[i].e2));
        float a = 0.f, b = 0.f;
        TQ_HIP(ctx, hipEventElapsedTime(&a, ctx->events[i].e0, ctx->events[i].e1));
        TQ_HIP(ctx, hipEventElapsedTime(&b, ctx->events[i].e1, ctx->events[i].e2));
        t_scan += a;
        t_svd += b;
    }
    if (total_ms) *total_ms = t_scan + t_svd;
    if (scan_ms) *scan_ms = t_scan;
    if (svd_ms) *svd_ms = t_svd;
    if (calls) *calls = ctx->timed_calls;
    ctx->events_used = 0;
    ctx->timed_calls = 0;
    return TQ_OK;
}

int tq_set_option(tq_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return TQ_ERR_INVALID_ARG;
    if (!strcmp(name, "nrep")) {
        if (value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || value == 32) ctx->nrep = (int)value;
        else if (value != 0) return fail(ctx, TQ_ERR_INVALID_ARG, "nrep must be 1,2,4,8,16 or 32");
        else ctx->nrep = 1;
        return TQ_OK;
    }
    if (!strcmp(name, "waves_per_cu")) {
        if (value < 0 || value > (1 << 20)) return fail(ctx, TQ_ERR_INVALID_ARG, "waves_per_cu must be 0..2^20");
        ctx->waves_per_cu = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "overlap")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "overlap must be >= 0");
        ctx->overlap = value;
        return TQ_OK;
    }
    if (!strcmp(name, "xcd_remap")) {
        ctx->xcd_remap = value != 0;
        return TQ_OK;
    }
    if (!strcmp(name, "svd_wpc")) {
        if (value < 0 || value > (1 << 20)) return fail(ctx, TQ_ERR_INVALID_ARG, "svd_wpc must be 0..2^20");
        ctx->svd_wpc = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "ov_scan_wgs")) {
        if (value < 0 || value > 8) return fail(ctx, TQ_ERR_INVALID_ARG, "ov_scan_wgs must be 0..8 (0: one block per workgroup)");
        ctx->ov_scan_wgs = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "ov_svd_waves")) {
        if (value < 1 || value > 32) return fail(ctx, TQ_ERR_INVALID_ARG, "ov_svd_waves must be 1..32");
        ctx->ov_svd_waves = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_wg")) {
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8 && value != 16)
            return fail(ctx, TQ_ERR_INVALID_ARG, "scan_wg must be 0 (default), 1, 2, 4, 8 or 16");
        ctx->scan_wg = value ? (int)value : 4;
        return TQ_OK;
    }
    if (!strcmp(name, "svd_method")) {
        if (value != 0 && value != 1) return fail(ctx, TQ_ERR_INVALID_ARG, "svd_method must be 0 (Jacobi) or 1 (HQR)");
        ctx->svd_method = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "order")) {
        if (value != 0 && value != 1) return fail(ctx, TQ_ERR_INVALID_ARG, "order must be 0 or 1");
        ctx->order = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_method")) {
        if (value != 0 && value != 1 && value != -1 && value != 2 && value != 3)
            return fail(ctx, TQ_ERR_INVALID_ARG, "scan_method must be -1 (auto), 0, 1 (or 2: timing diagnostic)");
        ctx->scan_method = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "batch")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "batch must be >= 0");
        ctx->batch = value ? value : (1 << 23);
        return TQ_OK;
    }
    if (!strcmp(name, "phases")) {
        if (value != 0 && value != 1 && value != 2 && value != 3)
            return fail(ctx, TQ_ERR_INVALID_ARG, "phases must be 1, 2 or 3");
        ctx->phases = value ? (int)value : 3;
        return TQ_OK;
    }
    return fail(ctx, TQ_ERR_INVALID_ARG, "unknown option '%s'", name);
}

int tq_set_source(tq_ctx *ctx, const uint8_t *seqarr, int64_t T, int64_t S0, const int64_t *spans, int64_t nloci)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!seqarr || !spans) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_source: NULL pointer");
    if (T < 1 || S0 < 1 || nloci < 1 || T > 0x7FFFFFFF)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_source: bad shape T=%lld S0=%lld nloci=%lld", (long long)T,
                    (long long)S0, (long long)nloci);
    int64_t maxw = 0;
    for (int64_t i = 0; i < nloci; ++i) {
        const int64_t a = spans[2 * i], b = spans[2 * i + 1];
        if (a < 0 || b <= a || b > S0)
            return fail(ctx, TQ_ERR_INVALID_ARG, "tq_set_source: span %lld = [%lld,%lld) outside [0,%lld)",
                        (long long)i, (long long)a, (long long)b, (long long)S0);
        if (b - a > maxw) maxw = b - a;
    }
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    free_source(ctx);
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_seqarr, (size_t)(T * S0)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_spans, (size_t)nloci * 16));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_lidxs, (size_t)nloci * 8));
    TQ_HIP(ctx, hipMemcpy(ctx->d_seqarr, seqarr, (size_t)(T * S0), hipMemcpyHostToDevice));
    TQ_HIP(ctx, hipMemcpy(ctx->d_spans, spans, (size_t)nloci * 16, hipMemcpyHostToDevice));
    ctx->src_T = T;
    ctx->src_S0 = S0;
    ctx->nloci = nloci;
    ctx->max_width = maxw;
    return TQ_OK;
}

int tq_bootstrap(tq_ctx *ctx, const int64_t *lidxs, int64_t n, uint64_t seed_shuffle, uint64_t seed_ambig,
                 int64_t *out_S)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->d_seqarr) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_source has not been called");
    if (!lidxs || n != ctx->nloci)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_bootstrap: lidxs must hold nloci=%lld locus indices", (long long)ctx->nloci);
    int64_t S = 0;
    for (int64_t i = 0; i < n; ++i)
        if (lidxs[i] < 0 || lidxs[i] >= ctx->nloci)
            return fail(ctx, TQ_ERR_INVALID_ARG, "tq_bootstrap: locus index %lld out of range", (long long)lidxs[i]);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->src_T;
    // worst-case replicate length; buffers grow only
    const int64_t cap = n * ctx->max_width;
    if (cap > ctx->boot_cap) {
        if (ctx->d_boot) (void)hipFree(ctx->d_boot);
        if (ctx->d_boot_tmp) (void)hipFree(ctx->d_boot_tmp);
        ctx->d_boot = nullptr;
        ctx->d_boot_tmp = nullptr;
        ctx->boot_cap = 0;
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_boot, (size_t)(2 * (n + 1) + 2 * cap) * sizeof(uint32_t)));
        size_t tmp = 0;
        uint32_t *u = ctx->d_boot;
        TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, u, u, (int)(n + 1)));
        size_t tmp2 = 0;
        TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp2, u, u, (int)(cap / 32 + 2)));
        if (tmp2 > tmp) tmp = tmp2;
        TQ_HIP(ctx, hipMalloc(&ctx->d_boot_tmp, tmp ? tmp : 16));
        ctx->boot_tmp_bytes = tmp;
        ctx->boot_cap = cap;
    }
    uint32_t *widths = ctx->d_boot, *offsets = widths + (n + 1);
    uint32_t *src_col = offsets + (n + 1), *site_locus = src_col + ctx->boot_cap;
    TQ_HIP(ctx, hipMemcpy(ctx->d_lidxs, lidxs, (size_t)n * 8, hipMemcpyHostToDevice));
    TQ_HIP(ctx, hipMemset(widths + n, 0, sizeof(uint32_t)));
    hipLaunchKernelGGL(tq_boot_width_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ctx->d_spans,
                       ctx->d_lidxs, n, ctx->nloci, widths);
    size_t tmp = ctx->boot_tmp_bytes;
    TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(ctx->d_boot_tmp, tmp, widths, offsets, (int)(n + 1)));
    uint32_t total = 0;
    TQ_HIP(ctx, hipMemcpy(&total, offsets + n, sizeof(uint32_t), hipMemcpyDeviceToHost));
    S = (int64_t)total;
    if (S < 1 || S > ctx->boot_cap) return fail(ctx, TQ_ERR_HIP, "tq_bootstrap: inconsistent replicate length %lld", (long long)S);
    const int64_t Sp = (int64_t)align_up((size_t)S, TILE);
    const int64_t W = Sp / 32;
    if (Sp > ctx->data_capacity || T != ctx->T) {
        free_data(ctx);
        const int64_t capSp = (int64_t)align_up((size_t)(Sp + Sp / 8), TILE);   // head-room: replicate lengths vary
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)(T * capSp)));
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_nib, (size_t)(T * capSp)));      // nib, then nib4
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)(T * (capSp / 32)) * sizeof(uint4)));
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes3, (size_t)(T * (capSp / 32) * 3 + capSp / 32) * sizeof(uint32_t)));
        ctx->plane_cap_W = capSp / 32;
        ctx->data_capacity = capSp;
    }
    hipLaunchKernelGGL(tq_boot_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ctx->d_spans,
                       ctx->d_lidxs, offsets, n, ctx->nloci, seed_shuffle, src_col, site_locus);
    const int64_t nw = T * W;
    hipLaunchKernelGGL(tq_boot_build_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, 0, ctx->d_seqarr,
                       ctx->src_S0, src_col, site_locus, S, Sp, W, (int32_t)T, seed_ambig, ctx->d_rows, ctx->d_nib,
                       ctx->d_nib + (size_t)T * (size_t)ctx->data_capacity / 2,
                       ctx->d_planes, ctx->d_planes3, ctx->d_planes3 + (size_t)T * (size_t)ctx->plane_cap_W * 3);
    TQ_HIP(ctx, hipGetLastError());
    TQ_HIP(ctx, hipDeviceSynchronize());
    ctx->T = T;
    ctx->S = S;
    ctx->Sp = Sp;
    ctx->W = W;
    ctx->have_data = true;
    ctx->locus_runs_ok = true;          // locus ids are the ordinals 0..n-1, one run each
    if (out_S) *out_S = S;
    return TQ_OK;
}

int tq_get_data(tq_ctx *ctx, uint8_t *tmparr, uint32_t *tmpmap)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "no replicate on the device");
    if (!tmparr || !tmpmap) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_get_data: NULL pointer");
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t T = ctx->T, S = ctx->S, W = ctx->W;
    const size_t bytes = align_up((size_t)(T * S), 256) + align_up((size_t)S * 8, 256) + align_up((size_t)(W + 1) * 8, 256);
    int rc = ensure_scratch(ctx, bytes);
    if (rc) return rc;
    uint8_t *d_arr = (uint8_t *)ctx->d_scratch;
    uint32_t *d_map = (uint32_t *)((char *)ctx->d_scratch + align_up((size_t)(T * S), 256));
    uint32_t *d_cnt = (uint32_t *)((char *)d_map + align_up((size_t)S * 8, 256));
    uint32_t *d_base = d_cnt + (W + 1);
    hipLaunchKernelGGL(tq_export_kernel, dim3((unsigned)((T * S + 255) / 256)), dim3(256), 0, 0, ctx->d_rows,
                       ctx->d_planes, S, ctx->Sp, W, (int32_t)T, d_arr, d_map);
    hipLaunchKernelGGL(tq_export_runcount_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, 0, ctx->d_planes, W,
                       d_cnt);
    size_t tmp = 0;
    TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, d_cnt, d_base, (int)W));
    void *d_tmp = nullptr;
    TQ_HIP(ctx, hipMalloc(&d_tmp, tmp ? tmp : 16));
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp, d_cnt, d_base, (int)W);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(tq_export_locus_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, 0, ctx->d_planes,
                           (const uint32_t *)d_base, S, W, d_map);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(tmparr, d_arr, (size_t)(T * S), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tmpmap, d_map, (size_t)S * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(ctx, TQ_ERR_HIP, "tq_get_data: %s", hipGetErrorString(e));
    return TQ_OK;
}

int tq_data_shape(tq_ctx *ctx, int64_t *T, int64_t *S)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (T) *T = ctx->have_data ? ctx->T : 0;
    if (S) *S = ctx->have_data ? ctx->S : 0;
    return TQ_OK;
}

int tq_debug_fetch(tq_ctx *ctx, int which, void *dst, int64_t bytes)
{
    if (!ctx || !dst || bytes < 0) return TQ_ERR_INVALID_ARG;
    const void *src = which == 0 ? (const void *)ctx->d_cm : which == 1 ? (const void *)ctx->d_de
                      : which == 2 ? (const void *)ctx->d_sv : nullptr;
    if (!src) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_debug_fetch: nothing to fetch (which=%d)", which);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    TQ_HIP(ctx, hipDeviceSynchronize());
    TQ_HIP(ctx, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return TQ_OK;
}

int tq_format_tsv(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q, char *out,
                  int64_t cap, int64_t *written)
{
    if (Q < 0 || cap < 0 || !written || (Q > 0 && (!quartets || !rstat || !rscor || !out))) return TQ_ERR_INVALID_ARG;
    const int64_t n = format_tsv(quartets, rstat, rscor, Q, out, cap);
    *written = n < 0 ? -n : n;
    return n < 0 ? TQ_ERR_OOM : TQ_OK;
}

int tq_format_qmc(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q, int weights,
                  int64_t min_snps, double min_ratio, char *out, int64_t cap, int64_t *written, int64_t *n_lines)
{
    if (Q < 0 || cap < 0 || !written || weights < 0 || weights > 3 ||
        (Q > 0 && (!quartets || !rstat || !rscor || !out)))
        return TQ_ERR_INVALID_ARG;
    const int64_t n = format_qmc(quartets, rstat, rscor, Q, weights, min_snps, min_ratio, out, cap, n_lines);
    *written = n < 0 ? -n : n;
    return n < 0 ? TQ_ERR_OOM : TQ_OK;
}

int tq_device_info(tq_ctx *ctx, int32_t *num_cu, int32_t *waves_per_cu, int64_t *row_pitch)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (num_cu) *num_cu = ctx->prop.multiProcessorCount;
    if (waves_per_cu) *waves_per_cu = ctx->waves_per_cu;
    if (row_pitch) *row_pitch = ctx->Sp;
    return TQ_OK;
}

}  // extern "C"

