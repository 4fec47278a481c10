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
//   nib     u8   [T][Sp/2] the same codes, two per byte (nib_offset())
//   nib5    u8   [T][Sp/2] the same with 4 for a missing cell (rows d1, d2 of the joint-histogram scan)
//   planes  u32x4[T][W]    per 32 sites: {missing bits, base bit 0, base bit 1, run-begin bits}, W = Sp/32
//   planes3 u32x3[T][W]    compact copy {missing, bit 0, bit 1}; runbeg u32 [W] run-begin bits, stored once
//
// Kernels (each in its own header of this directory, all included below into one translation unit):
//   prepare.hpp   layout build, lexicographic unranking, sort keys
//   scan.hpp      tq_scan_wg_kernel / tq_scan_kernel: site scan -> 256 pattern counts per quartet (nibble codes + plane
//                 records; the one-wave-per-quartet kernel of small calls)
//   scan_f4.hpp   tq_scan_f4_kernel: the cooperative scan on 12-byte plane records only (default of subsample mode)
//   scan_dp.hpp   tq_scan_dp_kernel: two quartets that share three taxa per wave, one joint histogram (default of full mode
//                 from 32 768 quartets on); unit list from the sorted order
//   scan_pb.hpp   tq_scan_pb_kernel: bank-private counters (A/B form)
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
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/tetrad_hip.h"

namespace {

#include "common.hpp"
#include "prepare.hpp"
#include "scan.hpp"
#include "scan_pb.hpp"
#include "scan_dp.hpp"
#include "scan_f4.hpp"
#include "jacobi.hpp"
#include "hqr.hpp"
#include "bootstrap.hpp"
#include "format.hpp"
#include "qmc.hpp"

}  // namespace

// ======================================================================================
// host side: context + C ABI
// ======================================================================================
enum { TAG_ORIGIN = -1, TAG_ORDER = 0, TAG_SCAN = 1, TAG_BIDIAG = 2, TAG_BDSQR = 3, TAG_SCORE = 4, TAG_COUNT = 5 };

struct tq_ctx {
    int device = 0;
    std::string err;
    hipDeviceProp_t prop{};
    // replicate data
    int64_t T = 0, S = 0, Sp = 0, W = 0;
    uint8_t *d_rows = nullptr;
    uint8_t *d_nib = nullptr;       // [T][Sp/2] nibble-packed copy of the rows (common.hpp: nib_offset)
    uint8_t *d_nib5 = nullptr;      // [T][Sp/2] the same with 4 = missing (scan_dp.hpp)
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
    std::vector<int64_t> h_spans;   // host copy of the spans (replicate lengths are computed on the host)
    int64_t *h_lidx_stage[2] = {nullptr, nullptr};   // page-locked staging of the resampled locus indices
    hipEvent_t ev_lidx[2] = {nullptr, nullptr};      // its H2D has been consumed
    unsigned lidx_turn = 0;
    uint32_t *d_boot = nullptr;     // widths/offsets [nloci+1] | src_col [cap] | site_locus [cap]
    int64_t boot_cap = 0;
    void *d_boot_tmp = nullptr;
    size_t boot_tmp_bytes = 0;
    // scratch for the host-buffer API
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // count slab between the scan and the singular-value stage: u32 [batch][256]; ordering scratch
    uint32_t *d_cm = nullptr;
    int64_t cm_quartets = 0;
    uint32_t *d_sort = nullptr;     // 4 arrays of cm_quartets u32: keys/idx in, keys/idx out
    void *d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    int order = 1;                  // 1 = process quartets in (a,b,c)-sorted order
    uint2 *d_units = nullptr;       // cm_quartets + 2 entries: unit list of the joint-histogram scan (scan_dp.hpp), then its count
    int scan_f4 = -1;               // the cooperative scan on 12-byte plane records only (SURVEY 8 row f4 as written; scan_f4.hpp):
                                    // -1 = in subsample mode (c3 scan 5.81 -> 5.58 ms), 1 = in both modes, 0 = never
    int scan_dp = 1;                // 1 = full-mode batches of >= dp_min_quartets go to tq_scan_dp_kernel (two quartets that share
                                    // (a,b,c) per wave, one LDS atomic per site and pair)
    int64_t dp_min_quartets = 32768;
    // singular-value stage scratch, sized for one chunk of `svd_chunk` quartets and re-used chunk after
    // chunk (so the bidiagonals / values of a chunk stay in the Infinity Cache between its three kernels):
    // de f64[3*chunk][32], sv f64[3*chunk][16], nsnps u32[chunk]
    double *d_de = nullptr, *d_sv = nullptr;
    uint32_t *d_nsnps = nullptr;
    int64_t svd_quartets = 0;
    int64_t svd_chunk = 1 << 18;    // quartets per pass of the singular-value stage (and per result D2H piece)
    int svd_streams = 2;            // chunks alternate between this many streams (1 or 2) so that the tail of one
                                    // chunk's kernels is filled by the next chunk's (each stream has its own scratch)
    hipStream_t sX = nullptr;       // the second stream of the singular-value stage
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    int svd_method = 1;             // 0 = one-sided Jacobi (tq_svd_kernel), 1 = Householder + bidiagonal QR
    int bidiag_layout = 1;          // 1 = matrix dealt 2 x 2 over the quad (tq_bidiag2_kernel), 0 = four column groups
    int bdsqr_maxit = 60;           // QR sweeps per singular value before a matrix is declared not converged
    uint64_t *d_bdsqr_stats = nullptr;   // diagnostics (option "bdsqr_stats"): {matrices, rotation steps of all lanes,
                                         // lane-slots issued (64 x wave iterations), sweeps} summed over the launches
    int scan_wg = 4;                // waves per workgroup of the cooperative scan kernel (1 = one wave per quartet)
    int64_t wg_min_quartets = 4096; // smaller batches go to the one-wave-per-quartet kernel: a call of a few thousand quartets does not
                                    // fill the chip and is bound by the latency of one quartet's 25 dependent steps, which the
                                    // cooperative kernels' per-step barrier and image hand-over only lengthen (1 000 random
                                    // quartets 0.185 -> 0.164 ms per call, 3 000: 0.198 -> 0.191; the cooperative kernels win from
                                    // ~4 000 on: tools/experiments/wg_min_threshold.py)
    int xcd_remap = 1;              // 1: scan workgroups of one XCD take a contiguous part of the sorted order
    int count_invariant = 0;        // 1: invariant sites (all four bases equal, none missing) are counted as well -- what the reference's
                                    // count kernels do when their caller's mask leaves such a site open (resolve_quartets.py:59-64);
                                    // the worker itself always masks them (:218).  One-wave-per-quartet kernel only.
    int scan_pair = 0;              // 1: two quartets per wavefront (tq_scan_wg2_kernel)
    int park_t = 1;                 // 1 (default): transposed pattern park of the set-bit walk (conflict-free byte reads at the price of
                                    // 2 more VALU per counted site: c3 6.79 -> 6.42 ms); 0: lane-contiguous park (A/B)
    int share_c = 0;                // 1: scan kernel variant that also shares row c inside a workgroup (scan.hpp: SHC;
                                    // measured slower everywhere -- the kernel is LDS/VALU-bound, not byte-bound -- kept as an A/B option)
    int svd_wpc = 0;                // blocks per CU of the bidiag / bdsqr grids (0 = one pass per block)
    // what tq_scan_dev left in the count slab (consumed by tq_svd_dev)
    const uint32_t *scanned_q = nullptr;
    int64_t scanned_Q = 0;
    // host-buffer API: own compute and copy streams, events for the D2H pipeline
    hipStream_t sK = nullptr, sC = nullptr;
    std::vector<hipEvent_t> pipe_events;
    // the asynchronous device API (caller's stream) and the synchronous host API (stream sK) share the count
    // slab, the ordering scratch and the singular-value scratch: the last device-API enqueue records this event
    // and the host API makes sK wait for it before it touches any of them
    hipEvent_t evDevApi = nullptr;
    bool dev_api_pending = false;
    hipStream_t last_dev_stream = nullptr;   // stream of the last device-API enqueue (enter_dev_api orders across streams)
    // options
    int nrep = 1;
    int waves_per_cu = 0;           // 0 = from the occupancy query
    int phases = 3;                 // diagnostics only: 1 = scan kernel only, 2 = SVD kernel only
    int scan_method = -1;           // 0 = EXEC-masked slot per site, 1 = set-bit walk, 6 = bank-private counters (scan_pb.hpp,
                                    // A/B form; 0 / 1 where it does not apply); -1 = 1 if subsample else 0
    int pb_ok = 0;                  // tq_scan_pb_kernel's counters sit on a 64 KiB LDS boundary (probed once in tq_create)
    int64_t batch = 1 << 23;        // quartets per scan batch (8 GiB count slab)
    // timing: a sequence of tagged HIP events on the launch stream; the time between two consecutive
    // marks is attributed to the tag of the later one (TAG_ORIGIN starts a sequence)
    bool timing = false;
    struct Mark { int tag; int lane; hipEvent_t ev; };      // lane: 0 = caller's stream, 1 = sX
    std::vector<Mark> marks;
    std::vector<hipEvent_t> event_pool;
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
    try {
        if (ctx) ctx->err = buf; else g_create_err = buf;
    } catch (...) {
    }
    return code;
}

#define TQ_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, e_ == hipErrorOutOfMemory ? TQ_ERR_OOM : TQ_ERR_HIP, "%s failed: %s", \
                        #call, hipGetErrorString(e_));                                        \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Pinned host memory pool (process-wide: blocks handed to a caller may outlive any context).
// Result arrays that live in such a block are written by the copy engine directly (no staging, no
// host pass); blocks are recycled because pinning pages costs about as much as a resolve call.
// ---------------------------------------------------------------------------------------------
struct PinnedPool {
    std::mutex mu;
    std::map<uintptr_t, size_t> live;              // base -> bytes of blocks handed out
    std::multimap<size_t, void *> idle;            // bytes -> base of cached blocks
    size_t idle_bytes = 0;
    static constexpr size_t IDLE_CAP = (size_t)4 << 30;
    static size_t round_up(size_t b)
    {
        size_t g = (size_t)1 << 16;
        while (g < b && g < ((size_t)1 << 24)) g <<= 1;         // 64 KiB .. 16 MiB: powers of two
        if (g >= b) return g;
        const size_t step = (size_t)1 << 24;                   // beyond: multiples of 16 MiB
        return (b + step - 1) / step * step;
    }
    int alloc(size_t bytes, void **out)
    {
        const size_t want = round_up(bytes ? bytes : 1);
        std::lock_guard<std::mutex> g(mu);
        try {
            auto it = idle.lower_bound(want);
            if (it != idle.end() && it->first <= want * 2) {
                void *p = it->second;
                const size_t sz = it->first;
                idle.erase(it);
                idle_bytes -= sz;
                live[(uintptr_t)p] = sz;
                *out = p;
                return TQ_OK;
            }
            void *p = nullptr;
            if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess || !p) return TQ_ERR_OOM;
            live[(uintptr_t)p] = want;
            *out = p;
            return TQ_OK;
        } catch (...) {
            return TQ_ERR_OOM;
        }
    }
    int release(void *p)
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = live.find((uintptr_t)p);
        if (it == live.end()) return TQ_ERR_INVALID_ARG;
        const size_t sz = it->second;
        live.erase(it);
        bool cached = false;
        if (idle_bytes + sz <= IDLE_CAP) {
            try {
                idle.emplace(sz, p);
                idle_bytes += sz;
                cached = true;
            } catch (...) {
            }
        }
        if (!cached) (void)hipHostFree(p);
        return TQ_OK;
    }
    bool owns(const void *p, size_t bytes)
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = live.upper_bound((uintptr_t)p);
        if (it == live.begin()) return false;
        --it;
        return (uintptr_t)p + bytes <= it->first + it->second;
    }
};

PinnedPool &pool()
{
    static PinnedPool *p = new PinnedPool();       // never destroyed: blocks may be alive at exit
    return *p;
}

// true when [p, p+bytes) is page-locked memory the copy engine can write asynchronously
bool is_pinned(const void *p, size_t bytes)
{
    if (!p) return false;
    if (pool().owns(p, bytes)) return true;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                               // unregistered memory: clear the sticky error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

void free_data(tq_ctx *ctx)
{
    if (ctx->d_rows) (void)hipFree(ctx->d_rows);
    if (ctx->d_nib) (void)hipFree(ctx->d_nib);
    if (ctx->d_nib5) (void)hipFree(ctx->d_nib5);
    if (ctx->d_planes) (void)hipFree(ctx->d_planes);
    if (ctx->d_planes3) (void)hipFree(ctx->d_planes3);
    ctx->d_rows = nullptr;
    ctx->d_nib = nullptr;
    ctx->d_nib5 = nullptr;
    ctx->d_planes = nullptr;
    ctx->d_planes3 = nullptr;
    ctx->have_data = false;
    ctx->data_capacity = 0;
    ctx->scanned_Q = 0;
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
    for (int i = 0; i < 2; ++i) {
        if (ctx->h_lidx_stage[i]) (void)pool().release(ctx->h_lidx_stage[i]);
        if (ctx->ev_lidx[i]) (void)hipEventDestroy(ctx->ev_lidx[i]);
        ctx->h_lidx_stage[i] = nullptr;
        ctx->ev_lidx[i] = nullptr;
    }
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

// count slab + ordering scratch for a scan batch of `quartets`
int ensure_cm(tq_ctx *ctx, int64_t quartets)
{
    if (quartets <= ctx->cm_quartets) return TQ_OK;
    if (ctx->d_cm) (void)hipFree(ctx->d_cm);
    if (ctx->d_sort) (void)hipFree(ctx->d_sort);
    if (ctx->d_sort_tmp) (void)hipFree(ctx->d_sort_tmp);
    if (ctx->d_units) (void)hipFree(ctx->d_units);
    ctx->d_cm = nullptr;
    ctx->d_sort = nullptr;
    ctx->d_sort_tmp = nullptr;
    ctx->d_units = nullptr;
    ctx->cm_quartets = 0;
    ctx->scanned_Q = 0;
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_cm, (size_t)quartets * 1024));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_sort, (size_t)quartets * 16));
    size_t tmp = 0;
    uint32_t *k = ctx->d_sort;
    TQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, k, k, k, k, (int)quartets));
    size_t tmp2 = 0;
    TQ_HIP(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tmp2, k, k, (int)quartets));
    if (tmp2 > tmp) tmp = tmp2;
    TQ_HIP(ctx, hipMalloc(&ctx->d_sort_tmp, tmp ? tmp : 16));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_units, ((size_t)quartets + 2) * sizeof(uint2)));
    ctx->sort_tmp_bytes = tmp;
    ctx->cm_quartets = quartets;
    return TQ_OK;
}

// scratch of the singular-value stage: two sets (one per stream) for chunks of `quartets`
int ensure_svd(tq_ctx *ctx, int64_t quartets)
{
    if (!ctx->sX) {
        TQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->sX, hipStreamNonBlocking));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evFork, hipEventDisableTiming));
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evJoin, hipEventDisableTiming));
    }
    if (quartets <= ctx->svd_quartets) return TQ_OK;
    if (ctx->d_de) (void)hipFree(ctx->d_de);
    if (ctx->d_sv) (void)hipFree(ctx->d_sv);
    if (ctx->d_nsnps) (void)hipFree(ctx->d_nsnps);
    ctx->d_de = ctx->d_sv = nullptr;
    ctx->d_nsnps = nullptr;
    ctx->svd_quartets = 0;
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_de, 2 * (size_t)quartets * 3 * 32 * sizeof(double)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_sv, 2 * (size_t)quartets * 3 * 16 * sizeof(double)));
    TQ_HIP(ctx, hipMalloc((void **)&ctx->d_nsnps, 2 * (size_t)quartets * sizeof(uint32_t)));
    ctx->svd_quartets = quartets;
    return TQ_OK;
}

// ---- timing marks -------------------------------------------------------------------------------
int mark(tq_ctx *ctx, int tag, hipStream_t stream, int lane = 0)
{
    if (!ctx->timing) return TQ_OK;
    hipEvent_t ev;
    if (!ctx->event_pool.empty()) {
        ev = ctx->event_pool.back();
        ctx->event_pool.pop_back();
    } else {
        TQ_HIP(ctx, hipEventCreate(&ev));
    }
    try {
        ctx->marks.push_back({tag, lane, ev});
    } catch (...) {
        (void)hipEventDestroy(ev);
        return fail(ctx, TQ_ERR_OOM, "out of host memory");
    }
    TQ_HIP(ctx, hipEventRecord(ev, stream));
    return TQ_OK;
}

// order[] for one batch: indices sorted by (first, second[, third]) taxon; nullptr = natural order
int make_order(tq_ctx *ctx, const uint32_t *dq, int64_t n, bool input_sorted, hipStream_t stream,
               const uint32_t **order)
{
    *order = nullptr;
    // below ~32k quartets the sort's twenty small launches cost more than the shared rows save
    // (tools/small_call_order.py: 8 000 quartets 0.34 ms sorted, 0.31 ms in natural order; 62 500: 1.1 vs 2.4 ms)
    if (!ctx->order || n < 32768 || ctx->T > 65535 || input_sorted) return TQ_OK;
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

// Joint-histogram scan (scan_dp.hpp) for this batch?  Full mode, automatic kernel choice, the default workgroup shape,
// keys that hold (a,b,c), and an order the pairing can rely on (sorted here, or sorted on arrival).
bool use_dp(const tq_ctx *ctx, int64_t n, int subsample, bool input_sorted)
{
    const uint64_t T = (uint64_t)ctx->T;
    return !subsample && ctx->scan_dp && ctx->scan_f4 <= 0 && ctx->scan_method < 0 && ctx->scan_wg == 4 && !ctx->count_invariant &&
           !ctx->share_c && !ctx->scan_pair && ctx->waves_per_cu == 0 && (ctx->order || input_sorted) &&
           n >= ctx->dp_min_quartets && n >= 2 && n <= 0x7FFFFFFF && T * T * T <= 0xFFFFFFFFull &&
           T * (uint64_t)ctx->Sp < 0xFFFF0000ull;
}

// ordering + unit list of the joint-histogram scan: d_units[0..count) = (first quartet, second quartet or DP_NONE) in
// (a,b,c)-sorted order, count in d_units[cm_quartets] (read by the kernel: nothing comes back to the host)
int make_units(tq_ctx *ctx, const uint32_t *dq, int64_t n, bool input_sorted, hipStream_t stream)
{
    uint32_t *keys_in = ctx->d_sort, *idx_in = keys_in + ctx->cm_quartets;
    uint32_t *keys_out = idx_in + ctx->cm_quartets, *idx_out = keys_out + ctx->cm_quartets;
    const uint64_t T = (uint64_t)ctx->T;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(tq_key_kernel, dim3(blocks), dim3(256), 0, stream, dq, n, (uint32_t)ctx->T, 1, keys_in, idx_in);
    TQ_HIP(ctx, hipGetLastError());
    const uint32_t *keys = keys_in, *idx = idx_in;
    uint32_t *flags = keys_out;
    if (!input_sorted) {
        int bits = 1;
        while (bits < 32 && (1ull << bits) < T * T * T) ++bits;
        size_t tmp = ctx->sort_tmp_bytes;
        TQ_HIP(ctx, hipcub::DeviceRadixSort::SortPairs(ctx->d_sort_tmp, tmp, keys_in, keys_out, idx_in, idx_out, (int)n,
                                                       0, bits, stream));
        keys = keys_out;
        idx = idx_out;
        flags = keys_in;
    }
    hipLaunchKernelGGL(tq_dp_flag_kernel, dim3(blocks), dim3(256), 0, stream, keys, idx, dq, n, (uint32_t)ctx->T, flags);
    TQ_HIP(ctx, hipGetLastError());
    size_t tmp = ctx->sort_tmp_bytes;
    TQ_HIP(ctx, hipcub::DeviceScan::InclusiveSum(ctx->d_sort_tmp, tmp, flags, flags, (int)n, stream));
    hipLaunchKernelGGL(tq_dp_units_kernel, dim3(blocks), dim3(256), 0, stream, keys, idx, (const uint32_t *)flags, n,
                       ctx->d_units, reinterpret_cast<uint32_t *>(ctx->d_units + ctx->cm_quartets));
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

DevData dev_data(const tq_ctx *ctx)
{
    DevData d;
    d.rows = ctx->d_rows;
    d.nib = ctx->d_nib;
    d.nib5 = ctx->d_nib5;
    d.planes = ctx->d_planes;
    d.planes3 = ctx->d_planes3;
    d.runbeg = ctx->d_planes3 + (size_t)ctx->T * (size_t)ctx->plane_cap_W * 3;
    d.pitch = ctx->Sp;
    d.W = ctx->W;
    d.T = (int32_t)ctx->T;
    d.ntiles = (int32_t)(ctx->Sp / TILE);
    d.inv = ctx->count_invariant ? 0xFFFFFFFFu : 0u;
    return d;
}

template <typename K>
int grid_for(tq_ctx *ctx, K kern, int64_t items, int64_t *grid, int wpc_kernel = 0)
{
    int wpc = wpc_kernel > 0 ? wpc_kernel : ctx->waves_per_cu;
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

template <bool SUB, int METHOD, int NW, bool SHC = false, bool PARK_T = false>
int launch_scan_wg(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, hipStream_t stream)
{
    auto kern = tq_scan_wg_kernel<SUB, METHOD, NW, SHC, PARK_T>;
    const int wgs = ctx->waves_per_cu > 0 ? (ctx->waves_per_cu + NW - 1) / NW : 0;
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

// bank-private counters (scan_pb.hpp): one block of 4 quartets per workgroup, as launch_scan_wg
template <bool SUB>
int launch_scan_pb(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, hipStream_t stream)
{
    auto kern = tq_scan_pb_kernel<SUB>;
    const int64_t nblk = (Q + PB_NW - 1) / PB_NW;
    int64_t grid = nblk, xcd_chunk = 0;
    if (ctx->xcd_remap && nblk >= 64) {                      // one block per workgroup, XCD-contiguous
        xcd_chunk = (nblk + 7) / 8;
        grid = xcd_chunk * 8;
    }
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(PB_NW * WAVE), 0, stream, dev_data(ctx), dq, order, Q, ctx->d_cm,
                       xcd_chunk, (uint32_t *)nullptr);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

// One-time check (tq_create) that the kernel's counters start on a 64 KiB LDS boundary, which its address arithmetic
// relies on: a launch with Q < 0 reports the offset and does nothing else.
int probe_scan_pb(tq_ctx *ctx)
{
    uint32_t *d = nullptr, h[2] = {0, 0};
    TQ_HIP(ctx, hipMalloc(&d, sizeof h));
    hipError_t e = hipMemset(d, 0, sizeof h);
    if (e == hipSuccess) {
        DevData none{};
        hipLaunchKernelGGL(tq_scan_pb_kernel<false>, dim3(1), dim3(PB_NW * WAVE), 0, 0, none, (const uint32_t *)nullptr,
                           (const uint32_t *)nullptr, (int64_t)-1, (uint32_t *)nullptr, (int64_t)0, d);
        hipLaunchKernelGGL(tq_scan_pb_kernel<true>, dim3(1), dim3(PB_NW * WAVE), 0, 0, none, (const uint32_t *)nullptr,
                           (const uint32_t *)nullptr, (int64_t)-1, (uint32_t *)nullptr, (int64_t)0, d + 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, TQ_ERR_HIP, "probe of tq_scan_pb_kernel failed: %s", hipGetErrorString(e));
    ctx->pb_ok = (h[0] == 0x80000000u && h[1] == 0x80000000u) ? 1 : -1;
    return TQ_OK;
}

template <bool SUB, int METHOD, int NW>
int launch_scan_wg2(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, hipStream_t stream)
{
    auto kern = tq_scan_wg2_kernel<SUB, METHOD, NW>;
    const int64_t nblk = (Q + 2 * NW - 1) / (2 * NW);
    int64_t grid = nblk, xcd_chunk = 0;
    if (ctx->xcd_remap && nblk >= 64) {                      // one block per workgroup, XCD-contiguous
        xcd_chunk = (nblk + 7) / 8;
        grid = xcd_chunk * 8;
    }
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * WAVE), 0, stream, dev_data(ctx), dq, order, Q, ctx->d_cm,
                       xcd_chunk);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

// joint-histogram scan over the unit list make_units left: at least ceil(Q / 2) units, at most Q -- the grid covers the
// least (rounded to the 8 XCDs) and a workgroup whose id + grid is still a block takes that one as well
int launch_scan_dp(tq_ctx *ctx, const uint32_t *dq, int64_t Q, hipStream_t stream)
{
    const int64_t least = ((Q + 1) / 2 + DP_NW - 1) / DP_NW;
    const int64_t grid = (least + 7) / 8 * 8;
    hipLaunchKernelGGL(tq_scan_dp_kernel<DP_NW>, dim3((unsigned)grid), dim3(DP_NW * WAVE), 0, stream, dev_data(ctx), dq,
                       (const uint2 *)ctx->d_units, reinterpret_cast<const uint32_t *>(ctx->d_units + ctx->cm_quartets),
                       ctx->d_cm);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

int launch_scan_n(tq_ctx *ctx, const uint32_t *dq, const uint32_t *order, int64_t Q, int subsample,
                  hipStream_t stream)
{
    if (ctx->scan_pair && ctx->scan_wg == 4 && Q >= 64 && !ctx->count_invariant && !ctx->share_c && ctx->waves_per_cu == 0 &&
        ctx->scan_method < 2 && (uint64_t)ctx->T * (uint64_t)ctx->Sp < 0xFFFF0000ull) {
        const int m = ctx->scan_method < 0 ? (subsample ? 1 : 0) : ctx->scan_method;
        if (subsample)
            return m ? launch_scan_wg2<true, 1, 4>(ctx, dq, order, Q, stream) : launch_scan_wg2<true, 0, 4>(ctx, dq, order, Q, stream);
        return m ? launch_scan_wg2<false, 1, 4>(ctx, dq, order, Q, stream) : launch_scan_wg2<false, 0, 4>(ctx, dq, order, Q, stream);
    }
    // row f4 as written (option scan_f4): own rows as 12-byte plane records only, pattern bits pulled out in the walk
    const bool f4 = ctx->scan_f4 < 0 ? (subsample != 0 && ctx->scan_method < 0 && ctx->park_t) : ctx->scan_f4 != 0;
    if (f4 && (ctx->scan_wg == 4 || ctx->scan_wg == 8 || ctx->scan_wg == 2) && Q >= ctx->wg_min_quartets && !ctx->count_invariant &&
        !ctx->share_c && !ctx->scan_pair && ctx->waves_per_cu == 0 && (ctx->scan_method < 0 || ctx->scan_method == 1) &&
        (uint64_t)ctx->T * (uint64_t)ctx->Sp < 0xFFFF0000ull) {
        const int nw = ctx->scan_wg;
        const int64_t nblk = (Q + nw - 1) / nw;
        int64_t grid = nblk, xcd_chunk = 0;
        if (ctx->xcd_remap && nblk >= 64) {
            xcd_chunk = (nblk + 7) / 8;
            grid = xcd_chunk * 8;
        }
#define TQ_F4_CASE(SUBF, NWF)                                                                                         \
        if ((subsample != 0) == SUBF && nw == NWF)                                                                    \
            hipLaunchKernelGGL((tq_scan_f4_kernel<SUBF, NWF>), dim3((unsigned)grid), dim3(NWF * WAVE), 0, stream,      \
                               dev_data(ctx), dq, order, Q, ctx->d_cm, xcd_chunk);
        TQ_F4_CASE(true, 4) TQ_F4_CASE(false, 4) TQ_F4_CASE(true, 8) TQ_F4_CASE(false, 8) TQ_F4_CASE(true, 2) TQ_F4_CASE(false, 2)
#undef TQ_F4_CASE
        TQ_HIP(ctx, hipGetLastError());
        return TQ_OK;
    }
    // bank-private counters (option scan_method = 6, an A/B form: conflict-free atomics, but an LDS atomic costs its 4
    // cycles of operand transfer either way and the 64 KiB of counters leave two workgroups per CU -- measured slower,
    // scan_pb.hpp); 16-bit counters, so only while a quartet has at most PB_MAX_TILES steps
    const bool pb_fits = ctx->pb_ok == 1 && ctx->scan_wg == 4 && Q >= 64 && !ctx->count_invariant && !ctx->share_c &&
                         !ctx->scan_pair && ctx->waves_per_cu == 0 && ctx->Sp / TILE <= PB_MAX_TILES &&
                         (uint64_t)ctx->T * (uint64_t)ctx->Sp < 0xFFFF0000ull;
    if (pb_fits && ctx->scan_method == 6)
        return subsample ? launch_scan_pb<true>(ctx, dq, order, Q, stream) : launch_scan_pb<false>(ctx, dq, order, Q, stream);
    if (ctx->scan_wg >= 2 && Q >= ctx->wg_min_quartets && !ctx->count_invariant &&
        (uint64_t)ctx->T * (uint64_t)ctx->Sp < 0xFFFF0000ull) {
        int m = ctx->scan_method < 0 ? (subsample ? 1 : 0) : ctx->scan_method;
        if (m == 6) m = subsample ? 1 : 0;
#define TQ_WG_CASE(NW)                                                                                   \
    if (ctx->scan_wg == NW) {                                                                            \
        if (m == 2) return launch_scan_wg<true, 2, NW>(ctx, dq, order, Q, stream);                       \
        if (m == 3) return launch_scan_wg<true, 3, NW>(ctx, dq, order, Q, stream);                       \
        if (NW == 4 && m == 4) return launch_scan_wg<true, 4, 4>(ctx, dq, order, Q, stream);             \
        if (NW == 4 && m == 5) return launch_scan_wg<true, 5, 4>(ctx, dq, order, Q, stream);             \
        if (subsample)                                                                                   \
            return m ? launch_scan_wg<true, 1, NW>(ctx, dq, order, Q, stream)                            \
                     : launch_scan_wg<true, 0, NW>(ctx, dq, order, Q, stream);                           \
        return m ? launch_scan_wg<false, 1, NW>(ctx, dq, order, Q, stream)                               \
                 : launch_scan_wg<false, 0, NW>(ctx, dq, order, Q, stream);                              \
    }
        TQ_WG_CASE(2)
        if (ctx->scan_wg == 4 && ctx->park_t && m == 1 && !ctx->share_c)
            return subsample ? launch_scan_wg<true, 1, 4, false, true>(ctx, dq, order, Q, stream)
                             : launch_scan_wg<false, 1, 4, false, true>(ctx, dq, order, Q, stream);
        if ((ctx->scan_wg == 8 || ctx->scan_wg == 6) && ctx->park_t && m == 1 && subsample && !ctx->share_c)
            return ctx->scan_wg == 8 ? launch_scan_wg<true, 1, 8, false, true>(ctx, dq, order, Q, stream)
                                     : launch_scan_wg<true, 1, 6, false, true>(ctx, dq, order, Q, stream);
        if (ctx->scan_wg == 4 && ctx->share_c && m <= 1) {
            if (subsample)
                return m ? launch_scan_wg<true, 1, 4, true>(ctx, dq, order, Q, stream)
                         : launch_scan_wg<true, 0, 4, true>(ctx, dq, order, Q, stream);
            return m ? launch_scan_wg<false, 1, 4, true>(ctx, dq, order, Q, stream)
                     : launch_scan_wg<false, 0, 4, true>(ctx, dq, order, Q, stream);
        }
        TQ_WG_CASE(4)
        TQ_WG_CASE(3)
        TQ_WG_CASE(6)
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
    const int method = (ctx->scan_method < 0 || ctx->scan_method == 6) ? (subsample ? 1 : 0) : ctx->scan_method;
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

// Jacobi path: count slab rows [cm, cm + n) -> outputs of the same rows
template <bool DEBUG>
int launch_svd(tq_ctx *ctx, const uint32_t *cm, const uint32_t *dq, int64_t n, const OutPtrs &out, hipStream_t stream,
               int lane)
{
    auto kern = tq_svd_kernel<DEBUG>;
    int64_t grid;
    int rc = grid_for(ctx, kern, (n + QPW - 1) / QPW, &grid);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WAVE), 0, stream, cm, dq, n, (int32_t)ctx->T, out);
    TQ_HIP(ctx, hipGetLastError());
    return mark(ctx, TAG_BIDIAG, stream, lane);
}

// Householder + QR path, one chunk (n <= svd scratch)
template <bool DEBUG>
int launch_hqr(tq_ctx *ctx, const uint32_t *cm, const uint32_t *dq, int64_t n, const OutPtrs &out, hipStream_t stream,
               int lane)
{
    // scratch set of this stream
    double *de = ctx->d_de + (size_t)lane * (size_t)ctx->svd_quartets * 96;
    double *sv = ctx->d_sv + (size_t)lane * (size_t)ctx->svd_quartets * 48;
    uint32_t *nsnps = ctx->d_nsnps + (size_t)lane * (size_t)ctx->svd_quartets;
    int64_t grid;
    auto k1 = ctx->bidiag_layout ? tq_bidiag2_kernel<DEBUG> : tq_bidiag_kernel<DEBUG>;
    // one pass per block unless told otherwise: the work per pass varies (QR iterations), and the
    // hardware dispatcher balances it better than a static grid-stride loop (3.8 ms vs 5.1 ms per 1e6)
    const int svd_wpc = ctx->svd_wpc > 0 ? ctx->svd_wpc : (1 << 20);
    const int tsplit = n < 32768 ? 1 : 0;          // small batches: one block per (pass, flattening), tq_bidiag_kernel
    int rc = grid_for(ctx, k1, (tsplit ? 3 : 1) * ((n + 15) / 16), &grid, svd_wpc);
    if (rc) return rc;
    hipLaunchKernelGGL(k1, dim3((unsigned)grid), dim3(WAVE), 0, stream, cm, n, de, nsnps, out.cmats, tsplit);
    TQ_HIP(ctx, hipGetLastError());
    if ((rc = mark(ctx, TAG_BIDIAG, stream, lane))) return rc;
    const int64_t nmat = 3 * n;
    rc = grid_for(ctx, tq_bdsqr_kernel, (nmat + WAVE - 1) / WAVE, &grid, svd_wpc);
    if (rc) return rc;
    hipLaunchKernelGGL(tq_bdsqr_kernel, dim3((unsigned)grid), dim3(WAVE), 0, stream, (const double *)de, nmat, sv,
                       ctx->bdsqr_maxit, (unsigned long long *)ctx->d_bdsqr_stats);
    TQ_HIP(ctx, hipGetLastError());
    if ((rc = mark(ctx, TAG_BDSQR, stream, lane))) return rc;
    hipLaunchKernelGGL(tq_score_kernel<DEBUG>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const double *)sv, (const uint32_t *)nsnps, dq, n, (int32_t)ctx->T, out);
    TQ_HIP(ctx, hipGetLastError());
    return mark(ctx, TAG_SCORE, stream, lane);
}

bool diagnostic_mode(const tq_ctx *ctx)
{
    return (ctx->scan_method >= 2 && ctx->scan_method <= 5) || ctx->phases != 3;
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

int check_ready(tq_ctx *ctx, int subsample)
{
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_data has not been called");
    if (subsample && !ctx->locus_runs_ok)
        return fail(ctx, TQ_ERR_LOCUS_ORDER,
                    "subsample mode needs each locus id in one contiguous run of sites (and no id 0xFFFFFFFF)");
    return TQ_OK;
}

// Stage 1 of a pass: ordering + site scan of quartets dq[0..n) into the count slab (n <= ctx->batch).
int stage_scan(tq_ctx *ctx, const uint32_t *dq, int64_t n, int subsample, bool input_sorted, hipStream_t stream)
{
    int rc = ensure_cm(ctx, n);
    if (rc) return rc;
    ctx->scanned_Q = 0;
    if ((rc = mark(ctx, TAG_ORIGIN, stream))) return rc;
    if (ctx->phases & 1) {
        if (use_dp(ctx, n, subsample, input_sorted)) {
            rc = make_units(ctx, dq, n, input_sorted, stream);
            if (rc) return rc;
            if ((rc = mark(ctx, TAG_ORDER, stream))) return rc;
            rc = launch_scan_dp(ctx, dq, n, stream);
        } else {
            const uint32_t *order = nullptr;
            rc = make_order(ctx, dq, n, input_sorted, stream, &order);
            if (rc) return rc;
            if ((rc = mark(ctx, TAG_ORDER, stream))) return rc;
            rc = launch_scan_n(ctx, dq, order, n, subsample, stream);
        }
        if (rc) return rc;
        if ((rc = mark(ctx, TAG_SCAN, stream))) return rc;
    }
    ctx->scanned_q = dq;
    ctx->scanned_Q = n;
    return TQ_OK;
}

// Rows per chunk of the singular-value stage for a range of n rows (the ONE place that decides it: stage_svd
// cuts by it and HostSink sizes its staging and picks its single-piece path by it).
int64_t svd_chunk_rows(const tq_ctx *ctx, int64_t n)
{
    int64_t chunk = ctx->svd_chunk < n ? ctx->svd_chunk : n;
    // a mid-size batch that would be one chunk is cut in two halves, one per stream: their tails fill each other
    // (125k quartets 1.58 -> 1.53 ms, 300k 3.45 -> 3.35 ms) and the host API gets a result piece to copy early
    if (ctx->svd_streams > 1 && n >= 65536 && n < 2 * ctx->svd_chunk && chunk > (n + 1) / 2) chunk = (n + 1) / 2;
    return chunk < 1 ? 1 : chunk;
}

// Stage 2 of a pass: singular values, ranks, scores and topology of rows [q0, q0+n) of the scanned
// batch, in chunks of svd_chunk quartets.  `out` points at the outputs of row q0.  Chunks alternate
// between the caller's stream and a second one (each with its own scratch set), so that the tail of one
// chunk's kernels -- a wave per 64 matrices with data-dependent iteration counts -- is filled by the next
// chunk's.  After each chunk `after_chunk(c0, cn, chunk_stream)` is called (c0 relative to q0) with the
// chunk's kernels enqueued on chunk_stream: the host-buffer API starts that chunk's result D2H there.
// On return the caller's stream has been made to wait for everything enqueued on the second stream.
template <typename F>
int stage_svd(tq_ctx *ctx, int64_t q0, int64_t n, bool debug, const OutPtrs &out, hipStream_t stream, F &&after_chunk)
{
    if (q0 < 0 || n < 0 || q0 + n > ctx->scanned_Q)
        return fail(ctx, TQ_ERR_INVALID_ARG, "rows [%lld,+%lld) are outside the scanned batch of %lld quartets",
                    (long long)q0, (long long)n, (long long)ctx->scanned_Q);
    const int64_t chunk = svd_chunk_rows(ctx, n);
    int rc = ensure_svd(ctx, chunk);
    if (rc) return rc;
    // timing-diagnostic modes produce wrong rows: none leaves the library unmarked
    const bool diag = diagnostic_mode(ctx);
    if (diag && !out.flags)
        return fail(ctx, TQ_ERR_INVALID_ARG, "a timing-diagnostic mode is set (scan_method 2..5 or phases 1 / 2): its rows are "
                                             "not results and are only handed out with a flags array (TQ_FLAG_INVALID_DIAGNOSTIC)");
    const bool two = ctx->svd_streams > 1 && n > chunk;
    if (two) {
        TQ_HIP(ctx, hipEventRecord(ctx->evFork, stream));
        TQ_HIP(ctx, hipStreamWaitEvent(ctx->sX, ctx->evFork, 0));
    }
    int64_t ci = 0;
    for (int64_t c0 = 0; c0 < n; c0 += chunk, ++ci) {
        const int64_t cn = (n - c0) < chunk ? (n - c0) : chunk;
        const int lane = two ? (int)(ci & 1) : 0;
        hipStream_t st = lane ? ctx->sX : stream;
        if (!(ctx->phases & 2)) {                   // no score kernel runs: flag the rows here
            if (hipMemsetAsync(out.flags + c0, TQ_FLAG_INVALID_DIAGNOSTIC, (size_t)cn, st) != hipSuccess) {
                rc = fail(ctx, TQ_ERR_HIP, "hipMemsetAsync(flags) failed");
                break;
            }
        }
        if (ctx->phases & 2) {
            const uint32_t *cm = ctx->d_cm + (size_t)(q0 + c0) * 256;
            const uint32_t *dq = ctx->scanned_q + (q0 + c0) * 4;
            OutPtrs o = offset_out(out, c0);
            o.flag_or = diag ? (uint32_t)TQ_FLAG_INVALID_DIAGNOSTIC : 0u;
            if ((rc = mark(ctx, TAG_ORIGIN, st, lane))) break;
            if (ctx->svd_method == 0)
                rc = debug ? launch_svd<true>(ctx, cm, dq, cn, o, st, lane) : launch_svd<false>(ctx, cm, dq, cn, o, st, lane);
            else
                rc = debug ? launch_hqr<true>(ctx, cm, dq, cn, o, st, lane) : launch_hqr<false>(ctx, cm, dq, cn, o, st, lane);
            if (rc) break;
        }
        if ((rc = after_chunk(c0, cn, st))) break;
    }
    if (two) {                                   // join, also on the error path: nothing may stay forked
        if (hipEventRecord(ctx->evJoin, ctx->sX) != hipSuccess || hipStreamWaitEvent(stream, ctx->evJoin, 0) != hipSuccess)
            if (!rc) rc = fail(ctx, TQ_ERR_HIP, "stream join failed");
    }
    return rc;
}

struct NoChunkHook {
    int operator()(int64_t, int64_t, hipStream_t) const { return TQ_OK; }
};

// One pass of the path over quartets dq[0..Q) with device outputs: scan batches of <= ctx->batch quartets,
// each followed by its singular-value stage.
template <typename F>
int launch(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, bool debug, bool input_sorted,
           const OutPtrs &out, hipStream_t stream, F &&after_chunk)
{
    int rc = check_ready(ctx, subsample);
    if (rc) return rc;
    if (Q == 0) return TQ_OK;
    if (ctx->timing) ctx->timed_calls++;
    const int64_t batch = Q < ctx->batch ? Q : ctx->batch;
    for (int64_t q0 = 0; q0 < Q; q0 += batch) {
        const int64_t n = (Q - q0) < batch ? (Q - q0) : batch;
        rc = stage_scan(ctx, dq + q0 * 4, n, subsample, input_sorted, stream);
        if (rc) return rc;
        rc = stage_svd(ctx, 0, n, debug, offset_out(out, q0), stream,
                       [&](int64_t c0, int64_t cn, hipStream_t st) { return after_chunk(q0 + c0, cn, st); });
        if (rc) return rc;
    }
    ctx->scanned_Q = 0;          // the slab belongs to this call only
    return TQ_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int ensure_streams(tq_ctx *ctx)
{
    if (!ctx->sK) {
        TQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->sK, hipStreamNonBlocking));
        TQ_HIP(ctx, hipStreamCreateWithFlags(&ctx->sC, hipStreamNonBlocking));
    }
    if (ctx->dev_api_pending) {                  // work of the device API may still be using the shared scratch
        TQ_HIP(ctx, hipStreamWaitEvent(ctx->sK, ctx->evDevApi, 0));
        ctx->dev_api_pending = false;
    }
    return TQ_OK;
}

// called before a device-API entry point enqueues work that touches the shared scratch (count slab, ordering and
// singular-value scratch, replicate layout): if the previous device-API call went to ANOTHER stream, this stream is
// made to wait for it -- two device-API calls of one context are ordered in call order whatever their streams
int enter_dev_api(tq_ctx *ctx, hipStream_t stream)
{
    if (ctx->dev_api_pending && ctx->evDevApi && stream != ctx->last_dev_stream)
        TQ_HIP(ctx, hipStreamWaitEvent(stream, ctx->evDevApi, 0));
    return TQ_OK;
}

// called after a device-API entry point has enqueued work on the caller's stream
int note_dev_api(tq_ctx *ctx, hipStream_t stream, int rc)
{
    if (!ctx->evDevApi) TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->evDevApi, hipEventDisableTiming));
    TQ_HIP(ctx, hipEventRecord(ctx->evDevApi, stream));
    ctx->dev_api_pending = true;
    ctx->last_dev_stream = stream;
    return rc;
}

int pipe_event(tq_ctx *ctx, size_t i, hipEvent_t *ev)
{
    try {
        while (ctx->pipe_events.size() <= i) {
            hipEvent_t e;
            TQ_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->pipe_events.push_back(e);
        }
    } catch (...) {
        return fail(ctx, TQ_ERR_OOM, "out of host memory");
    }
    *ev = ctx->pipe_events[i];
    return TQ_OK;
}

// Results of the host-buffer API: kernels on stream sK, result copies on stream sC.  The copy of chunk i
// (8 + 24 + 1 bytes per quartet) runs under the kernels of chunk i+1.  Destinations in page-locked
// memory (tq_host_alloc, hipHostMalloc, hipHostRegister) are written by the copy engine directly;
// pageable destinations go through two pinned staging pieces and one host memcpy per chunk, which the
// host does while the GPU works on the next chunk.
struct HostSink {
    tq_ctx *ctx;
    uint32_t *rstat;
    double *rscor;
    uint8_t *flags;
    const OutPtrs *dev;            // device outputs of row 0
    bool direct = false;
    bool single = false;           // pageable destinations, one piece: ONE D2H of the contiguous device region
    size_t off_rscor = 0, off_flags = 0;           // byte offsets of rscor / flags inside a staging piece
    char *stage[2] = {nullptr, nullptr};
    int64_t stage_rows = 0, total_rows = 0;
    int64_t single_q0 = 0, single_n = 0;
    hipStream_t single_stream = nullptr;
    struct Pending { int64_t q0, n; hipEvent_t done; int buf; };
    Pending pend[2];
    int npend = 0;
    size_t nchunk = 0;

    int begin(int64_t Q)
    {
        total_rows = Q;
        // (a call whose results fit one small piece is cheaper as ONE staged copy than as three direct ones)
        direct = (size_t)Q * 33 > ((size_t)1 << 20) && is_pinned(rstat, (size_t)Q * 8) &&
                 is_pinned(rscor, (size_t)Q * 24) && (!flags || is_pinned(flags, (size_t)Q));
        if (!direct) {
            // what stage_svd will cut this call into: one scan batch of min(Q, batch) rows at a time, each in
            // chunks of svd_chunk_rows(batch rows) -- the last batch may be shorter, never longer
            const int64_t first_batch = Q < ctx->batch ? Q : ctx->batch;
            stage_rows = svd_chunk_rows(ctx, first_batch);
            if (Q > first_batch) {
                const int64_t tail = svd_chunk_rows(ctx, Q % first_batch ? Q % first_batch : first_batch);
                if (tail > stage_rows) stage_rows = tail;
            }
            // small calls (the reference's distributor hands out chunks of a few thousand quartets,
            // run_inference.py:73-96) are dominated by the number of HIP calls: when the whole call is ONE
            // chunk, the device outputs [rstat | rscor | flags] are one contiguous region -> one copy
            single = Q <= ctx->batch && stage_rows >= Q && dev->flags &&
                     (const char *)dev->rscor > (const char *)dev->rstat && (const char *)dev->flags > (const char *)dev->rscor;
            off_rscor = single ? (size_t)((const char *)dev->rscor - (const char *)dev->rstat) : (size_t)stage_rows * 8;
            off_flags = single ? (size_t)((const char *)dev->flags - (const char *)dev->rstat) : (size_t)stage_rows * 32;
            const size_t bytes = off_flags + (size_t)stage_rows;
            for (int i = 0; i < (single ? 1 : 2); ++i)
                if (pool().alloc(bytes, (void **)&stage[i]) != TQ_OK)
                    return fail(ctx, TQ_ERR_OOM, "out of page-locked host memory for the result staging");
        }
        return TQ_OK;
    }
    int drain_one()
    {
        const Pending p = pend[0];
        pend[0] = pend[1];
        --npend;
        TQ_HIP(ctx, hipEventSynchronize(p.done));
        const char *s = stage[p.buf];
        memcpy(rstat + p.q0 * 2, s, (size_t)p.n * 8);
        memcpy(rscor + p.q0 * 3, s + off_rscor, (size_t)p.n * 24);
        if (flags) memcpy(flags + p.q0, s + off_flags, (size_t)p.n);
        return TQ_OK;
    }
    // the kernels of chunk [q0, q0+n) have been enqueued on `st`
    int chunk(int64_t q0, int64_t n, hipStream_t st)
    {
        if (!direct && n > stage_rows)
            return fail(ctx, TQ_ERR_HIP, "internal: result chunk of %lld rows exceeds the staging piece of %lld",
                        (long long)n, (long long)stage_rows);
        if (single) {
            // one piece, one copy, nothing to overlap with: the copy goes behind the kernels on their own stream
            // and finish() waits for that stream -- no events, no second stream (a 1 000-quartet call is a
            // dozen HIP calls; every one of them shows)
            if (q0 != 0 || n != total_rows || nchunk != 0)
                return fail(ctx, TQ_ERR_HIP, "internal: single-piece result path got chunk [%lld,+%lld) of %lld rows",
                            (long long)q0, (long long)n, (long long)total_rows);
            TQ_HIP(ctx, hipMemcpyAsync(stage[0], dev->rstat, off_flags + (size_t)n, hipMemcpyDeviceToHost, st));
            single_q0 = q0;
            single_n = n;
            single_stream = st;
            ++nchunk;
            return TQ_OK;
        }
        hipEvent_t ready, done;
        int rc = pipe_event(ctx, 2 * (nchunk % 4), &ready);
        if (!rc) rc = pipe_event(ctx, 2 * (nchunk % 4) + 1, &done);
        if (rc) return rc;
        TQ_HIP(ctx, hipEventRecord(ready, st));
        TQ_HIP(ctx, hipStreamWaitEvent(ctx->sC, ready, 0));
        if (direct) {
            TQ_HIP(ctx, hipMemcpyAsync(rstat + q0 * 2, dev->rstat + q0 * 2, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->sC));
            TQ_HIP(ctx, hipMemcpyAsync(rscor + q0 * 3, dev->rscor + q0 * 3, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->sC));
            if (flags)
                TQ_HIP(ctx, hipMemcpyAsync(flags + q0, dev->flags + q0, (size_t)n, hipMemcpyDeviceToHost, ctx->sC));
        } else {
            if (npend == 2) {                       // the staging piece this chunk needs is still in flight
                rc = drain_one();
                if (rc) return rc;
            }
            const int b = (int)(nchunk & 1);
            char *s = stage[b];
            TQ_HIP(ctx, hipMemcpyAsync(s, dev->rstat + q0 * 2, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->sC));
            TQ_HIP(ctx, hipMemcpyAsync(s + off_rscor, dev->rscor + q0 * 3, (size_t)n * 24, hipMemcpyDeviceToHost, ctx->sC));
            if (flags)
                TQ_HIP(ctx, hipMemcpyAsync(s + off_flags, dev->flags + q0, (size_t)n, hipMemcpyDeviceToHost, ctx->sC));
            TQ_HIP(ctx, hipEventRecord(done, ctx->sC));
            pend[npend++] = Pending{q0, n, done, b};
            if (npend == 2) {                       // copy out the older piece while the GPU works on
                rc = drain_one();
                if (rc) return rc;
            }
        }
        ++nchunk;
        return TQ_OK;
    }
    int finish()
    {
        int rc = TQ_OK;
        if (single && nchunk) {
            TQ_HIP(ctx, hipStreamSynchronize(single_stream));
            const char *s = stage[0];
            memcpy(rstat + single_q0 * 2, s, (size_t)single_n * 8);
            memcpy(rscor + single_q0 * 3, s + off_rscor, (size_t)single_n * 24);
            if (flags) memcpy(flags + single_q0, s + off_flags, (size_t)single_n);
            return TQ_OK;
        }
        while (npend && !rc) rc = drain_one();
        if (!rc && hipStreamSynchronize(ctx->sC) != hipSuccess) rc = fail(ctx, TQ_ERR_HIP, "result copy failed");
        return rc;
    }
    ~HostSink()
    {
        // never leave copies in flight into memory the caller may free
        (void)hipStreamSynchronize(ctx->sK);
        (void)hipStreamSynchronize(ctx->sC);
        for (int i = 0; i < 2; ++i)
            if (stage[i]) (void)pool().release(stage[i]);
    }
};

struct NoHostWork {
    int operator()() const { return TQ_OK; }
};

// device quartets -> host results (synchronous).  `while_gpu_works()` runs on the host after everything has been
// enqueued and before the results are waited for (a non-zero return aborts the call after the streams have drained).
template <typename F = NoHostWork>
int resolve_to_host(tq_ctx *ctx, const uint32_t *dq, int64_t Q, int subsample, bool input_sorted, uint32_t *rstat,
                    double *rscor, uint8_t *flags, uint32_t *d_rstat, double *d_rscor, uint8_t *d_flags,
                    F &&while_gpu_works = F())
{
    OutPtrs out{d_rstat, d_rscor, d_flags, nullptr, nullptr, nullptr};
    HostSink sink{ctx, rstat, rscor, flags, &out};
    int rc = sink.begin(Q);
    if (rc) return rc;
    rc = launch(ctx, dq, Q, subsample, false, input_sorted, out, ctx->sK,
                [&](int64_t q0, int64_t n, hipStream_t st) { return sink.chunk(q0, n, st); });
    if (rc) return rc;
    if ((rc = while_gpu_works())) return rc;          // ~HostSink drains the streams
    return sink.finish();
}

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
    if (int rc = probe_scan_pb(ctx)) {
        g_create_err = ctx->err;
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
    if (ctx->d_cm) (void)hipFree(ctx->d_cm);
    if (ctx->d_sort) (void)hipFree(ctx->d_sort);
    if (ctx->d_sort_tmp) (void)hipFree(ctx->d_sort_tmp);
    if (ctx->d_units) (void)hipFree(ctx->d_units);
    if (ctx->d_de) (void)hipFree(ctx->d_de);
    if (ctx->d_sv) (void)hipFree(ctx->d_sv);
    if (ctx->d_nsnps) (void)hipFree(ctx->d_nsnps);
    if (ctx->d_bdsqr_stats) (void)hipFree(ctx->d_bdsqr_stats);
    if (ctx->sK) (void)hipStreamDestroy(ctx->sK);
    if (ctx->sC) (void)hipStreamDestroy(ctx->sC);
    if (ctx->sX) {
        (void)hipStreamDestroy(ctx->sX);
        (void)hipEventDestroy(ctx->evFork);
        (void)hipEventDestroy(ctx->evJoin);
    }
    if (ctx->evDevApi) (void)hipEventDestroy(ctx->evDevApi);
    for (auto e : ctx->pipe_events) (void)hipEventDestroy(e);
    for (auto &m : ctx->marks) (void)hipEventDestroy(m.ev);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    delete ctx;
}

const char *tq_last_error(const tq_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int tq_host_alloc(int64_t bytes, void **out)
{
    if (!out || bytes < 0) return TQ_ERR_INVALID_ARG;
    *out = nullptr;
    return pool().alloc((size_t)bytes, out);
}

int tq_host_free(void *p)
{
    if (!p) return TQ_OK;
    return pool().release(p);
}

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
    std::vector<uint32_t> loc;
    bool ok = true, sorted = true;
    try {
        loc.resize((size_t)S);
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
    } catch (const std::bad_alloc &) {
        return fail(ctx, TQ_ERR_OOM, "tq_set_data: out of host memory");
    }
    ctx->locus_runs_ok = ok;

    const int64_t Sp = (int64_t)align_up((size_t)S, TILE);
    const int64_t W = Sp / 32;
    ctx->T = T; ctx->S = S; ctx->Sp = Sp; ctx->W = W;
    ctx->data_capacity = Sp;
    uint8_t *d_raw = nullptr;
    uint32_t *d_loc = nullptr;
    // every allocation of this call is released on every failure path (the resident arrays by free_data,
    // the two upload temporaries here)
    hipError_t e = hipMalloc((void **)&ctx->d_rows, (size_t)(T * Sp));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_nib, (size_t)(T * Sp / 2));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_nib5, (size_t)(T * Sp / 2));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_planes, (size_t)(T * W) * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_planes3, (size_t)(T * W * 3 + W) * sizeof(uint32_t));
    ctx->plane_cap_W = W;
    if (e == hipSuccess) e = hipMalloc((void **)&d_raw, (size_t)(T * S));
    if (e == hipSuccess) e = hipMalloc((void **)&d_loc, (size_t)S * sizeof(uint32_t));
    if (e != hipSuccess) {
        if (d_raw) (void)hipFree(d_raw);
        if (d_loc) (void)hipFree(d_loc);
        free_data(ctx);
        return fail(ctx, e == hipErrorOutOfMemory ? TQ_ERR_OOM : TQ_ERR_HIP, "tq_set_data: hipMalloc failed: %s",
                    hipGetErrorString(e));
    }
    e = hipMemcpy(d_raw, tmparr, (size_t)(T * S), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_loc, loc.data(), (size_t)S * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int64_t n = T * W;
        hipLaunchKernelGGL(tq_prepare_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_raw, d_loc, S, Sp,
                           W, (int32_t)T, ctx->d_rows, ctx->d_nib, ctx->d_nib5, ctx->d_planes,
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
    if (int rc = enter_dev_api(ctx, (hipStream_t)stream)) return rc;
    return note_dev_api(ctx, (hipStream_t)stream,
                        launch(ctx, d_quartets, Q, subsample, false, false, out, (hipStream_t)stream, NoChunkHook()));
}

int tq_scan_dev(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, int subsample, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (Q < 1 || !d_quartets) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_scan_dev: NULL pointer or Q < 1");
    if (Q > ctx->batch)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_scan_dev: Q=%lld exceeds the scan batch of %lld quartets (option 'batch')",
                    (long long)Q, (long long)ctx->batch);
    int rc = check_ready(ctx, subsample);
    if (rc) return rc;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->timing) ctx->timed_calls++;
    if ((rc = enter_dev_api(ctx, (hipStream_t)stream))) return rc;
    return note_dev_api(ctx, (hipStream_t)stream, stage_scan(ctx, d_quartets, Q, subsample, false, (hipStream_t)stream));
}

int tq_svd_dev(tq_ctx *ctx, int64_t q0, int64_t n, uint32_t *d_rstat, double *d_rscor, uint8_t *d_flags, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!d_rstat || !d_rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_svd_dev: NULL pointer or negative n");
    if (ctx->scanned_Q == 0) return fail(ctx, TQ_ERR_NO_DATA, "tq_svd_dev: no scanned batch (call tq_scan_dev first)");
    if (n == 0) return TQ_OK;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    OutPtrs out{d_rstat, d_rscor, d_flags, nullptr, nullptr, nullptr};
    if (int rc = enter_dev_api(ctx, (hipStream_t)stream)) return rc;
    return note_dev_api(ctx, (hipStream_t)stream, stage_svd(ctx, q0, n, false, out, (hipStream_t)stream, NoChunkHook()));
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
    // consecutive lexicographic ranks are in (a,b,c) order already
    if (int rc2 = enter_dev_api(ctx, (hipStream_t)stream)) return rc2;
    return note_dev_api(ctx, (hipStream_t)stream,
                        launch(ctx, dq, Q, subsample, false, true, out, (hipStream_t)stream, NoChunkHook()));
}

int tq_resolve_to_host(tq_ctx *ctx, const uint32_t *d_quartets, int64_t Q, int subsample, uint32_t *rstat,
                       double *rscor, uint8_t *flags)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (Q < 0 || (Q > 0 && (!d_quartets || !rstat || !rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve_to_host: NULL pointer or negative Q");
    int rc = check_ready(ctx, subsample);
    if (rc) return rc;
    if (Q == 0) return TQ_OK;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_streams(ctx))) return rc;
    const size_t o_rstat = 0;
    const size_t o_rscor = align_up(o_rstat + (size_t)Q * 8, 256);
    const size_t o_flags = align_up(o_rscor + (size_t)Q * 24, 256);
    if ((rc = ensure_scratch(ctx, align_up(o_flags + (size_t)Q, 256)))) return rc;
    char *base = (char *)ctx->d_scratch;
    return resolve_to_host(ctx, d_quartets, Q, subsample, false, rstat, rscor, flags, (uint32_t *)(base + o_rstat),
                           (double *)(base + o_rscor), (uint8_t *)(base + o_flags));
}

int tq_resolve_debug(tq_ctx *ctx, const uint32_t *quartets, int64_t Q, int subsample, uint32_t *rstat,
                     double *rscor, uint8_t *flags, uint32_t *cmats, double *svds, int32_t *ranks)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (Q < 0 || (Q > 0 && (!quartets || !rstat || !rscor)))
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_resolve: NULL pointer or negative Q");
    int rc = check_ready(ctx, subsample);
    if (rc) return rc;
    if (Q == 0) return TQ_OK;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    // Taxon indices are checked on the host (the kernels re-check and flag them, and never dereference a bad
    // one).  For chunks of the size the reference's distributor hands out the same pass notices input that
    // already is in (a,b,c) order -- its default mode's lexicographic chunks are (combinations.py:40-55) -- so
    // that the device sort (about twenty small kernels, most of the time of a chunk of a few thousand
    // quartets) is skipped.  A large batch is checked WHILE THE GPU WORKS on it (the pass over 16 B per quartet
    // costs ~1 ms per 1e6 on the host, the sort it could save 0.2 ms).
    constexpr int64_t CHECK_FIRST = 1 << 16;
    auto check_indices = [ctx, quartets, Q]() -> int {
        const uint32_t T = (uint32_t)ctx->T;
        uint32_t worst = 0;
        for (int64_t i = 0; i < 4 * Q; ++i) worst = quartets[i] > worst ? quartets[i] : worst;      // vectorises
        if (worst < T) return TQ_OK;
        for (int64_t i = 0; i < 4 * Q; ++i)
            if (quartets[i] >= T)
                return fail(ctx, TQ_ERR_INVALID_ARG, "quartet %lld has taxon index %u >= T=%lld", (long long)(i / 4),
                            quartets[i], (long long)ctx->T);
        return TQ_OK;
    };
    bool input_sorted = false;
    if (Q <= CHECK_FIRST) {
        if ((rc = check_indices())) return rc;
        bool sorted = true;
        uint64_t prev_key = 0;
        for (int64_t i = 0; i < Q; ++i) {
            const uint32_t *q = quartets + i * 4;
            const uint64_t key = ((uint64_t)q[0] << 42) | ((uint64_t)q[1] << 21) | (uint64_t)q[2];
            sorted &= key >= prev_key;
            prev_key = key;
        }
        input_sorted = sorted && ctx->T < (1 << 21);
    }
    const bool debug = cmats || svds || ranks;
    const size_t o_q = 0;
    const size_t o_rstat = align_up(o_q + (size_t)Q * 16, 256);
    const size_t o_rscor = align_up(o_rstat + (size_t)Q * 8, 256);
    const size_t o_flags = align_up(o_rscor + (size_t)Q * 24, 256);
    const size_t o_cm = align_up(o_flags + (size_t)Q, 256);
    const size_t o_sv = align_up(o_cm + (cmats ? (size_t)Q * 3072 : 0), 256);
    const size_t o_rk = align_up(o_sv + (svds ? (size_t)Q * 384 : 0), 256);
    const size_t total = align_up(o_rk + (ranks ? (size_t)Q * 12 : 0), 256);
    if ((rc = ensure_scratch(ctx, total))) return rc;
    if ((rc = ensure_streams(ctx))) return rc;
    char *base = (char *)ctx->d_scratch;
    // quartets H2D on the compute stream (asynchronous when the caller's array is page-locked)
    TQ_HIP(ctx, hipMemcpyAsync(base + o_q, quartets, (size_t)Q * 16, hipMemcpyHostToDevice, ctx->sK));
    if (!debug) {
        if (Q <= CHECK_FIRST)
            return resolve_to_host(ctx, (const uint32_t *)(base + o_q), Q, subsample, input_sorted, rstat, rscor, flags,
                                   (uint32_t *)(base + o_rstat), (double *)(base + o_rscor), (uint8_t *)(base + o_flags));
        return resolve_to_host(ctx, (const uint32_t *)(base + o_q), Q, subsample, false, rstat, rscor, flags,
                               (uint32_t *)(base + o_rstat), (double *)(base + o_rscor), (uint8_t *)(base + o_flags),
                               check_indices);
    }
    if (Q > CHECK_FIRST && (rc = check_indices())) {
        (void)hipStreamSynchronize(ctx->sK);
        return rc;
    }
    OutPtrs out{};
    out.rstat = (uint32_t *)(base + o_rstat);
    out.rscor = (double *)(base + o_rscor);
    out.flags = (uint8_t *)(base + o_flags);
    out.cmats = cmats ? (uint32_t *)(base + o_cm) : nullptr;
    out.svds = svds ? (double *)(base + o_sv) : nullptr;
    out.ranks = ranks ? (int32_t *)(base + o_rk) : nullptr;
    rc = launch(ctx, (const uint32_t *)(base + o_q), Q, subsample, true, input_sorted, out, ctx->sK, NoChunkHook());
    if (rc) {
        (void)hipStreamSynchronize(ctx->sK);
        return rc;
    }
    TQ_HIP(ctx, hipStreamSynchronize(ctx->sK));
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

int tq_timing_read_kernels(tq_ctx *ctx, double *ms, int n_ms, int64_t *calls)
{
    if (!ctx || (n_ms > 0 && !ms)) return TQ_ERR_INVALID_ARG;
    double t[TAG_COUNT] = {0, 0, 0, 0, 0};
    hipEvent_t prev[2] = {nullptr, nullptr};
    int rc = TQ_OK;
    for (auto &m : ctx->marks) {
        if (!rc && hipEventSynchronize(m.ev) != hipSuccess) rc = fail(ctx, TQ_ERR_HIP, "timing event failed");
        if (!rc && m.tag >= 0 && m.tag < TAG_COUNT && prev[m.lane]) {
            float a = 0.f;
            if (hipEventElapsedTime(&a, prev[m.lane], m.ev) == hipSuccess) t[m.tag] += a;
        }
        prev[m.lane] = m.ev;
        ctx->event_pool.push_back(m.ev);
    }
    ctx->marks.clear();
    for (int i = 0; i < n_ms; ++i) ms[i] = i < TAG_COUNT ? t[i] : 0.0;
    if (calls) *calls = ctx->timed_calls;
    ctx->timed_calls = 0;
    return rc;
}

int tq_timing_read_split(tq_ctx *ctx, double *total_ms, double *scan_ms, double *svd_ms, int64_t *calls)
{
    double t[TAG_COUNT];
    const int rc = tq_timing_read_kernels(ctx, t, TAG_COUNT, calls);
    if (rc) return rc;
    const double a = t[TAG_ORDER] + t[TAG_SCAN], b = t[TAG_BIDIAG] + t[TAG_BDSQR] + t[TAG_SCORE];
    if (total_ms) *total_ms = a + b;
    if (scan_ms) *scan_ms = a;
    if (svd_ms) *svd_ms = b;
    return TQ_OK;
}

int tq_timing_read(tq_ctx *ctx, double *kernel_ms, int64_t *launches)
{
    return tq_timing_read_split(ctx, kernel_ms, nullptr, nullptr, launches);
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
    if (!strcmp(name, "xcd_remap")) {
        ctx->xcd_remap = value != 0;
        return TQ_OK;
    }
    if (!strcmp(name, "count_invariant")) {
        ctx->count_invariant = value != 0;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_pair")) {
        ctx->scan_pair = value != 0;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_f4")) {
        if (value < -1 || value > 1) return fail(ctx, TQ_ERR_INVALID_ARG, "scan_f4 must be -1 (subsample mode only), 0 or 1");
        ctx->scan_f4 = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_dp")) {
        if (value != 0 && value != 1) return fail(ctx, TQ_ERR_INVALID_ARG, "scan_dp must be 0 or 1");
        ctx->scan_dp = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "dp_min_quartets")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "dp_min_quartets must be >= 0");
        ctx->dp_min_quartets = value ? value : 32768;
        return TQ_OK;
    }
    if (!strcmp(name, "wg_min_quartets")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "wg_min_quartets must be >= 0");
        ctx->wg_min_quartets = value ? value : 4096;
        return TQ_OK;
    }
    if (!strcmp(name, "bidiag_layout")) {
        if (value < -1 || value > 1) return fail(ctx, TQ_ERR_INVALID_ARG, "bidiag_layout must be -1 (default), 0 or 1");
        ctx->bidiag_layout = value < 0 ? 1 : (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "park_t")) {
        ctx->park_t = value != 0;
        return TQ_OK;
    }
    if (!strcmp(name, "share_c")) {
        ctx->share_c = value != 0;
        return TQ_OK;
    }
    if (!strcmp(name, "svd_wpc")) {
        if (value < 0 || value > (1 << 20)) return fail(ctx, TQ_ERR_INVALID_ARG, "svd_wpc must be 0..2^20");
        ctx->svd_wpc = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "svd_chunk")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "svd_chunk must be >= 0");
        ctx->svd_chunk = value ? value : (1 << 18);
        return TQ_OK;
    }
    if (!strcmp(name, "svd_streams")) {
        if (value < 0 || value > 2) return fail(ctx, TQ_ERR_INVALID_ARG, "svd_streams must be 0 (default), 1 or 2");
        ctx->svd_streams = value ? (int)value : 2;
        return TQ_OK;
    }
    if (!strcmp(name, "bdsqr_stats")) {                  // 1: allocate + zero the counters, 0: free them
        if (ctx->d_bdsqr_stats) (void)hipFree(ctx->d_bdsqr_stats);
        ctx->d_bdsqr_stats = nullptr;
        if (value) {
            TQ_HIP(ctx, hipMalloc((void **)&ctx->d_bdsqr_stats, 8 * sizeof(uint64_t)));
            TQ_HIP(ctx, hipMemset(ctx->d_bdsqr_stats, 0, 8 * sizeof(uint64_t)));
        }
        return TQ_OK;
    }
    if (!strcmp(name, "bdsqr_maxit")) {
        if (value < 0 || value > 1000) return fail(ctx, TQ_ERR_INVALID_ARG, "bdsqr_maxit must be 0..1000");
        ctx->bdsqr_maxit = value ? (int)value : 60;
        return TQ_OK;
    }
    if (!strcmp(name, "scan_wg")) {
        if (value != 0 && value != 1 && value != 2 && value != 3 && value != 4 && value != 6 && value != 8 && value != 16)
            return fail(ctx, TQ_ERR_INVALID_ARG, "scan_wg must be 0 (default), 1, 2, 3, 4, 6, 8 or 16");
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
        if (value < -1 || value > 6)
            return fail(ctx, TQ_ERR_INVALID_ARG, "scan_method must be -1 (auto), 0, 1, 6 (or 2..5: timing diagnostics)");
        ctx->scan_method = (int)value;
        return TQ_OK;
    }
    if (!strcmp(name, "batch")) {
        if (value < 0) return fail(ctx, TQ_ERR_INVALID_ARG, "batch must be >= 0");
        if (value > 0x7FFFFFFFll) value = 0x7FFFFFFFll;      // item counts reach hipCUB as int
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
    try {
        ctx->h_spans.assign(spans, spans + 2 * nloci);
    } catch (const std::bad_alloc &) {
        return fail(ctx, TQ_ERR_OOM, "tq_set_source: out of host memory");
    }
    for (int i = 0; i < 2; ++i) {
        if (pool().alloc((size_t)nloci * 8, (void **)&ctx->h_lidx_stage[i]) != TQ_OK)
            return fail(ctx, TQ_ERR_OOM, "tq_set_source: out of page-locked host memory");
        TQ_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_lidx[i], hipEventDisableTiming));
    }
    ctx->src_T = T;
    ctx->src_S0 = S0;
    ctx->nloci = nloci;
    ctx->max_width = maxw;
    return TQ_OK;
}

int tq_bootstrap_async(tq_ctx *ctx, const int64_t *lidxs, int64_t n, uint64_t seed_shuffle, uint64_t seed_ambig,
                       int64_t *out_S, void *stream_)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->d_seqarr) return fail(ctx, TQ_ERR_NO_DATA, "tq_set_source has not been called");
    if (!lidxs || n != ctx->nloci)
        return fail(ctx, TQ_ERR_INVALID_ARG, "tq_bootstrap: lidxs must hold nloci=%lld locus indices", (long long)ctx->nloci);
    hipStream_t stream = (hipStream_t)stream_;
    // replicate length on the host (jit/resample.py:7-17): no device round trip, the call stays asynchronous
    int64_t S = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (lidxs[i] < 0 || lidxs[i] >= ctx->nloci)
            return fail(ctx, TQ_ERR_INVALID_ARG, "tq_bootstrap: locus index %lld out of range", (long long)lidxs[i]);
        S += ctx->h_spans[2 * lidxs[i] + 1] - ctx->h_spans[2 * lidxs[i]];
    }
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc0 = enter_dev_api(ctx, stream)) return rc0;      // a resolve still scanning the previous replicate on another stream
    const int64_t T = ctx->src_T;
    // worst-case replicate length; buffers grow only
    const int64_t cap = n * ctx->max_width;
    if (cap > ctx->boot_cap) {
        TQ_HIP(ctx, hipDeviceSynchronize());
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
    if (S < 1 || S > ctx->boot_cap) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_bootstrap: replicate length %lld", (long long)S);
    const int64_t Sp = (int64_t)align_up((size_t)S, TILE);
    const int64_t W = Sp / 32;
    if (Sp > ctx->data_capacity || T != ctx->T) {
        TQ_HIP(ctx, hipDeviceSynchronize());           // kernels of the previous replicate may still read the old buffers
        free_data(ctx);
        const int64_t capSp = (int64_t)align_up((size_t)(Sp + Sp / 8), TILE);   // head-room: replicate lengths vary
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_rows, (size_t)(T * capSp)));
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_nib, (size_t)(T * capSp / 2)));
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_nib5, (size_t)(T * capSp / 2)));
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)(T * (capSp / 32)) * sizeof(uint4)));
        TQ_HIP(ctx, hipMalloc((void **)&ctx->d_planes3, (size_t)(T * (capSp / 32) * 3 + capSp / 32) * sizeof(uint32_t)));
        ctx->plane_cap_W = capSp / 32;
        ctx->data_capacity = capSp;
    }
    uint32_t *widths = ctx->d_boot, *offsets = widths + (n + 1);
    uint32_t *src_col = offsets + (n + 1), *site_locus = src_col + ctx->boot_cap;
    // locus indices through a page-locked staging piece (two in turn, so that the draws of the next replicate
    // can be handed over while this copy is still queued)
    const unsigned turn = ctx->lidx_turn++ & 1u;
    TQ_HIP(ctx, hipEventSynchronize(ctx->ev_lidx[turn]));
    memcpy(ctx->h_lidx_stage[turn], lidxs, (size_t)n * 8);
    TQ_HIP(ctx, hipMemcpyAsync(ctx->d_lidxs, ctx->h_lidx_stage[turn], (size_t)n * 8, hipMemcpyHostToDevice, stream));
    TQ_HIP(ctx, hipEventRecord(ctx->ev_lidx[turn], stream));
    TQ_HIP(ctx, hipMemsetAsync(widths + n, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(tq_boot_width_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ctx->d_spans,
                       ctx->d_lidxs, n, ctx->nloci, widths);
    size_t tmp = ctx->boot_tmp_bytes;
    TQ_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(ctx->d_boot_tmp, tmp, widths, offsets, (int)(n + 1), stream));
    hipLaunchKernelGGL(tq_boot_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ctx->d_spans,
                       ctx->d_lidxs, offsets, n, ctx->nloci, seed_shuffle, src_col, site_locus);
    const int64_t nw = T * W;
    hipLaunchKernelGGL(tq_boot_build_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, stream, ctx->d_seqarr,
                       ctx->src_S0, src_col, site_locus, S, Sp, W, (int32_t)T, seed_ambig, ctx->d_rows, ctx->d_nib, ctx->d_nib5,
                       ctx->d_planes, ctx->d_planes3, ctx->d_planes3 + (size_t)T * (size_t)ctx->plane_cap_W * 3);
    TQ_HIP(ctx, hipGetLastError());
    ctx->T = T;
    ctx->S = S;
    ctx->Sp = Sp;
    ctx->W = W;
    ctx->have_data = true;
    ctx->locus_runs_ok = true;          // locus ids are the ordinals 0..n-1, one run each
    ctx->scanned_Q = 0;
    if (out_S) *out_S = S;
    return note_dev_api(ctx, stream, TQ_OK);       // the host API must not scan a half-built replicate
}

int tq_bootstrap(tq_ctx *ctx, const int64_t *lidxs, int64_t n, uint64_t seed_shuffle, uint64_t seed_ambig,
                 int64_t *out_S)
{
    const int rc = tq_bootstrap_async(ctx, lidxs, n, seed_shuffle, seed_ambig, out_S, nullptr);
    if (rc) return rc;
    TQ_HIP(ctx, hipStreamSynchronize(nullptr));
    return TQ_OK;
}

int tq_sample_quartets_dev(tq_ctx *ctx, uint64_t seed, int64_t Q, uint64_t *d_ranks, uint32_t *d_quartets, void *stream)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data && !ctx->d_seqarr) return fail(ctx, TQ_ERR_NO_DATA, "no data on the device (T unknown)");
    if (Q < 0 || (Q > 0 && !d_quartets)) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_sample_quartets_dev: NULL pointer or negative Q");
    const uint64_t T = (uint64_t)(ctx->have_data ? ctx->T : ctx->src_T);
    const uint64_t total = T < 4 ? 0 : T * (T - 1) / 2 * (T - 2) / 3 * (T - 3) / 4;
    if ((uint64_t)Q > total)
        return fail(ctx, TQ_ERR_INVALID_ARG, "cannot draw %lld distinct quartets from C(%llu,4)=%llu", (long long)Q,
                    (unsigned long long)T, (unsigned long long)total);
    if (Q == 0) return TQ_OK;
    int half = 1;
    while (half < 32 && (1ull << (2 * half)) < total) ++half;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(tq_sample_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, total,
                       half, Q, (int32_t)T, d_ranks, d_quartets);
    TQ_HIP(ctx, hipGetLastError());
    return TQ_OK;
}

int tq_get_data(tq_ctx *ctx, uint8_t *tmparr, uint32_t *tmpmap)
{
    if (!ctx) return TQ_ERR_INVALID_ARG;
    if (!ctx->have_data) return fail(ctx, TQ_ERR_NO_DATA, "no replicate on the device");
    if (!tmparr || !tmpmap) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_get_data: NULL pointer");
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    TQ_HIP(ctx, hipDeviceSynchronize());     // a replicate may still be queued (tq_bootstrap_async) on any stream
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
                      : which == 2 ? (const void *)ctx->d_sv : which == 3 ? (const void *)ctx->d_bdsqr_stats : nullptr;
    if (!src) return fail(ctx, TQ_ERR_INVALID_ARG, "tq_debug_fetch: nothing to fetch (which=%d)", which);
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    TQ_HIP(ctx, hipDeviceSynchronize());
    TQ_HIP(ctx, hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return TQ_OK;
}

int tq_debug_bdsqr(tq_ctx *ctx, const double *de, int64_t nmat, double *sv, uint32_t *work, int reps, double *ms)
{
    if (!ctx || !de || nmat < 1 || reps < 1) return TQ_ERR_INVALID_ARG;
    TQ_HIP(ctx, hipSetDevice(ctx->device));
    double *d_de = nullptr, *d_sv = nullptr;
    uint32_t *d_work = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = TQ_OK;
    auto step = [&](hipError_t e, const char *what) {
        if (!rc && e != hipSuccess) rc = fail(ctx, e == hipErrorOutOfMemory ? TQ_ERR_OOM : TQ_ERR_HIP, "tq_debug_bdsqr: %s: %s", what, hipGetErrorString(e));
    };
    step(hipMalloc(&d_de, (size_t)nmat * 256), "hipMalloc");
    if (!rc) step(hipMalloc(&d_sv, (size_t)nmat * 128), "hipMalloc");
    if (!rc) step(hipMalloc(&d_work, (size_t)nmat * 4), "hipMalloc");
    if (!rc) step(hipMemcpy(d_de, de, (size_t)nmat * 256, hipMemcpyHostToDevice), "H2D");
    if (!rc) step(hipEventCreate(&e0), "event");
    if (!rc) step(hipEventCreate(&e1), "event");
    const unsigned grid = (unsigned)((nmat + WAVE - 1) / WAVE);
    if (!rc) {
        // once with the per-matrix counters (slower), then `reps` timed launches of the product form
        hipLaunchKernelGGL(tq_bdsqr_kernel, dim3(grid), dim3(WAVE), 0, 0, (const double *)d_de, nmat, d_sv, ctx->bdsqr_maxit,
                           (unsigned long long *)nullptr, d_work);
        step(hipGetLastError(), "launch");
        if (!rc) step(hipEventRecord(e0, 0), "record");
        for (int i = 0; i < reps && !rc; ++i)
            hipLaunchKernelGGL(tq_bdsqr_kernel, dim3(grid), dim3(WAVE), 0, 0, (const double *)d_de, nmat, d_sv, ctx->bdsqr_maxit,
                               (unsigned long long *)nullptr, (uint32_t *)nullptr);
        if (!rc) step(hipEventRecord(e1, 0), "record");
        if (!rc) step(hipEventSynchronize(e1), "sync");
        float t = 0.f;
        if (!rc) step(hipEventElapsedTime(&t, e0, e1), "elapsed");
        if (ms) *ms = (double)t / reps;
    }
    if (!rc && sv) step(hipMemcpy(sv, d_sv, (size_t)nmat * 128, hipMemcpyDeviceToHost), "D2H");
    if (!rc && work) step(hipMemcpy(work, d_work, (size_t)nmat * 4, hipMemcpyDeviceToHost), "D2H");
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_de) (void)hipFree(d_de);
    if (d_sv) (void)hipFree(d_sv);
    if (d_work) (void)hipFree(d_work);
    return rc;
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

int tq_qmc_splits(const uint32_t *quartets, const uint32_t *rstat, const double *rscor, int64_t Q, int weights,
                  int64_t min_snps, double min_ratio, uint32_t *splits, double *wout, int64_t *n_rows)
{
    if (Q < 0 || !n_rows || weights < 0 || weights > 3 || (Q > 0 && (!quartets || !rstat || !rscor || !splits || !wout)))
        return TQ_ERR_INVALID_ARG;
    *n_rows = qmc_splits(quartets, rstat, rscor, Q, weights, min_snps, min_ratio, splits, wout);
    return TQ_OK;
}

int tq_numpy_choice_tail(void *np_bitgen, uint64_t pop, int64_t size, int64_t *out)
{
    if (!np_bitgen || !out || size < 1 || (uint64_t)size > pop || pop < 2 || pop > 0xFFFFFFFEull) return TQ_ERR_INVALID_ARG;
    return numpy_choice_tail((NpBitgen *)np_bitgen, pop, size, out);
}

int tq_unrank(const uint64_t *ranks, uint64_t first_rank, int64_t Q, int64_t T, uint32_t *quartets)
{
    if (Q < 0 || T < 4 || T > 100000 || (Q > 0 && !quartets)) return TQ_ERR_INVALID_ARG;     // C(T,4) must fit 64 bits
    const uint64_t t = (uint64_t)T, total = t * (t - 1) / 2 * (t - 2) / 3 * (t - 3) / 4;
    if (ranks) {
        for (int64_t i = 0; i < Q; ++i)
            if (ranks[i] >= total) return TQ_ERR_INVALID_ARG;
    } else if (first_rank + (uint64_t)Q > total) {
        return TQ_ERR_INVALID_ARG;
    }
    unrank_host(ranks, first_rank, Q, (int32_t)T, quartets);
    return TQ_OK;
}

int tq_qmc_tree(const uint32_t *splits, const double *weights, int64_t n, int64_t ntaxa, uint64_t seed, char *out,
                int64_t cap, int64_t *written)
{
    if (n < 0 || cap < 0 || !written || ntaxa < 1 || ntaxa > (1 << 24) || (n > 0 && !splits) || (cap > 0 && !out))
        return TQ_ERR_INVALID_ARG;
    try {
        std::string nwk;
        const int rc = qmc_tree(splits, weights, n, ntaxa, seed, nwk);
        if (rc) return rc;
        *written = (int64_t)nwk.size();
        if ((int64_t)nwk.size() > cap) return TQ_ERR_OOM;
        memcpy(out, nwk.data(), nwk.size());
        return TQ_OK;
    } catch (const std::bad_alloc &) {
        *written = 0;
        return TQ_ERR_OOM;
    }
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

