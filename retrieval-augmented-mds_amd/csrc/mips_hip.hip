// C ABI of the MI355X MIPS backend (include/mips_hip.h): index storage in HBM, host-side
// orchestration of the fused scan + merge kernels.  gfx950 only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include <cstdlib>

#include "../../include/mips_hip.h"
#include "aux_kernels.hpp"
#include "scan_kernel.hpp"
#include "scan_kernel_v3.hpp"
#include "scan_kernel_f8.hpp"
#include "scan_kernel_f8x.hpp"
#include "scan_kernel_v4.hpp"
#ifdef MIPS_EXPERIMENTAL // measured alternatives that never became a default (profiles/r2_v5_64q, r2_pitch1024): A/B library only
#include "scan_kernel_v5.hpp"
#include "scan_kernel_ks.hpp"
#endif
#include "scan_kernel_k3.hpp"
#include "scan_kernel_e8.hpp"
#include "tiny_search.hpp"
#include "resolve_kernels.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? MIPS_E_NOMEM : MIPS_E_HIP, "%s failed: %s", \
                        #expr, hipGetErrorString(e_));                                         \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
constexpr int64_t kRowAlign = 256;   // index capacity granule: the largest document tile of any scan variant
constexpr int64_t kQueryAlign = 256; // query staging buffer granule: the largest power-of-two query tile of any scan variant

struct Buffer {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return MIPS_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(MIPS_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        bytes = want;
        return MIPS_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

int grid_for(int64_t items, int block) {
    int64_t g = (items + block - 1) / block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, 256 * 16));
}

} // namespace

struct mips_index {
    int device = 0;
    int64_t d = 0;
    int ld = 0;    // row length in elements: d padded to a multiple of 64 (bf16) or 256 (fp8)
    int esize = 2; // bytes per stored element
    int qsize = 2; // bytes per STAGED query element: esize, except MIPS_DTYPE_FP8_E4M3_DOCS (e4m3 rows, bf16 queries: `mixed`)
    bool mixed = false;
    int doc_dtype = MIPS_DTYPE_BF16;
    int metric = MIPS_METRIC_IP;
    int64_t ntotal = 0;
    int64_t capacity = 0; // rows allocated, multiple of TM
    uint8_t* rows = nullptr; // [capacity][ld] elements of esize bytes
    // fp32-exact mode (doc_dtype F32): rows = bf16 planes [hi | lo] (ld = 2 * plane) for the scan,
    // rows_f32 = the fp32 originals [capacity][plane] for the exact re-score; qf32 = staged fp32 queries
    int plane = 0;
    float* rows_f32 = nullptr;
    // two-stage search of the fp32-exact index ("f32_fast", d <= 1024): rows_hi = bf16(x) alone at a row pitch the
    // query-stationary kernels take ([capacity][hp]; converted lazily from rows_f32 up to hi_rows).  Stage 1 scans it
    // like a bf16 index and re-scores on the fp32 rows; the margin check, widened by the representation error
    // |x - bf16 x| |q| + |bf16 x| |q - bf16 q|, sends the queries it cannot certify to the three-segment scan.
    uint8_t* rows_hi = nullptr;
    int hp = 0;
    int64_t hi_rows = 0;
    double* dres2_dev = nullptr; // max_i |x_i - bf16 x_i|^2
    bool dres2_valid = false;
    int opt_f32_fast = 1;        // 0 off, 1 when the call may synchronise (host buffers / margin_check = 2), 2 always
    bool fast_f32 = false;       // launch_search: stage 1 in progress (index viewed as bf16 rows_hi)
    // "Optimistic" scan (calls that certify, i.e. may synchronise): pools of 16 / 32 candidates selected from the 16x16x32
    // kernel's 4 sub-lists of 6 instead of from true K'-entry lists on the 4-wave configuration.  What the pool may have
    // excluded is bounded all the same (merge_select: sub-lists' last entries), so the margin check decides per query;
    // flagged queries are re-scanned with true K' = 32 lists.
    bool optimistic = false;
    // "margin_check" = 3: device-output searches re-scan the queries they flag WITHOUT a synchronisation -- the flag list is
    // compacted on the device and the second scan, sized for all queries, lets the workgroups past the count leave
    const int* nq_dev = nullptr;         // launch_search: device-side query count of the re-scan in progress
    const int* first_nflag_dev = nullptr; // flagged count of the first scan of the last mode-3 search (margin stats)
    int plane_keep = 0;
    int fast_skip = 0;           // calls left to skip stage 1 for: set by a SYNCHRONISING call that flagged too many queries for the
                                 // optimistic scan to pay (its count is known when it returns; stream-ordered calls never set it,
                                 // so what a search does depends on the calls before it, not on when a device store lands)
    bool phi_valid = false;
    int call_metric = MIPS_METRIC_IP; // metric of the search in progress (index metric unless MIPS_FORCE_IP)
    bool phi_override = false; // phi was set from outside (global maximum of a sharded index): adds do not reset it
    double phi = 0.0;
    Buffer qbuf, qf32, part_s, part_i, stage, out_s, out_i, scalar, gthr, cand;
    // ring of HIP event pairs around the scan kernel (bench.py reads the average launch duration)
    static constexpr int kEvRing = 128;
    // tuning knobs (mips_index_set_param); 0 = automatic
    int opt_nsplit = 0;
    int opt_qgroups = 0;
    size_t err_off = 0; // word offset of the scan kernel's error flag inside gthr (0 = none this call)
    int opt_sub = 0;
    int opt_spin_limit = 0; // test-only: polls of the split barrier before a wave gives up (0 = 1 << 22, < 0 = flag forced)
    // sticky scan-error flag: one pinned, mapped host word.  The exact re-score sets it (system-scope store) when
    // the scan kernel of its call gave up on the split barrier; the host reads it without a device round trip.
    unsigned* sticky_host = nullptr;
    unsigned* sticky_dev = nullptr;
    char last_kernel[96] = ""; // instance mips_search dispatched last (mips_index_last_kernel)
    // margin check.  0 = off, 1 = flag and count on the device (never synchronises), 2 = certify:
    // synchronise, re-scan the flagged queries with the widest lists.  Host-output searches always certify (they
    // synchronise anyway) unless the check is off.
    int opt_margin = 1;
    Buffer mbnd, mflag, qbuf2, qf32b, tmp_s, tmp_i, ids, qhi, qerr2, keyk, qqv, hit_d, hit_i, hit_n, qnorm;
    int opt_resolve = 1; // flagged queries: 1 = exact brute-force resolution (resolve_kernels.hpp; 2 = its plain form, no MFMA pre-filter), 0 = re-scan with the widest lists
    int resolve_budget = 0; // "resolve_budget" > 0: flagged queries a search resolves at most (0 = RESOLVE_MAX); a search that flags
                            // more keeps its first results (counted unresolved) -- or, if its first scan was an optimistic one,
                            // goes through the stream-ordered re-scan with true K' = 32 lists
    double* xmax2_dev = nullptr; // max_i |x_i|^2 of the LOCAL rows, on the device (no host copy: never synchronises)
    bool xmax2_valid = false;
    unsigned* nflag_host = nullptr; // pinned: flagged-query count of the last certified search
    int64_t last_flagged = -1, last_rescanned = 0, last_unresolved = 0;
    bool last_fallback = false; // the last search enqueued the gated fall-back re-scan behind its exact pass
    int last_max_n = 0; // flagged queries the exact pass of the last search would resolve at most (statistics: over budget = none settled)
    int rescan_depth = 0;
    unsigned* last_nflag_dev = nullptr;
    // Split-tail searches (mips_search_split): the scan runs on one stream, select + exact re-score on another, so the
    // NEXT search's scan can start behind this one's.  Two scratch sets alternate; `alt_*` is the one not in use.
    Buffer alt_qbuf, alt_qf32, alt_gthr, alt_part_s, alt_part_i, alt_cand, alt_mbnd, alt_mflag;
    int cur_set = 0;
    hipEvent_t scan_done = nullptr;
    hipEvent_t tail_done[2] = {nullptr, nullptr};
    bool tail_pending[2] = {false, false};
    int opt_tiny = 1;              // 1 = searches of <= 16 queries over a small bf16 index take the one-launch kernel
    unsigned* tiny_words = nullptr; // [0] ticket (reset by the kernel's last workgroup), [1] flag counter
    int opt_variant = 0; // 0 = automatic, 1 = scan_kernel (128x128 tiles), 3 = scan_kernel_v3 (32x32x16), 4 = scan_kernel_v4 (16x16x32)
    hipEvent_t ev0[kEvRing] = {}, ev1[kEvRing] = {};
    int ev_count = 0; // pairs recorded since the last reset (saturates at kEvRing)
    bool timing_armed = false; // event pairs are recorded only inside a measurement window (mips_scan_timing reset):
                               // an event record costs ~5.7 us of stream time on this part, 11 us per search
    int ev_next = 0;
    // The scratch buffers are shared by every call on this index.  Calls on ONE stream are ordered by the
    // stream; a call arriving on another stream first waits for `busy`, recorded at the end of the last call.
    hipEvent_t busy = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
};

namespace {

// Rows the per-query buffers (staged queries, insert bounds, partial lists) are padded to: whole query tiles of every kernel that
// may take the search -- 256 (128 / 256-query tiles), and at bf16 row pitch 1024 also scan_kernel_k3's 192-query tiles
inline int64_t query_pad(const mips_index* ix, int64_t n) {
    const bool pitch_1024 = ix->esize == 2 && (ix->ld == 1024 || ix->hp == 1024);
    return round_up(n, pitch_1024 ? 768 : kQueryAlign);
}

// Orders the calls on one index across streams (see mips_index::busy).  Nothing is recorded per call (an event
// record costs ~5.7 us of stream time here): when a call arrives on ANOTHER stream than the previous one, the
// event is recorded on the previous stream at that moment and the new stream waits for it.  A stream handed to
// the library must therefore stay valid until the next call on the index (torch's pooled streams do).
struct StreamOrder {
    mips_index* ix;
    hipStream_t st;
    bool ok = true;
    StreamOrder(mips_index* ix_, hipStream_t st_) : ix(ix_), st(st_) {
        if (ix->has_last && ix->last_stream != st)
            ok = hipEventRecord(ix->busy, ix->last_stream) == hipSuccess && hipStreamWaitEvent(st, ix->busy, 0) == hipSuccess;
    }
    ~StreamOrder() {
        ix->last_stream = st;
        ix->has_last = true;
    }
};
#define ORDER_ON(ix, st)             \
    StreamOrder order_guard(ix, st); \
    if (!order_guard.ok) return fail(MIPS_E_HIP, "hipStreamWaitEvent failed")

// A scan kernel whose split barrier timed out poisons its call's output and raises the sticky flag; whoever looks
// first (the next call on the index, mips_index_check_error, a host-output search) reports and clears it.
int take_scan_error(mips_index* ix, const char* who) {
    if (ix->sticky_host == nullptr) return MIPS_OK;
    if (__atomic_load_n(ix->sticky_host, __ATOMIC_ACQUIRE) == 0u) return MIPS_OK;
    __atomic_store_n(ix->sticky_host, 0u, __ATOMIC_RELEASE);
    return fail(MIPS_E_SCAN_TIMEOUT,
                "%s: a scan kernel on this index gave up on its block barrier (spin bound reached); the results of "
                "that search were poisoned (idx %d, NaN scores) and must be discarded", who, MIPS_IDX_POISON);
}

void set_kernel_name(mips_index* ix, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ix->last_kernel, sizeof ix->last_kernel, fmt, ap);
    va_end(ap);
}

// exact = false: geometric growth for repeated adds; true: mips_index_reserve's exact reservation
int grow(mips_index* ix, int64_t need_rows, hipStream_t st, bool exact = false) {
    if (need_rows <= ix->capacity) return MIPS_OK;
    int64_t cap = exact ? need_rows : std::max<int64_t>(need_rows, ix->capacity + ix->capacity / 2);
    cap = round_up(cap, kRowAlign);
    uint8_t* fresh = nullptr;
    float* fresh32 = nullptr;
    uint8_t* fresh_hi = nullptr;
    const size_t row_bytes = (size_t)ix->ld * ix->esize;
    const size_t bytes = (size_t)cap * row_bytes;
    const size_t b32 = (size_t)cap * ix->plane * sizeof(float);
    hipError_t e = hipMalloc((void**)&fresh, bytes);
    if (e != hipSuccess) return fail(MIPS_E_NOMEM, "hipMalloc(%zu) for the index failed: %s", bytes, hipGetErrorString(e));
    if (ix->plane > 0) {
        e = hipMalloc((void**)&fresh32, b32);
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(MIPS_E_NOMEM, "hipMalloc(%zu) for the fp32 rows failed: %s", b32, hipGetErrorString(e));
        }
    }
    const size_t bhi = (size_t)cap * ix->hp * 2;
    if (ix->plane > 0 && ix->hp > 0) {
        e = hipMalloc((void**)&fresh_hi, bhi);
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            (void)hipFree(fresh32);
            return fail(MIPS_E_NOMEM, "hipMalloc(%zu) for the bf16 rows of the fp32 index failed: %s", bhi, hipGetErrorString(e));
        }
    }
    // copy the rows in use; rows past ntotal are read by the last (ragged) tile: keep them defined
    const size_t used = (size_t)ix->ntotal * row_bytes;
    const size_t u32 = (size_t)ix->ntotal * ix->plane * sizeof(float);
    if (used) e = hipMemcpyAsync(fresh, ix->rows, used, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(fresh + used, 0, bytes - used, st);
    if (e == hipSuccess && fresh32 && u32) e = hipMemcpyAsync(fresh32, ix->rows_f32, u32, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && fresh32) e = hipMemsetAsync((char*)fresh32 + u32, 0, b32 - u32, st);
    const size_t uhi = (size_t)ix->hi_rows * ix->hp * 2;
    if (e == hipSuccess && fresh_hi && uhi) e = hipMemcpyAsync(fresh_hi, ix->rows_hi, uhi, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && fresh_hi) e = hipMemsetAsync(fresh_hi + uhi, 0, bhi - uhi, st);
    if (e == hipSuccess && ix->rows) e = hipStreamSynchronize(st); // the old storage is freed below
    if (e != hipSuccess) {
        (void)hipFree(fresh);
        if (fresh32) (void)hipFree(fresh32);
        if (fresh_hi) (void)hipFree(fresh_hi);
        return fail(MIPS_E_HIP, "growing the index to %lld rows failed: %s", (long long)cap, hipGetErrorString(e));
    }
    if (ix->rows) (void)hipFree(ix->rows);
    if (ix->rows_f32) (void)hipFree(ix->rows_f32);
    if (ix->rows_hi) (void)hipFree(ix->rows_hi);
    ix->rows = fresh;
    ix->rows_f32 = fresh32;
    ix->rows_hi = fresh_hi;
    ix->capacity = cap;
    return MIPS_OK;
}

// convert [n][d] of src_dtype (host or device) into dst [n][ld] of the index element type on the device
// pad_rows / zero / zero_words: query staging only -- that many zero rows behind the last converted one and a
// word range to clear, both done by the launch that converts the last chunk (bf16 and fp8 storage)
// out_esize: bytes per OUTPUT element (0 = the index storage's; query staging passes ix->qsize)
int convert_into(mips_index* ix, const void* src, int64_t n, int src_dtype, int src_is_device, uint8_t* dst,
                 hipStream_t st, float* keep_f32 = nullptr, int64_t pad_rows = 0, uint32_t* zero = nullptr,
                 int64_t zero_words = 0, int out_esize = 0) {
    const int d = (int)ix->d, ld = ix->ld;
    if (out_esize == 0) out_esize = ix->esize;
    const size_t esz = src_dtype == MIPS_DTYPE_F32 ? 4 : src_dtype == MIPS_DTYPE_BF16 ? 2 : 1;
    const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(64u << 20) / (int64_t)(d * esz));
    for (int64_t r0 = 0; r0 < n; r0 += chunk_rows) {
        const int64_t nr = std::min(chunk_rows, n - r0);
        const void* s = (const char*)src + (size_t)r0 * d * esz;
        if (!src_is_device) {
            int rc = ix->stage.ensure((size_t)nr * d * esz);
            if (rc) return rc;
            // the staging buffer is reused by the next chunk: this copy is synchronous for pageable memory
            HIP_TRY(hipMemcpyAsync(ix->stage.p, s, (size_t)nr * d * esz, hipMemcpyHostToDevice, st));
            s = ix->stage.p;
        }
        uint8_t* out = dst + (size_t)r0 * ld * out_esize;
        if (ix->plane > 0) { // fp32-exact mode: bf16 planes [hi | lo] + the fp32 originals
            const int64_t items = nr * (ix->plane / 8);
            float* keep = keep_f32 ? keep_f32 + (size_t)r0 * ix->plane : nullptr;
            if (src_dtype == MIPS_DTYPE_F32)
                mips::split_rows_kernel<float><<<grid_for(items, 256), 256, 0, st>>>((const float*)s, nr, d, d, (uint16_t*)out, ix->plane, keep);
            else
                mips::split_rows_kernel<uint16_t><<<grid_for(items, 256), 256, 0, st>>>((const uint16_t*)s, nr, d, d, (uint16_t*)out, ix->plane, keep);
        } else {
            const bool last = r0 + nr == n;
            const int64_t n_out = nr + (last ? pad_rows : 0);
            uint32_t* z = last ? zero : nullptr;
            const int64_t zw = last ? zero_words : 0;
            if (out_esize == 2) {
                const int64_t items = n_out * (ld / 8);
                if (src_dtype == MIPS_DTYPE_FP8_E4M3) return fail(MIPS_E_INVALID, "e4m3 bytes cannot be staged as bf16 rows");
                if (src_dtype == MIPS_DTYPE_F32)
                    mips::convert_rows_kernel<float><<<grid_for(items, 256), 256, 0, st>>>((const float*)s, nr, d, (uint16_t*)out, ld, n_out, z, zw);
                else
                    mips::convert_rows_kernel<uint16_t><<<grid_for(items, 256), 256, 0, st>>>((const uint16_t*)s, nr, d, (uint16_t*)out, ld, n_out, z, zw);
            } else {
                const int64_t items = n_out * (ld / 16);
                if (src_dtype == MIPS_DTYPE_F32)
                    mips::convert_rows_f8_kernel<float><<<grid_for(items, 256), 256, 0, st>>>((const float*)s, nr, d, out, ld, n_out, z, zw);
                else if (src_dtype == MIPS_DTYPE_BF16)
                    mips::convert_rows_f8_kernel<uint16_t><<<grid_for(items, 256), 256, 0, st>>>((const uint16_t*)s, nr, d, out, ld, n_out, z, zw);
                else
                    mips::convert_rows_f8_kernel<uint8_t><<<grid_for(items, 256), 256, 0, st>>>((const uint8_t*)s, nr, d, out, ld, n_out, z, zw);
            }
        }
        HIP_TRY(hipGetLastError());
        if (!src_is_device) HIP_TRY(hipStreamSynchronize(st));
    }
    return MIPS_OK;
}

// for_query: queries of an e4m3-documents / bf16-queries index are float32 or bf16 (raw e4m3 bytes are rows only)
bool src_dtype_ok(const mips_index* ix, int t, bool for_query = false) {
    return t == MIPS_DTYPE_F32 || t == MIPS_DTYPE_BF16 ||
           (t == MIPS_DTYPE_FP8_E4M3 && ix->esize == 1 && ix->plane == 0 && !(for_query && ix->mixed));
}

int compute_phi(mips_index* ix, hipStream_t st) {
    if (ix->phi_valid) return MIPS_OK;
    int rc = ix->scalar.ensure(16);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ix->scalar.p, 0, 8, st));
    if (ix->ntotal > 0) {
        const int grid = (int)((ix->ntotal + 255) / 256);
        if (ix->plane > 0)
            mips::row_sumsq_max_kernel<mips::ElemF32><<<grid, 256, 0, st>>>(ix->rows_f32, ix->ntotal, ix->plane,
                                                                             (unsigned long long*)ix->scalar.p);
        else if (ix->esize == 2)
            mips::row_sumsq_max_kernel<mips::ElemBF16><<<grid, 256, 0, st>>>((const uint16_t*)ix->rows, ix->ntotal, ix->ld,
                                                                              (unsigned long long*)ix->scalar.p);
        else
            mips::row_sumsq_max_kernel<mips::ElemF8><<<grid, 256, 0, st>>>(ix->rows, ix->ntotal, ix->ld,
                                                                            (unsigned long long*)ix->scalar.p);
        HIP_TRY(hipGetLastError());
    }
    unsigned long long bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, ix->scalar.p, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::memcpy(&ix->phi, &bits, 8);
    ix->phi_valid = true;
    return MIPS_OK;
}

// max_i |x_i|^2 of the local rows into a device scalar, stream-ordered, no host copy (the margin check's error bound)
int ensure_xmax2(mips_index* ix, hipStream_t st) {
    if (ix->xmax2_valid) return MIPS_OK;
    if (!ix->xmax2_dev) HIP_TRY(hipMalloc((void**)&ix->xmax2_dev, 8));
    HIP_TRY(hipMemsetAsync(ix->xmax2_dev, 0, 8, st));
    if (ix->ntotal > 0) {
        const int grid = (int)((ix->ntotal + 255) / 256);
        unsigned long long* out = (unsigned long long*)ix->xmax2_dev;
        if (ix->plane > 0) mips::row_sumsq_max_kernel<mips::ElemF32><<<grid, 256, 0, st>>>(ix->rows_f32, ix->ntotal, ix->plane, out);
        else if (ix->esize == 2) mips::row_sumsq_max_kernel<mips::ElemBF16><<<grid, 256, 0, st>>>((const uint16_t*)ix->rows, ix->ntotal, ix->ld, out);
        else mips::row_sumsq_max_kernel<mips::ElemF8><<<grid, 256, 0, st>>>(ix->rows, ix->ntotal, ix->ld, out);
        HIP_TRY(hipGetLastError());
    }
    ix->xmax2_valid = true;
    return MIPS_OK;
}

// two-stage fp32-exact search: bf16 rows of the rows added since the last call, and the residual bound
int ensure_hi(mips_index* ix, hipStream_t st) {
    if (ix->hi_rows < ix->ntotal) {
        const int64_t nr = ix->ntotal - ix->hi_rows;
        const int64_t items = nr * (ix->hp / 8);
        mips::convert_rows_kernel<float><<<grid_for(items, 256), 256, 0, st>>>(ix->rows_f32 + (size_t)ix->hi_rows * ix->plane, nr, ix->plane,
                                                                               (uint16_t*)ix->rows_hi + (size_t)ix->hi_rows * ix->hp, ix->hp);
        HIP_TRY(hipGetLastError());
        ix->hi_rows = ix->ntotal;
    }
    if (!ix->dres2_valid) {
        if (!ix->dres2_dev) HIP_TRY(hipMalloc((void**)&ix->dres2_dev, 8));
        HIP_TRY(hipMemsetAsync(ix->dres2_dev, 0, 8, st));
        if (ix->ntotal > 0) {
            mips::row_resid_sumsq_max_kernel<<<(int)((ix->ntotal + 255) / 256), 256, 0, st>>>(ix->rows_f32, ix->ntotal, ix->plane,
                                                                                             (unsigned long long*)ix->dres2_dev);
            HIP_TRY(hipGetLastError());
        }
        ix->dres2_valid = true;
    }
    return MIPS_OK;
}

void swap_scratch_sets(mips_index* ix) {
    std::swap(ix->qbuf, ix->alt_qbuf);
    std::swap(ix->qf32, ix->alt_qf32);
    std::swap(ix->gthr, ix->alt_gthr);
    std::swap(ix->part_s, ix->alt_part_s);
    std::swap(ix->part_i, ix->alt_part_i);
    std::swap(ix->cand, ix->alt_cand);
    std::swap(ix->mbnd, ix->alt_mbnd);
    std::swap(ix->mflag, ix->alt_mflag);
    ix->cur_set ^= 1;
}

// ---- The shipped instances of the query-stationary bf16 scans, ONE ROW EACH: what selects an instance (row pitch, document cache
// policy, published rank), what it needs (waves -> threads, ring stages -> LDS bytes), its entry point and the name rocprofv3
// prints.  launch_search looks the row up; a new pitch / list length / policy is one more row here.
struct ScanInstance {
    int ld;            // row pitch in elements
    bool nt;           // non-temporal document DMA (searches of one query tile: every block has a single reader)
    int pub;           // scan_kernel_v4: rank every sub-list publishes (1: pools of 8, 4: pools of 32); 0 for scan_kernel_v3
    int waves, stages; // per workgroup / of the LDS ring
    const void* fn;    // __global__ entry taking ScanArgs by value
    const char* name;  // printf format; scan_kernel_v3 rows take K' as their one %d
};
inline int scan_instance_lds(const ScanInstance& e) { // ring + class-word copies (1 KiB per wave) + dump area + arrival counter
    return e.stages * mips::V3_DB * e.ld * 2 + e.waves * 1024 + 1024 + 16;
}
inline const ScanInstance* find_instance(const ScanInstance* t, int n, int ld, bool nt, int pub) {
    for (int i = 0; i < n; ++i)
        if (t[i].ld == ld && t[i].nt == nt && t[i].pub == pub) return &t[i];
    return nullptr;
}
// scan_kernel_v4: 16x16x32 MFMA, 8 waves x 32 queries, 3-stage ring, 4 sub-lists of 6 per (query, split)
#define MIPS_V4_ROW(KS, NT, PUB)                                                                                   \
    {KS * 32, NT, PUB, 8, 3, (const void*)mips::scan_kernel_v4<6, KS, 2, 0, NT, PUB>, "mips::scan_kernel_v4<6, " #KS ", 2, 0, " #NT ", " #PUB ">"}
#define MIPS_V4_PITCH(KS) MIPS_V4_ROW(KS, false, 1), MIPS_V4_ROW(KS, true, 1), MIPS_V4_ROW(KS, false, 4), MIPS_V4_ROW(KS, true, 4)
inline const ScanInstance* v4_instances(int* n) {
    static const ScanInstance t[] = {MIPS_V4_PITCH(12), MIPS_V4_PITCH(16), MIPS_V4_PITCH(20), MIPS_V4_PITCH(24)};
    *n = (int)(sizeof t / sizeof t[0]);
    return t;
}
// scan_kernel_v3: 32x32x16 MFMA, true K'-entry lists.  K' <= 10 at pitch <= 768: 8 waves (two per SIMD), 3-stage ring; K' = 16 /
// 32 there: 4 waves (one per SIMD, 512 registers), 3-stage ring; pitch 1024 (256 fragment registers): 4 waves, 2 stages of 64 KiB
#define MIPS_V3_ROW8(KS16, NT)                                                                                     \
    {KS16 * 16, NT, 0, 8, 3, (const void*)mips::scan_kernel_v3<KL, KS16, 1, 2, true, 0, 2, 8, 3, true, NT>,        \
     "mips::scan_kernel_v3<%d, " #KS16 ", 1, 2, true, 0, 2, 8, 3, true, " #NT ", 8>"}
#define MIPS_V3_ROW4(KS16, NT)                                                                                     \
    {KS16 * 16, NT, 0, 4, 3, (const void*)mips::scan_kernel_v3<KL, KS16, 1, 4, true, 0, 2, 4, 3, true, NT>,        \
     "mips::scan_kernel_v3<%d, " #KS16 ", 1, 4, true, 0, 2, 4, 3, true, " #NT ", 8>"}
#define MIPS_V3_ROW1024(NT)                                                                                        \
    {1024, NT, 0, 4, 2, (const void*)mips::scan_kernel_v3<KL, 64, 1, 4, false, 0, 2, 4, 2, true, NT>,              \
     "mips::scan_kernel_v3<%d, 64, 1, 4, false, 0, 2, 4, 2, true, " #NT ", 8>"}
template <int KL>
const ScanInstance* v3_instances(int* n) {
    if constexpr (KL <= 10) {
        static const ScanInstance t[] = {MIPS_V3_ROW8(8, false),  MIPS_V3_ROW8(8, true),  MIPS_V3_ROW8(16, false), MIPS_V3_ROW8(16, true),
                                         MIPS_V3_ROW8(24, false), MIPS_V3_ROW8(24, true), MIPS_V3_ROW8(32, false), MIPS_V3_ROW8(32, true),
                                         MIPS_V3_ROW8(40, false), MIPS_V3_ROW8(40, true), MIPS_V3_ROW8(48, false), MIPS_V3_ROW8(48, true),
                                         MIPS_V3_ROW1024(false),  MIPS_V3_ROW1024(true)};
        *n = (int)(sizeof t / sizeof t[0]);
        return t;
    } else {
        static const ScanInstance t[] = {MIPS_V3_ROW4(16, false), MIPS_V3_ROW4(16, true), MIPS_V3_ROW4(32, false), MIPS_V3_ROW4(32, true),
                                         MIPS_V3_ROW4(48, false), MIPS_V3_ROW4(48, true), MIPS_V3_ROW1024(false), MIPS_V3_ROW1024(true)};
        *n = (int)(sizeof t / sizeof t[0]);
        return t;
    }
}

// scan_kernel_e8 (e4m3 documents x bf16 queries): instance by row pitch, configuration, document cache policy.
// scan_kernel_e8 configurations: tiles of 16 / 32 / 64 queries (ncb = 1 / 2 / 4).  K parts (kw): 8 for the 16-query tile, 4 beyond
// (half the partial sums through LDS: two slot buffers -- one barrier per block -- then fit for every tile but 64 queries at
// pitch 1024); ring depth 4 where 160 KiB allow it.  old_rules: the first version's configurations (A/B library only).
struct E8Config {
    int ncb, stages, kw;
    bool pipe;
};
inline E8Config e8_config(int ld, int64_t nq, bool old_rules) {
    E8Config c;
    if (old_rules) {
        c.ncb = nq <= 16 ? 1 : (ld == 1024 || nq <= 32) ? 2 : 4;
        c.kw = 8;
        c.pipe = c.ncb == 1 || (c.ncb == 2 && ld <= 768);
        c.stages = c.ncb == 1 && ld <= 768 ? 4 : 3;
    } else {
        c.ncb = nq <= 16 ? 1 : nq <= 32 ? 2 : 4;
        c.kw = c.ncb == 1 ? 8 : 4;
        c.pipe = c.ncb <= 2 || ld <= 768;
        c.stages = c.ncb <= 2 && ld <= 768 ? 4 : 3;
    }
    return c;
}
template <int PUB>
int launch_e8(mips_index* ix, const mips::ScanArgsE8& fa, int grid, const E8Config& c, bool nt, hipStream_t st, int slot) {
    // ring + slot buffer(s) + class words + dump + counters
    const int lds = c.stages * mips::V3_DB * ix->ld + (c.pipe ? 2 : 1) * c.kw * (2 * c.ncb) * 1024 + 2048 + 1024 + 64;
    auto go = [&](auto kern) -> int {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
        kern<<<grid, 512, lds, st>>>(fa);
        return MIPS_OK;
    };
#define MIPS_E8_ROW(LDB, NCB, STG, PIPE, KW)                                               \
    if (c.ncb == NCB && c.stages == STG && c.pipe == PIPE && c.kw == KW)                   \
        return nt ? go(mips::scan_kernel_e8<6, LDB, NCB, STG, true, PUB, PIPE, KW>) : go(mips::scan_kernel_e8<6, LDB, NCB, STG, false, PUB, PIPE, KW>)
#ifdef MIPS_EXPERIMENTAL
#define MIPS_E8_OLD(LDB)                                                        \
        MIPS_E8_ROW(LDB, 2, 3, (LDB <= 768), 8);                                \
        if constexpr (LDB <= 768) { MIPS_E8_ROW(LDB, 4, 3, false, 8); }
#else
#define MIPS_E8_OLD(LDB)
#endif
#define MIPS_E8_PITCH(LDB)                                                      \
    if (ix->ld == LDB) {                                                        \
        MIPS_E8_ROW(LDB, 1, (LDB <= 768 ? 4 : 3), true, 8);                     \
        MIPS_E8_ROW(LDB, 2, (LDB <= 768 ? 4 : 3), true, 4);                     \
        MIPS_E8_ROW(LDB, 4, 3, (LDB <= 768), 4);                                \
        MIPS_E8_OLD(LDB)                                                        \
    }
    MIPS_E8_PITCH(256)
    MIPS_E8_PITCH(512)
    MIPS_E8_PITCH(768)
    MIPS_E8_PITCH(1024)
#undef MIPS_E8_PITCH
#undef MIPS_E8_OLD
#undef MIPS_E8_ROW
    return fail(MIPS_E_UNSUPPORTED, "e4m3-documents index: no scan instance for row pitch %d with %d query blocks", ix->ld, c.ncb);
}

// tail_st: stream of the select + exact re-score launches (nullptr or == st: the scan's own stream)
template <int KL>
int launch_search(mips_index* ix, int64_t nq, int k, float* d_out_s, int64_t* d_out_i, int64_t* d_out_packed,
                  int64_t idx_offset, hipStream_t st, hipStream_t tail_st = nullptr, bool split = false) {
    // variant 3 (query-stationary, LDS-DMA): the whole K of a wave's 32 queries lives in its VGPRs, so it
    // exists for a few row lengths only: 256 / 512 / 768 (8 waves, 2 per SIMD) and 1024 (4 waves, 1 per SIMD)
    int variant = ix->opt_variant;
    // scan_kernel_v4 (16x16x32 MFMA shape, 4 sub-lists of 6): row pitch 384 .. 768 (at 256 the shorter chain no
    // longer pays: 2.03 vs 2.00 ms), k <= 5, bf16 storage.  It is the default
    // there when more than one query tile shares the document stream (the MFMA-bound regime, where the shape's
    // higher clock pays: 4.54 vs 4.78 ms at BASELINE config 2); single-tile searches are HBM-bound and keep
    // scan_kernel_v3's non-temporal document DMA.  "variant" = 3 / 4 forces one of the two.
    const bool v4_opt = ix->optimistic && ix->rescan_depth == 0 && (KL == 16 || KL == 32) && ix->opt_variant == 0 && ix->opt_sub == 0 &&
                        ix->ld % 128 == 0 && ix->ld >= 384 && ix->ld <= 768 && ix->esize == 2 && ix->plane == 0;
    const bool v4_shape = ix->ld % 128 == 0 && ix->ld >= 384 && ix->ld <= 768 && (KL == 8 || v4_opt) && ix->esize == 2 && ix->plane == 0;
    // scan_kernel_v5 (64 stationary queries per wave, one wave per SIMD): row pitches whose 64-k slabs divide evenly
    // among 4 waves
    const bool v5_shape = (ix->ld == 768 || ix->ld == 512) && KL == 8 && ix->esize == 2 && ix->plane == 0;
#ifdef MIPS_EXPERIMENTAL
    const bool want_v5 = variant == 5 && v5_shape;
#else
    const bool want_v5 = false; // ("variant" = 5 / 6 select kernels of the A/B library only; the shipped library ignores them)
    (void)v5_shape;
#endif
    const bool v4_forced = variant == 4 && v4_shape;
    const bool v4_auto = variant == 0 && ix->opt_sub == 0 && v4_shape;
    if (variant != 1 && variant != 3) variant = 3; // (4 / 5 were decided above; the rest of the function only knows 1 and 3)
    const bool v3_dim = (ix->ld % 128 == 0 && ix->ld <= 768) || ix->ld == 1024;
    const bool v3_long = ix->ld == 256 || ix->ld == 512 || ix->ld == 768 || ix->ld == 1024; // pitches with K' = 16 / 32 instances
    constexpr bool kl_short = KL <= 10; // K' = 8 / 10 lists fit the 8-wave (two per SIMD) configuration
    const bool f8 = ix->esize == 1; // e4m3 index: scan_kernel_f8 only (row lengths 256..1024, K' <= 16)
    const bool f32x = ix->plane > 0; // fp32-exact mode: generic kernel over the [hi | lo] planes, three k segments
    // e4m3 documents x bf16 queries (MIPS_DTYPE_FP8_E4M3_DOCS): scan_kernel_e8, tiles of 32 queries (up to 32 queries, and at
    // row pitch 1024) or 64; pools of 8 / 10 / 16 / 32 out of 8 sub-lists of 6 per (query, split), the class words vouching for
    // 8 PUB documents
    const bool e8 = ix->mixed;
#ifdef MIPS_EXPERIMENTAL
    const E8Config e8c = e8_config(ix->ld, nq, ix->opt_sub == 71); // ("sub" = 71: the first version's configurations)
#else
    const E8Config e8c = e8_config(ix->ld, nq, false);
#endif
    const int e8_ncb = e8c.ncb;
    if (e8) {
        if (ix->ld % 256 != 0 || ix->ld > 1024) return fail(MIPS_E_UNSUPPORTED, "e4m3-documents index: d must pad to 256/512/768/1024");
        variant = 3;
    } else if (f8) {
        if (ix->ld % 256 != 0 || ix->ld > 1024 || KL > 16) return fail(MIPS_E_UNSUPPORTED, "fp8 index: d must pad to 256/512/768/1024 and k <= 13");
        variant = 3;
    } else if (f32x || !v3_dim || (!kl_short && !v3_long && !v4_opt)) {
        variant = 1; // no query-stationary configuration: generic tiles
    }
    // K' = 8 / 10 (k <= 7) at d <= 768: 8 waves, two per SIMD (256 registers each, no spill up to K' = 10).
    // Longer lists or d = 1024 do not fit next to the fragments there: 4 waves, one per SIMD, 512 registers,
    // 128 queries per workgroup.
    const int v3_waves = (!f8 && !v4_opt && (ix->ld == 1024 || !kl_short)) ? 4 : 8;
    // fp8: scan_kernel_f8x (16x16x128 MFMA shape, 64-document blocks, 4 sub-lists of 6) for k <= 5 and row pitches
    // up to 768 bytes; scan_kernel_f8 (32x32x64, 32-document blocks) otherwise or when "variant" = 3 asks for it
    const bool want_f8x = f8 && !e8 && KL == 8 && ix->ld <= 768 && ix->opt_variant != 3;
    // scan_kernel_ks (K split over a wave pair, two waves per SIMD): row pitch 1024, k <= 5.  Selectable ("variant" =
    // 6), not the default: measured 30.6 vs 31.3 ms at 2^22 x 1024 against the one-wave-per-SIMD scan_kernel_v3
    // configuration (profiles/r2_pitch1024) -- both sit on the L2 -> LDS fill of 128 stationary queries per CU
    // (round 3, later: pools of 16 / 32 out of scan_kernel_k3's sub-lists, every sub-list vouching for its 2nd / 4th best -- the
    // "optimistic" pools of scan_kernel_v4 at this pitch: first stage of the fp32-exact search at d in (768, 1024], bf16 searches
    // with 8 <= k <= 29, and k <= 5 on large indexes, where the MFMA error bound at K = 1024 reaches the 8th best score of one
    // query in a few thousand and a flagged query costs a pass over the index)
    const bool k3_opt = ix->optimistic && ix->rescan_depth == 0 && (KL == 16 || KL == 32) && ix->opt_variant == 0 && ix->opt_sub == 0 &&
                        ix->ld == 1024 && ix->esize == 2 && ix->plane == 0 && !ix->mixed && nq > 256;
    const bool ks_shape = ix->ld == 1024 && (KL == 8 || k3_opt) && ix->esize == 2 && ix->plane == 0;
#ifdef MIPS_EXPERIMENTAL
    const bool want_ks = ks_shape && ix->opt_variant == 6;
#else
    const bool want_ks = false;
#endif
    // scan_kernel_k3 (round 3): the wave pairs of scan_kernel_ks with 48 queries each -- 192 stationary queries per CU, a third
    // less L2 -> LDS fill per flop, which is what bounds pitch 1024.  Default there once several 192-query tiles share the
    // document stream (the MFMA-bound regime); smaller searches keep the 128-query configuration ("variant" = 7 / 3 force one)
    const bool want_k3 = k3_opt || (ks_shape && (ix->opt_sub == 0 || (ix->opt_sub >= 61 && ix->opt_sub <= 68)) && (ix->opt_variant == 7 || (ix->opt_variant == 0 && nq > 256)));
    constexpr int K3_KLL = 4; // entries per sub-list (the third accumulator set is paid for with shorter lists)
    // (a variant on 16-document stages -- 4-stage ring, three blocks in flight, one barrier per 16 documents -- was built and
    // measured 18 % SLOWER, 34.8 vs 29.5 ms at 2^22 x 1024: profiles/r3_pitch1024/README.md; what parks the waves is the barrier
    // itself, not the landing of the pieces)
    const int tm = variant == 1 ? mips::TM : want_f8x ? mips::F8X_DB : mips::V3_DB; // documents per scheduling unit ("tile")
    const int tn = variant == 1 ? mips::TN : e8 ? 16 * e8_ncb : want_k3 ? 192 : v3_waves * 32; // queries per workgroup
    const int wg_target = variant == 1 ? 512 : 256;                   // resident workgroups on 256 CUs
    const int64_t nq_pad = query_pad(ix, nq);
    const int nqt = (int)((nq + tn - 1) / tn);
    // One query tile (round 2): with non-temporal document DMA the 16x16x32 kernel ties scan_kernel_v3 in the HBM-bound
    // regime on large indexes (3.80 vs 3.82 ms at Q = 64 on 2^24 rows), loses 3-8 % on short streams at Q = 8 (0.315 vs
    // 0.304 ms at 2^20 rows, 0.091 vs 0.084 at 2^17) and wins once several waves multiply (3.89 vs 4.15 ms at Q = 128,
    // 5.57 vs 5.98 at Q = 256; profiles/r2_final/ab_single_tile.md): scan_kernel_v3 up to 64 queries, v4 beyond
    const bool want_v4 = !want_v5 && (v4_forced || v4_opt || (v4_auto && (nqt > 1 || nq > 64)));
    const int lists = (want_ks || want_k3 || e8) ? 8 : (want_v4 || want_v5 || want_f8x) ? 4 : 2; // running lists per (query, split)
    const int ntiles = (int)((ix->ntotal + tm - 1) / tm);
    // Index splits (a multiple of 8: one XCD group each).  The grid nqt x nsplit should come in whole
    // "rounds" of wg_target resident workgroups: among the multiples of 8 up to 64 take the one whose last
    // round is fullest (ties: fewer splits = longer streams, fewer lists to merge).
    int nsplit;
    if (ix->opt_nsplit > 0) {
        nsplit = (int)round_up(ix->opt_nsplit, 8);
    } else {
        nsplit = (int)round_up(std::max(1, (wg_target + nqt - 1) / nqt), 8);
        double best = -1.0;
        for (int cand = 8; cand <= 64 && (int64_t)nqt * cand <= 16 * (int64_t)wg_target; cand += 8) {
            const int64_t wgs = (int64_t)nqt * cand;
            if (wgs < wg_target && cand < nsplit) continue; // never leave CUs idle on purpose
            const double eff = (double)wgs / (double)(((wgs + wg_target - 1) / wg_target) * wg_target);
            if (eff > best + 0.02) {
                best = eff;
                nsplit = cand;
            }
        }
    }
    nsplit = (int)std::min<int64_t>(nsplit, round_up(ntiles, 8));
    const int tps = (ntiles + nsplit - 1) / nsplit;
    // query-tile groups per XCD.  Variant 1 re-reads its query tiles from L2 for every document tile:
    // keep an XCD's query working set at <= 8 tiles (1.5 MiB of its 4 MiB L2).  Variant 3 holds the
    // queries in registers: give every XCD as many query tiles of ONE split as possible instead, so a
    // document block is fetched from HBM once and served to the other tiles from that XCD's L2.
    int qgroups = ix->opt_qgroups;
    if (qgroups != 1 && qgroups != 2 && qgroups != 4 && qgroups != 8) {
        if (variant == 1) qgroups = nqt <= 8 ? 1 : nqt <= 16 ? 2 : nqt <= 32 ? 4 : 8;
        else qgroups = nqt <= 32 ? 1 : nqt <= 64 ? 2 : nqt <= 128 ? 4 : 8;
    }
    const int qt_per_group = (nqt + qgroups - 1) / qgroups;

    // v4 keeps 4 sub-lists per (query, split); each needs k (<= 5) + 1 entries only, the re-score pool is
    // still the K' = 8 best of their union
    constexpr int V4_KLL = 6;
#ifdef MIPS_EXPERIMENTAL
    const bool short_lists = !want_v4 && KL == 8 && variant == 3 && !f8 && ix->ld == 768 && (ix->opt_sub == 10 || ix->opt_sub == 11);
#else
    const bool short_lists = false;
#endif
    int list_len = want_k3 ? K3_KLL : (want_v4 || want_v5 || want_ks || want_f8x || short_lists || e8) ? V4_KLL : KL; // entries per running list
#ifdef MIPS_EXPERIMENTAL
    // shorter sub-lists buy registers for a deeper A-fragment prefetch in scan_kernel_v4 ("sub" = 55 / 56 / 57: lists of 5 at depth 2,
    // lists of 5 at depth 3, lists of 4 at depth 3; same results -- what a shorter list drops the margin check prices)
    if (want_v4 && ix->ld == 768 && (ix->opt_sub == 55 || ix->opt_sub == 56)) list_len = 5;
    if (want_v4 && ix->ld == 768 && ix->opt_sub == 57) list_len = 4;
#endif
    const size_t ncand = (size_t)nsplit * lists * list_len;
    int rc = ix->part_s.ensure((size_t)nq_pad * ncand * sizeof(float));
    if (rc) return rc;
    rc = ix->part_i.ensure((size_t)nq_pad * ncand * sizeof(int));
    if (rc) return rc;

    mips::ScanArgs a;
    a.docs = (const uint16_t*)ix->rows;
    a.qbuf = (const uint16_t*)ix->qbuf.p;
    a.ntotal = ix->ntotal;
    a.ld = ix->ld;
    a.ksteps = f32x ? 3 * ix->plane / mips::BK : ix->ld / mips::BK;
    a.plane = ix->plane;
    a.ntiles = ntiles;
    a.tiles_per_split = tps;
    a.nsplit = nsplit;
    a.nqt = nqt;
    a.nq = (int)nq;
    a.nq_dev = ix->nq_dev;
    a.qgroups = qgroups;
    a.qt_per_group = qt_per_group;
    a.splits_per_group = nsplit / (8 / qgroups);
    a.part_s = (float*)ix->part_s.p;
    a.part_i = (int*)ix->part_i.p;
    a.gthr = nullptr;
    a.err = nullptr;
    a.spin_limit = ix->opt_spin_limit != 0 ? ix->opt_spin_limit : (1 << 22);
    ix->err_off = 0;
    if (variant == 3) {
        // shared insert bounds: 8 class words per query (2 lane-half words in the older layouts) + error word
        const size_t thr_words = (size_t)nq_pad * 8;
        // (allocated and cleared by mips_search together with the query staging)
        a.gthr = (unsigned*)ix->gthr.p;
        a.err = a.gthr + thr_words;
        ix->err_off = thr_words;
    }

    const int grid = qt_per_group * qgroups * nsplit;
    const int slot = ix->ev_next;
    // launch one row of the instance tables above
    auto launch_row = [&](const ScanInstance* e, int name_arg) -> int {
        if (e == nullptr) return fail(MIPS_E_UNSUPPORTED, "no scan-kernel instance for row pitch %d, K' = %d", ix->ld, KL);
        const int lds = scan_instance_lds(*e);
        HIP_TRY(hipFuncSetAttribute(e->fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
        void* kargs[] = {(void*)&a};
        HIP_TRY(hipLaunchKernel(e->fn, dim3((unsigned)grid), dim3((unsigned)(e->waves * 64)), kargs, (size_t)lds, st));
        set_kernel_name(ix, e->name, name_arg);
        return MIPS_OK;
    };
    if (e8) {
        mips::ScanArgsE8 fa;
        fa.docs = ix->rows;
        fa.c = a;
        constexpr int PUB = KL <= 8 ? 1 : KL <= 16 ? 2 : 4;
        const bool nt = nqt == 1;
        int rc2 = launch_e8<PUB>(ix, fa, grid, e8c, nt, st, slot);
        if (rc2) return rc2;
        set_kernel_name(ix, "mips::scan_kernel_e8<6, %d, %d, %d, %s, %d, %s, %d>", ix->ld, e8c.ncb, e8c.stages, nt ? "true" : "false", PUB,
                        e8c.pipe ? "true" : "false", e8c.kw);
    } else if (want_k3) {
        if constexpr (KL == 8 || KL == 16 || KL == 32) {
            constexpr int K3_PUB = KL / 8; // pool of 8 PUB candidates: every sub-list vouches for its PUB-th best
            const int lds = 2 * mips::V3_DB * ix->ld * 2 + 4 * 1536 + 8 * 3072 + 64 + 256; // ring + the pairs' class-word copies + exchange slots + counters
            auto gok3 = [&](auto kern) -> int {
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
                kern<<<grid, 512, lds, st>>>(a);
                return MIPS_OK;
            };
            int rc3;
#ifdef MIPS_EXPERIMENTAL
            // diagnostic builds (wrong results by design; profiles/r3_pitch1024): no epilogue / no document DMA / no pair hand-shake
            if (ix->opt_sub == 61) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 1>);
            else if (ix->opt_sub == 62) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 2>);
            else if (ix->opt_sub == 63) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 3>);
            else if (ix->opt_sub == 64) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 4>); // other schedules of the DMA pieces (results unchanged)
            else if (ix->opt_sub == 65) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 5>);
            else if (ix->opt_sub == 66) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 6>);
            else if (ix->opt_sub == 67) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 7>); // + L2 prefetch three blocks ahead
            else if (ix->opt_sub == 68) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 8>); // ... six blocks ahead
            else
#endif
            rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 0, K3_PUB>);
            if (rc3) return rc3;
            set_kernel_name(ix, "mips::scan_kernel_k3<%d, 32, 2, 0, %d>", K3_KLL, K3_PUB);
        }
#ifdef MIPS_EXPERIMENTAL
    } else if (want_ks) {
        if constexpr (KL == 8) {
            const int lds = 2 * mips::V3_DB * ix->ld * 2 + 8 * 1024 + 8 * 2048 + 64; // ring + class-word copies + exchange slots + counters
            HIP_TRY(hipFuncSetAttribute((const void*)mips::scan_kernel_ks<V4_KLL, 32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
            mips::scan_kernel_ks<V4_KLL, 32, 2><<<grid, 512, lds, st>>>(a);
            set_kernel_name(ix, "mips::scan_kernel_ks<%d, 32, 2, 0>", V4_KLL);
        }
    } else if (want_v5) {
        if constexpr (KL == 8) {
            const int lds = 3 * mips::V3_DB * ix->ld * 2 + 4 * 2048 + 1024 + 16; // ring + threshold words + dump area + arrival counter
            auto go5 = [&](auto kern) -> int {
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
                kern<<<grid, 256, lds, st>>>(a);
                return MIPS_OK;
            };
            int rc2;
            if (ix->ld == 768 && ix->opt_sub == 8) rc2 = go5(mips::scan_kernel_v5<V4_KLL, 24, 2, 1>); // timing only: no epilogue
            else if (ix->ld == 768 && ix->opt_sub == 21) rc2 = go5(mips::scan_kernel_v5<V4_KLL, 24, 3>); // prefetch depth 3
            else if (ix->ld == 768 && ix->opt_sub == 22) rc2 = go5(mips::scan_kernel_v5<V4_KLL, 24, 4>); // prefetch depth 4
            else if (ix->ld == 768) rc2 = go5(mips::scan_kernel_v5<V4_KLL, 24>);
            else rc2 = go5(mips::scan_kernel_v5<V4_KLL, 16>);
            if (rc2) return rc2;
            set_kernel_name(ix, "mips::scan_kernel_v5<%d, %d, 2, 0>", V4_KLL, ix->ld / 32);
        }
#endif
    } else if (want_v4) {
        if constexpr (KL == 8 || KL == 16 || KL == 32) {
#ifdef MIPS_EXPERIMENTAL
            const int lds = 3 * mips::V3_DB * ix->ld * 2 + 8 * 1024 + 1024 + 16; // ring + threshold words + dump area + arrival counter
            auto go4 = [&](auto kern) -> int {
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
                kern<<<grid, 512, lds, st>>>(a);
                return MIPS_OK;
            };
#endif
            int rc2;
            bool named = false;
            const bool nt = nqt == 1 && ix->opt_sub != 30; // one query tile: every document block has a single reader
#ifdef MIPS_EXPERIMENTAL
            if (ix->ld == 768 && ix->opt_sub == 8) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 1>); // timing only: no epilogue
            else if (ix->ld == 768 && ix->opt_sub == 43) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 3>); // static priority for waves 4 .. 7
            else if (ix->ld == 768 && ix->opt_sub == 44) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 4>); // s_nop arrival poll
            else if (ix->ld == 768 && ix->opt_sub == 45) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 5>); // both
            else if (ix->ld == 768 && ix->opt_sub == 46) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 6>); // timing only: no document DMA
            else if (ix->ld == 768 && ix->opt_sub == 47) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 7>); // timing only: no block barrier wait
            else if (ix->ld == 768 && ix->opt_sub == 48) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 8>); // timing only: no DMA, no epilogue
            else if (ix->ld == 768 && ix->opt_sub == 55) rc2 = go4(mips::scan_kernel_v4<5, 24, 2, 0>);
            else if (ix->ld == 768 && ix->opt_sub == 56) rc2 = go4(mips::scan_kernel_v4<5, 24, 3, 0>);
            else if (ix->ld == 768 && ix->opt_sub == 57) rc2 = go4(mips::scan_kernel_v4<4, 24, 3, 0>);
            else if (ix->ld == 768 && ix->opt_sub == 49) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 9>); // SIMD partners issue their DMA pieces half a period apart
            else if (ix->ld == 768 && ix->opt_sub == 50) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 10>); // second wave of a SIMD starts 64 cycles late
            else if (ix->ld == 768 && ix->opt_sub == 52) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 11>); // ... 128
            else if (ix->ld == 768 && ix->opt_sub == 53) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 12>); // ... 192
            else
#endif
            {   // pools of 8 (PUB 1) or, optimistic, of 16 / 32: every sub-list vouches for its 4th best (8 x 4 = 32 documents)
                int nrow = 0;
                const ScanInstance* rows = v4_instances(&nrow);
                rc2 = launch_row(find_instance(rows, nrow, ix->ld, nt, v4_opt ? 4 : 1), 0);
                named = rc2 == MIPS_OK;
            }
            if (rc2) return rc2;
            if (!named) set_kernel_name(ix, "mips::scan_kernel_v4 experimental instance sub=%d", ix->opt_sub);
        }
    } else if (want_f8x) {
        if constexpr (KL == 8) {
            mips::ScanArgsF8 fa;
            fa.docs = ix->rows;
            fa.qbuf = (const uint8_t*)ix->qbuf.p;
            fa.c = a;
            const int lds = 3 * mips::F8X_DB * ix->ld + 8 * 1024 + 1024 + 16; // ring + threshold words + dump area + arrival counter
            auto gox = [&](auto kern) -> int {
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
                kern<<<grid, 512, lds, st>>>(fa);
                return MIPS_OK;
            };
            int rc2;
#ifdef MIPS_EXPERIMENTAL
            if (ix->ld == 768 && ix->opt_sub == 8) rc2 = gox(mips::scan_kernel_f8x<V4_KLL, 768, 2, 1>); // timing only: no epilogue
            else
#endif
            if (nqt == 1) { // one query tile: non-temporal document DMA
                if (ix->ld == 768) rc2 = gox(mips::scan_kernel_f8x<V4_KLL, 768, 2, 0, true>);
                else if (ix->ld == 512) rc2 = gox(mips::scan_kernel_f8x<V4_KLL, 512, 2, 0, true>);
                else rc2 = gox(mips::scan_kernel_f8x<V4_KLL, 256, 2, 0, true>);
            } else
            if (ix->ld == 768) rc2 = gox(mips::scan_kernel_f8x<V4_KLL, 768, 2>);
            else if (ix->ld == 512) rc2 = gox(mips::scan_kernel_f8x<V4_KLL, 512, 2>);
            else rc2 = gox(mips::scan_kernel_f8x<V4_KLL, 256, 2>);
            if (rc2) return rc2;
            set_kernel_name(ix, nqt == 1 ? "mips::scan_kernel_f8x<%d, %d, 2, 0, true>" : "mips::scan_kernel_f8x<%d, %d, 2, 0, false>", V4_KLL, ix->ld);
        }
    } else if (f8) {
        if constexpr (KL <= 16) {
            mips::ScanArgsF8 fa;
            fa.docs = ix->rows;
            fa.qbuf = (const uint8_t*)ix->qbuf.p;
            fa.c = a;
            const int lds = 3 * mips::V3_DB * ix->ld + 8 * 1024 + 1024 + 16; // ring + threshold words + dump area + arrival counter
            auto gof8 = [&](auto kern) -> int {
                HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
                if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
                kern<<<grid, 512, lds, st>>>(fa);
                return MIPS_OK;
            };
            int rc2;
            if (nqt == 1) { // one query tile: non-temporal document DMA
                if (ix->ld == 1024) rc2 = gof8(mips::scan_kernel_f8<KL, 1024, 2, true>);
                else if (ix->ld == 768) rc2 = gof8(mips::scan_kernel_f8<KL, 768, 2, true>);
                else if (ix->ld == 512) rc2 = gof8(mips::scan_kernel_f8<KL, 512, 2, true>);
                else rc2 = gof8(mips::scan_kernel_f8<KL, 256, 2, true>);
            } else
            if (ix->ld == 1024) rc2 = gof8(mips::scan_kernel_f8<KL, 1024, 2>);
            else if (ix->ld == 768) rc2 = gof8(mips::scan_kernel_f8<KL, 768, 2>);
            else if (ix->ld == 512) rc2 = gof8(mips::scan_kernel_f8<KL, 512, 2>);
            else rc2 = gof8(mips::scan_kernel_f8<KL, 256, 2>);
            if (rc2) return rc2;
            set_kernel_name(ix, nqt == 1 ? "mips::scan_kernel_f8<%d, %d, 2, true>" : "mips::scan_kernel_f8<%d, %d, 2, false>", KL, ix->ld);
        }
    } else if (variant == 1) {
        HIP_TRY(hipFuncSetAttribute((const void*)mips::scan_kernel<KL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    mips::SCAN_LDS_BYTES));
        if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
        mips::scan_kernel<KL><<<grid, mips::SCAN_THREADS, mips::SCAN_LDS_BYTES, st>>>(a);
        set_kernel_name(ix, "mips::scan_kernel<%d>", KL);
    } else if constexpr (!kl_short) {
        // 4-wave configuration, 3-stage ring (d <= 768: 3 x 48 KiB; pitch 1024: 2 x 64 KiB)
#ifdef MIPS_EXPERIMENTAL
        const int lds = (ix->ld == 1024 ? 2 : 3) * mips::V3_DB * ix->ld * 2 + 4 * 1024 + 1024 + 16;
        auto go4 = [&](auto kern) -> int {
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
            kern<<<grid, 256, lds, st>>>(a);
            return MIPS_OK;
        };
#endif
        int rc2;
#ifdef MIPS_EXPERIMENTAL
        if (ix->ld == 768 && ix->opt_sub == 51) { // ring-depth experiment (profiles/r2_pitch1024): the same kernel on a 2-stage ring
            rc2 = go4(mips::scan_kernel_v3<KL, 48, 1, 4, true, 0, 2, 4, 2>);
            if (rc2) return rc2;
            set_kernel_name(ix, "mips::scan_kernel_v3 experimental instance sub=51 (4 waves, 2 stages)");
        } else
#endif
        {   // true K' = 16 / 32 lists: pitches 256 / 512 / 768 and (round 3) 1024 -- k = 8 .. 29 and stage 1 of the two-stage fp32
            // search at Longformer-large width no longer fall back to the generic kernel there
            int nrow = 0;
            const ScanInstance* rows = v3_instances<KL>(&nrow);
            rc2 = launch_row(find_instance(rows, nrow, ix->ld, nqt == 1, 0), KL);
        }
        if (rc2) return rc2;
    } else {
        int rc2 = MIPS_OK;
        bool launched = false;
#ifdef MIPS_EXPERIMENTAL
        const int lds = (v3_waves == 4 ? 2 : 3) * mips::V3_DB * ix->ld * 2 + v3_waves * 1024 + 1024 + 16; // ring + threshold words + dump area + arrival counter
        auto go = [&](auto kern, int threads) -> int {
            HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            if (ix->timing_armed) HIP_TRY(hipEventRecord(ix->ev0[slot], st));
            kern<<<grid, threads, lds, st>>>(a);
            return MIPS_OK;
        };
        // A/B instances of the experiment logs under profiles/ (tools/ab.py builds the library with
        // -DMIPS_EXPERIMENTAL; the shipped library does not contain them: sub 8 / 9 return wrong results by design)
        const int sub = (KL == 8 && ix->ld == 768) ? ix->opt_sub : 0;
        launched = sub != 0;
        if (sub == 3) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, true, 0, 2, 8, 3, false>, 512);  // hardware s_barrier per block
        else if (sub == 7) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, true, 0, 2, 8, 3, true, true>, 512);  // nt document DMA
        else if (sub == 6) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, true, 0, 1, 8, 3, true, false, 1>, 512);  // shared K'-th bests, re-read every block
        else if (sub == 15) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, true, 0, 2, 8, 3, true, false, 1>, 512); // class maxima re-read every block
        else if (sub == 1) rc2 = go(mips::scan_kernel_v3<KL, 48, 2, 6, true, 0, 0, 4, 3, true>, 256);  // 4 waves x 64 queries
        else if (sub == 10) rc2 = go(mips::scan_kernel_v3<6, 48, 1, 2, true>, 512);             // 6-entry lists
        else if (sub == 11) rc2 = go(mips::scan_kernel_v3<6, 48, 1, 3, true>, 512);             // 6-entry lists, prefetch depth 3
        else if (sub == 2) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, false>, 512);            // DMA issued in one burst
        else if (sub == 4) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, true, 0, 0>, 512);       // no shared thresholds
        else if (sub == 5) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 3, true>, 512);             // prefetch depth 3
        else if (sub == 8) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, true, 1>, 512);          // timing only: no epilogue
        else if (sub == 9) rc2 = go(mips::scan_kernel_v3<KL, 48, 1, 2, true, 2>, 512);          // timing only: pre-test only
        else launched = false;
        if (launched) set_kernel_name(ix, "mips::scan_kernel_v3 experimental instance sub=%d", sub);
#endif
        if (!launched) { // K' = 8 / 10: one query tile -> non-temporal document DMA (HBM-bound regime: 5.5 -> 5.9 TB/s at pitch 768)
            int nrow = 0;
            const ScanInstance* rows = v3_instances<KL>(&nrow);
            rc2 = launch_row(find_instance(rows, nrow, ix->ld, nqt == 1, 0), KL);
        }
        if (rc2) return rc2;
    }
    HIP_TRY(hipGetLastError());
    if (ix->timing_armed) {
        HIP_TRY(hipEventRecord(ix->ev1[slot], st));
        ix->ev_next = (slot + 1) % mips_index::kEvRing;
        if (++ix->ev_count == mips_index::kEvRing) ix->timing_armed = false; // window full
    }

    mips::MergeArgs m;
    m.part_s = a.part_s;
    m.part_i = a.part_i;
    m.ncand = (int)ncand;
    const bool f32r = f32x || ix->fast_f32; // exact re-score on the fp32 rows (stage 1 of the two-stage search included)
    m.docs = f32r ? (const void*)ix->rows_f32 : (const void*)ix->rows;
    m.qbuf = f32r ? (const void*)ix->qf32.p : (const void*)a.qbuf;
    m.ld = ix->fast_f32 ? ix->plane_keep : f32x ? ix->plane : ix->ld;
    m.k = k;
    m.metric = ix->call_metric;
    m.phi = ix->phi;
    m.idx_offset = idx_offset;
    m.out_s = d_out_s;
    m.out_i = d_out_i;
    m.out_packed = d_out_packed;
    m.err = a.err;
    m.sticky = ix->sticky_dev;
    m.ll = list_len;
    m.pre_bnd = nullptr;
    m.npre = 0;
    m.bnd = nullptr;
    m.flag = nullptr;
    m.nflag = nullptr;
    m.xmax2 = ix->xmax2_dev;
    // MFMA score = fp32 accumulation of exact products (bf16 x bf16 and e4m3 x e4m3 fit fp32): |error| <= (terms) u
    // sum |q_j x_j| <= d 2^-23 |q| |x| (u = 2^-23 allows truncating adders).  fp32-exact mode scans hi.qhi + hi.qlo +
    // lo.qhi of bf16 splits: the dropped lo.qlo term adds 2^-16 |q| |x|, and there are three times the terms.
    m.err_c = f32x ? (3.0 * (double)ix->d * 1.1920928955078125e-07 + 1.52587890625e-05) : (double)ix->d * 1.1920928955078125e-07;
    m.nq_dev = ix->nq_dev;
    if (ix->fast_f32) { // the scan's operands are bf16(q), bf16(x): norms within 2^-8 of |q|, |x|
        m.err_c *= 1.01;
        m.dres2 = ix->dres2_dev;
        m.qerr2 = (const double*)ix->qerr2.p;
    }
    if (ix->opt_margin != 0) {
        rc = ix->mbnd.ensure((size_t)nq * sizeof(float));
        if (rc) return rc;
        rc = ix->mflag.ensure((size_t)nq);
        if (rc) return rc;
        rc = ensure_xmax2(ix, st);
        if (rc) return rc;
        m.xmax2 = ix->xmax2_dev;
        m.bnd = (float*)ix->mbnd.p;
        m.flag = (unsigned char*)ix->mflag.p;
        m.nflag = (unsigned*)ix->gthr.p + (size_t)nq_pad * 8 + 1; // zeroed with the insert bounds by the query staging
        ix->last_nflag_dev = m.nflag;
        if (ix->rescan_depth == 0) { // what the exact resolution of flagged queries starts from (resolve_kernels.hpp)
            rc = ix->keyk.ensure((size_t)nq * sizeof(float));
            if (rc) return rc;
            rc = ix->qqv.ensure((size_t)nq * sizeof(double));
            if (rc) return rc;
            m.keyk = (float*)ix->keyk.p;
            m.qq_out = (double*)ix->qqv.p;
        }
    }
    // (1) K' best candidates per query by MFMA score, (2) lane-packed exact re-score + final order
    rc = ix->cand.ensure((size_t)nq * KL * sizeof(int));
    if (rc) return rc;
    int* cand = (int*)ix->cand.p;
    const hipStream_t scan_st = st;
    if (split) { // the tail goes to its own stream, behind the scan
        HIP_TRY(hipEventRecord(ix->scan_done, scan_st));
        HIP_TRY(hipStreamWaitEvent(tail_st, ix->scan_done, 0));
        st = tail_st;
    }
    mips::merge_select_kernel<KL><<<(int)nq, 64, 0, st>>>(m, cand);
    HIP_TRY(hipGetLastError());
    const bool l2 = ix->call_metric == MIPS_METRIC_L2;
    const int rgrid = (int)((nq + (64 / KL) - 1) / (64 / KL));
    if (f32r && l2) mips::rescore_rank_kernel<KL, mips::ElemF32, true><<<rgrid, 64, 0, st>>>(m, cand, nq);
    else if (f32r) mips::rescore_rank_kernel<KL, mips::ElemF32, false><<<rgrid, 64, 0, st>>>(m, cand, nq);
    else if (e8 && l2) mips::rescore_rank_kernel<KL, mips::ElemF8, true, mips::ElemBF16><<<rgrid, 64, 0, st>>>(m, cand, nq);
    else if (e8) mips::rescore_rank_kernel<KL, mips::ElemF8, false, mips::ElemBF16><<<rgrid, 64, 0, st>>>(m, cand, nq);
    else if (f8 && l2) mips::rescore_rank_kernel<KL, mips::ElemF8, true><<<rgrid, 64, 0, st>>>(m, cand, nq);
    else if (f8) mips::rescore_rank_kernel<KL, mips::ElemF8, false><<<rgrid, 64, 0, st>>>(m, cand, nq);
    else if (l2) mips::rescore_rank_kernel<KL, mips::ElemBF16, true><<<rgrid, 64, 0, st>>>(m, cand, nq);
    else mips::rescore_rank_kernel<KL, mips::ElemBF16, false><<<rgrid, 64, 0, st>>>(m, cand, nq);
    HIP_TRY(hipGetLastError());
    if (split) {
        HIP_TRY(hipEventRecord(ix->tail_done[ix->cur_set], st));
        ix->tail_pending[ix->cur_set] = true;
    }
    return MIPS_OK;
}

// One-launch search for the reference's own call shape (tiny_search.hpp): <= 16 queries, bf16 index of at most
// kTinyMaxRows rows, k_fetch <= 6.  q must be device memory.
constexpr int64_t kTinyMaxRows = 1 << 16;
// certifies: the call settles the queries it flags (host buffers, "margin_check" = 2, or stream-ordered).  The fp32-exact
// index takes the one-launch kernel only then: its scan sees bf16(x) . bf16(q) (stage 1 of the two-stage search), which is
// admissible because of the certificate alone
bool tiny_eligible(const mips_index* ix, int64_t nq, int k_fetch, bool certifies) {
    if (!(ix->opt_tiny != 0 && nq >= 1 && nq <= 16 && k_fetch <= mips::TINY_MAXK && ix->ntotal > 0 && ix->ntotal <= kTinyMaxRows &&
          ix->esize == 2 && ix->rescan_depth == 0))
        return false;
    if (ix->plane > 0) return certifies && ix->opt_f32_fast != 0 && ix->hp > 0 && ix->hp <= 1024 && ix->hp % 128 == 0;
    return ix->ld <= 1024 && ix->ld % 128 == 0;
}

int ensure_resolve_buffers(mips_index* ix, int64_t nq) {
    int rc = ix->ids.ensure((size_t)(nq + 4) * sizeof(int));
    if (rc) return rc;
    rc = ix->hit_d.ensure((size_t)mips::RESOLVE_MAX * mips::RESOLVE_CAP * sizeof(double));
    if (rc) return rc;
    rc = ix->hit_i.ensure((size_t)mips::RESOLVE_MAX * mips::RESOLVE_CAP * sizeof(int));
    if (rc) return rc;
    rc = ix->hit_n.ensure((size_t)mips::RESOLVE_MAX * sizeof(int));
    if (rc) return rc;
    rc = ix->keyk.ensure((size_t)nq * sizeof(float));
    if (rc) return rc;
    return ix->qqv.ensure((size_t)nq * sizeof(double));
}

// handoff: prepare the stream-ordered exact pass (resolve_flagged with skip_compact) -- the kernel's last workgroup writes
// the flag list, clears the hit counters and copies the staged queries out when something was flagged
int tiny_search(mips_index* ix, const void* q_dev, int q_dtype, int64_t nq, int k_fetch, int k_out, int normalize, const int64_t* ignore_dev,
                float* d_s, int64_t* d_i, bool packed, int64_t idx_offset, hipStream_t st, bool handoff) {
    const bool f32x = ix->plane > 0;
    const int tld = f32x ? ix->hp : ix->ld; // row pitch of the scanned bf16 rows
    if (f32x) { // bf16 rows + residual bound up to date, max |x|^2 on the fp32 rows
        int rc0 = ensure_hi(ix, st);
        if (rc0) return rc0;
    }
    if (!ix->tiny_words) {
        HIP_TRY(hipMalloc((void**)&ix->tiny_words, 64));
        HIP_TRY(hipMemsetAsync(ix->tiny_words, 0, 64, st)); // the ticket starts at 0; the kernel's last workgroup resets it
    }
    const int ntiles = (int)((ix->ntotal + 15) / 16);
    // one tile per wave, 8 waves per workgroup, while the chip has the CUs for it (more workgroups = more candidates for
    // the last one to sift; spreading thinner did not make the first tiles arrive sooner)
    const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>(mips::TINY_MAX_WG, (ntiles + mips::TINY_WAVES - 1) / mips::TINY_WAVES));
    mips::TinyArgs a;
    a.docs = f32x ? (const uint16_t*)ix->rows_hi : (const uint16_t*)ix->rows;
    a.rows_f32 = ix->rows_f32;
    a.plane = ix->plane;
    a.q = q_dev;
    a.q_is_f32 = q_dtype == MIPS_DTYPE_F32 ? 1 : 0;
    a.normalize = normalize;
    a.nq = (int)nq;
    a.d = (int)ix->d;
    a.ld = tld;
    a.ntotal = ix->ntotal;
    a.res_ids = nullptr;
    a.res_cnt = nullptr;
    a.res_hit_n = nullptr;
    a.res_unres = nullptr;
    a.q_out = nullptr;
    a.ntiles = ntiles;
    a.nwaves = nwg * mips::TINY_WAVES;
    a.force_slow = ix->opt_tiny == 2 ? 1 : 0;
    a.ticket = ix->tiny_words;
    a.ignore = ignore_dev;
    a.k_out = k_out;
    a.out_s = d_s;
    a.out_i = d_i;
    a.out_packed = packed ? d_i : nullptr;
    const size_t ncand = (size_t)nwg * mips::TINY_POOL; // per query: the 8 best of every workgroup
    int rc = ix->part_s.ensure(16 * (ncand + nwg) * sizeof(float)); // + the workgroups' bounds behind the candidates
    if (rc) return rc;
    rc = ix->part_i.ensure(16 * ncand * sizeof(int));
    if (rc) return rc;
    mips::MergeArgs& m = a.m;
    m.part_s = (const float*)ix->part_s.p;
    m.part_i = (const int*)ix->part_i.p;
    m.ncand = (int)ncand;
    m.pre_bnd = (const float*)ix->part_s.p + 16 * ncand;
    m.npre = nwg;
    m.docs = a.docs;
    m.qbuf = nullptr;
    m.ld = tld;
    m.k = k_fetch;
    m.metric = ix->call_metric;
    m.phi = ix->phi;
    m.idx_offset = idx_offset;
    m.out_s = nullptr;
    m.out_i = nullptr;
    m.out_packed = nullptr;
    m.err = nullptr;
    m.sticky = ix->sticky_dev;
    m.ll = 0x7fffffff; // final level: no "last entry of a full list" rule, the workgroups' bounds carry that
    m.bnd = nullptr;
    m.flag = nullptr;
    m.nflag = nullptr;
    m.xmax2 = ix->xmax2_dev;
    m.err_c = (double)ix->d * 1.1920928955078125e-07 * (f32x ? 1.01 : 1.0); // (F32: the scan's operands are bf16(q), bf16(x))
    m.dres2 = ix->dres2_dev;
    ix->last_flagged = -1;
    ix->last_rescanned = 0;
    ix->last_unresolved = 0;
    ix->last_nflag_dev = nullptr;
    ix->first_nflag_dev = nullptr;
    if (ix->opt_margin != 0) {
        rc = ix->mbnd.ensure(16 * sizeof(float));
        if (rc) return rc;
        rc = ix->mflag.ensure(16);
        if (rc) return rc;
        rc = ensure_xmax2(ix, st);
        if (rc) return rc;
        m.xmax2 = ix->xmax2_dev;
        m.bnd = (float*)ix->mbnd.p;
        m.flag = (unsigned char*)ix->mflag.p;
        m.nflag = ix->tiny_words + 1;
        ix->last_nflag_dev = m.nflag;
        if (handoff) {
            rc = ensure_resolve_buffers(ix, nq);
            if (rc) return rc;
            if (f32x) rc = ix->qf32.ensure((size_t)16 * ix->plane * sizeof(float));
            else rc = ix->qbuf.ensure((size_t)16 * ix->ld * 2);
            if (rc) return rc;
            a.res_ids = (int*)ix->ids.p;
            a.res_cnt = a.res_ids + nq;
            a.res_unres = (unsigned*)(a.res_ids + nq + 1);
            a.res_hit_n = (int*)ix->hit_n.p;
            a.q_out = f32x ? ix->qf32.p : ix->qbuf.p;
            m.keyk = (float*)ix->keyk.p;
            m.qq_out = (double*)ix->qqv.p;
        }
    } else if (f32x) {
        return fail(MIPS_E_INVALID, "tiny_search: the fp32-exact index needs the margin check");
    }
    const int lds = mips::tiny_lds_bytes(tld, ix->plane);
#ifdef MIPS_EXPERIMENTAL
    static unsigned long long* dbg_dev = nullptr;
    const bool dbg = getenv("MIPS_TINY_DBG") != nullptr;
    if (dbg && !dbg_dev) HIP_TRY(hipMalloc((void**)&dbg_dev, 256 * 16 * 8));
    a.dbg = dbg ? dbg_dev : nullptr;
    if (dbg) HIP_TRY(hipMemsetAsync(dbg_dev, 0, 256 * 16 * 8, st));
#endif
    auto go = [&](auto kern) -> int {
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        kern<<<nwg, mips::TINY_THREADS, lds, st>>>(a);
        return MIPS_OK;
    };
    const bool l2m = ix->call_metric == MIPS_METRIC_L2;
    if (f32x) rc = l2m ? go(mips::tiny_search_kernel<true, true>) : go(mips::tiny_search_kernel<false, true>);
    else rc = l2m ? go(mips::tiny_search_kernel<true, false>) : go(mips::tiny_search_kernel<false, false>);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
#ifdef MIPS_EXPERIMENTAL
    if (dbg) { // phase stamps (10 ns units) relative to the first workgroup's start: the slowest workgroup per phase and the last one
        std::vector<unsigned long long> h(256 * 16);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h.data(), dbg_dev, 256 * 16 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        int last = 0;
        for (int b = 0; b < nwg; ++b) {
            t0 = std::min(t0, h[b * 16]);
            if (h[b * 16 + 11]) last = b;
        }
        unsigned long long mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < nwg; ++b)
            for (int i = 0; i < 8; ++i) mx[i] = std::max(mx[i], h[b * 16 + i] - t0);
        fprintf(stderr, "tiny x10ns max/%d wgs: start %llu docs-issued %llu zero-rows %llu rows-staged %llu barrier %llu scanned %llu selected %llu ticket %llu | last wg %d: ticket %llu select2 %llu dots %llu ranked %llu end %llu | shader clock %.0f MHz\n",
                nwg, mx[0], mx[1], mx[2], mx[3], mx[4], mx[5], mx[6], mx[7], last, h[last * 16 + 7] - t0, h[last * 16 + 8] - t0,
                h[last * 16 + 9] - t0, h[last * 16 + 10] - t0, h[last * 16 + 11] - t0,
                (double)(h[last * 16 + 13] - h[last * 16 + 12]) / ((double)(h[last * 16 + 11] - h[last * 16]) * 0.01));
    }
#endif
    set_kernel_name(ix, "mips::tiny_search_kernel<%s, %s>", l2m ? "true" : "false", f32x ? "true" : "false");
    return MIPS_OK;
}

// Exact resolution of the flagged queries (resolve_kernels.hpp): flag list + count on the device, one pass over the stored
// rows per 8 flagged queries computing canonical scores, the hit lists ranked over the first results.
// certify_now: the call synchronises anyway (host buffers / "margin_check" = 2): the count is read first, and a search that
// flagged more than RESOLVE_MAX queries is handed to the tile re-scan (return value kUseRescan).  Otherwise everything is
// enqueued blind; the counts travel to host-visible words for the next search to look at (mips_index::stats_host).
constexpr int kUseRescan = 1;
// skip_compact: the flag list, its count and the cleared counters are already on the device (the one-launch kernel's hand-off).
// ignore / k_out: the fused hook call's ignore filter (ResolveArgs), d_s / d_i then are [nq][k_out].
int resolve_flagged(mips_index* ix, int64_t nq, int k, float* d_s, int64_t* d_i, bool packed, int64_t idx_offset, hipStream_t st, bool certify_now,
                    bool skip_compact = false, const int64_t* ignore = nullptr, int k_out = 0) {
    int rc = ensure_resolve_buffers(ix, nq); // (never reallocates behind a hand-off: same sizes as tiny_search asked for)
    if (rc) return rc;
    int* ids = (int*)ix->ids.p;
    int* cnt = ids + nq;
    unsigned* unres = (unsigned*)(ids + nq + 1);
    if (!ix->nflag_host) HIP_TRY(hipHostMalloc((void**)&ix->nflag_host, 64, hipHostMallocDefault));
    const int max_n = ix->resolve_budget > 0 ? std::min(ix->resolve_budget, mips::RESOLVE_MAX) : mips::RESOLVE_MAX;
    if (!skip_compact)
        mips::compact_flags_kernel<<<1, 256, 0, st>>>((const unsigned char*)ix->mflag.p, (int)nq, ids, cnt, (int*)ix->hit_n.p, mips::RESOLVE_MAX, unres);
    if (certify_now) {
        HIP_TRY(hipMemcpyAsync(&ix->nflag_host[0], cnt, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const int64_t n = (int64_t)ix->nflag_host[0];
        ix->last_flagged = n;
        ix->last_rescanned = 0;
        ix->last_unresolved = 0;
        if (n == 0) return MIPS_OK;
        if (n > max_n && !skip_compact) return kUseRescan; // (the one-launch kernel's <= 16 queries have no tile re-scan to go to)
    }
    mips::ResolveArgs a;
    const bool f32x = ix->plane > 0;
    a.rows = f32x ? (const void*)ix->rows_f32 : (const void*)ix->rows;
    a.y = f32x ? (const void*)ix->qf32.p : (const void*)ix->qbuf.p;
    a.ld = f32x ? ix->plane : ix->ld;
    a.ntotal = ix->ntotal;
    a.ids = ids;
    a.n_dev = cnt;
    a.max_n = max_n;
    ix->last_max_n = a.max_n;
    ix->last_fallback = false;
    a.keyk = (const float*)ix->keyk.p;
    a.qq = (const double*)ix->qqv.p;
    a.phi = ix->phi;
    a.hit_d = (double*)ix->hit_d.p;
    a.hit_i = (int*)ix->hit_i.p;
    a.hit_n = (int*)ix->hit_n.p;
    a.k = k;
    a.idx_offset = idx_offset;
    a.out_s = d_s;
    a.out_i = d_i;
    a.out_packed = packed ? d_i : nullptr;
    a.unresolved = unres;
    a.ignore = ignore;
    a.k_out = ignore ? k_out : 0;
    int lds = mips::RESOLVE_QB * a.ld * (int)sizeof(double) + mips::RESOLVE_WAVES * 64 * 9 * 16;
    const int rows_per_wg = 64 * mips::RESOLVE_WAVES;
    int grid = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (ix->ntotal + rows_per_wg - 1) / rows_per_wg));
    const bool l2 = ix->call_metric == MIPS_METRIC_L2;
    auto go = [&](auto kern) -> int {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        kern<<<grid, 64 * mips::RESOLVE_WAVES, lds, st>>>(a);
        return MIPS_OK;
    };
    // bf16-stored rows and the fp32-exact index: the same pass behind an MFMA pre-filter (16 flagged queries per pass, canonical
    // evaluation of the few rows whose approximate score comes within the error bound of the k-th key; "resolve" = 2 keeps the
    // plain form -- tests compare the two)
    const int ffld = f32x ? ix->hp : ix->ld;
    const bool mfma_filter = ix->opt_resolve == 1 && (f32x ? (ix->hp > 0 && ix->rows_hi != nullptr) : (ix->esize == 2 && !ix->mixed)) &&
                             ffld % 64 == 0 && ffld <= 1024 && a.ld <= 1024;
    if (mfma_filter) {
        if (f32x) {
            rc = ensure_hi(ix, st);
            if (rc) return rc;
        }
        rc = ensure_xmax2(ix, st);
        if (rc) return rc;
        a.frows = f32x ? (const uint16_t*)ix->rows_hi : (const uint16_t*)ix->rows;
        a.fld = ffld;
        a.xmax2 = ix->xmax2_dev;
        a.dres2 = ix->dres2_dev;
        a.err_c = (double)ix->d * 1.1920928955078125e-07 * (f32x ? 1.01 : 1.0);
        // query image (row pitch + 16 B) + the waves' product staging + statistics / thresholds / keys / ids
        lds = ((mips::RESOLVE_QM * (ffld * 2 + 16) + 15) & ~15) + mips::RESOLVE_WAVES * 64 * 8 * 8 + 2 * mips::RESOLVE_QM * 8 + 3 * mips::RESOLVE_QM * 4;
        const int64_t tiles = (ix->ntotal + 15) / 16;
        grid = (int)std::max<int64_t>(1, std::min<int64_t>(512, (tiles + mips::RESOLVE_WAVES - 1) / mips::RESOLVE_WAVES));
        if (f32x) rc = l2 ? go(mips::exact_filter_mfma_kernel<true, true>) : go(mips::exact_filter_mfma_kernel<false, true>);
        else rc = l2 ? go(mips::exact_filter_mfma_kernel<true, false>) : go(mips::exact_filter_mfma_kernel<false, false>);
    } else if (f32x) rc = l2 ? go(mips::exact_filter_kernel<mips::ElemF32, true>) : go(mips::exact_filter_kernel<mips::ElemF32, false>);
    else if (ix->mixed) rc = l2 ? go(mips::exact_filter_kernel<mips::ElemF8, true, mips::ElemBF16>) : go(mips::exact_filter_kernel<mips::ElemF8, false, mips::ElemBF16>);
    else if (ix->esize == 1) rc = l2 ? go(mips::exact_filter_kernel<mips::ElemF8, true>) : go(mips::exact_filter_kernel<mips::ElemF8, false>);
    else rc = l2 ? go(mips::exact_filter_kernel<mips::ElemBF16, true>) : go(mips::exact_filter_kernel<mips::ElemBF16, false>);
    if (rc) return rc;
    const int fgrid = (int)std::min<int64_t>(nq, certify_now ? (int64_t)ix->nflag_host[0] : (int64_t)a.max_n); // (at least one block: it counts an over-budget search)
    if (l2) mips::resolve_finalize_kernel<true><<<fgrid, 64, 0, st>>>(a);
    else mips::resolve_finalize_kernel<false><<<fgrid, 64, 0, st>>>(a);
    HIP_TRY(hipGetLastError());
    ix->first_nflag_dev = (const int*)cnt;
    ix->last_nflag_dev = unres;
    if (certify_now) {
        HIP_TRY(hipMemcpyAsync(&ix->nflag_host[1], unres, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        ix->last_rescanned = ix->last_flagged > a.max_n ? 0 : ix->last_flagged; // (over the budget: nothing was settled)
        ix->last_unresolved = (int64_t)ix->nflag_host[1];
    } else {
        ix->last_flagged = -1; // (device only: mips_index_margin_stats fetches the two counters when asked)
    }
    return MIPS_OK;
}

// "margin_check" = 3, device outputs: the re-scan of finish_margin without its two synchronisations.  The flags of the first
// scan are compacted into a list + count ON THE DEVICE; the staged rows of the flagged queries are gathered; the second scan
// (widest lists) is launched for ALL nq queries' worth of workgroups, which read the count and leave when they are past it
// (ScanArgs::nq_dev); select, re-score and the scatter over the first results do the same.  With nothing flagged this costs
// a handful of empty launches (tens of microseconds); mips_index_margin_stats reads both counters when asked.
// gate_above >= 0: the fall-back behind the exact pass for searches whose FIRST scan was an optimistic one (two-stage fp32 search,
// pools of 32 out of sub-lists): the exact pass leaves a search that flagged more than gate_above queries alone, and first results
// selected by bf16 scores / short sub-lists must not stand uncertified -- so this re-scan runs exactly then (its launches are
// sized by a count that is 0 otherwise) and its own still-flagged count replaces the exact pass's "unresolved".
template <int KL>
int rescan_on_stream(mips_index* ix, int64_t nq, int k, float* d_s, int64_t* d_i, bool packed, int64_t idx_offset, hipStream_t st, bool fast_first,
                     int gate_above = -1) {
    const bool f8 = ix->esize == 1;
    const int wide = f8 ? (KL < 16 ? 16 : 0) : (KL < 32 || fast_first ? 32 : 0);
    if (gate_above < 0) {
        ix->last_flagged = -1; // (device only)
        ix->last_max_n = 0;
    }
    if (wide == 0) return MIPS_OK; // already on the widest lists: counted only
    const int64_t n_pad = query_pad(ix, nq);
    const size_t row_bytes = (size_t)ix->ld * ix->qsize;
    int rc = ix->ids.ensure((size_t)(nq + 4) * sizeof(int));
    if (rc) return rc;
    int* ids = (int*)ix->ids.p;
    int* cnt = ids + nq + (gate_above >= 0 ? 2 : 0); // (gated: the exact pass's own count and unresolved counter stay where they are)
    unsigned* const exact_unres = (unsigned*)(ids + nq + 1);
    const int* const first_keep = ix->first_nflag_dev;
    rc = ix->qbuf2.ensure((size_t)n_pad * row_bytes);
    if (rc) return rc;
    rc = ix->tmp_s.ensure((size_t)nq * k * sizeof(float));
    if (rc) return rc;
    rc = ix->tmp_i.ensure((size_t)nq * k * sizeof(int64_t) * 2);
    if (rc) return rc;
    mips::compact_flags_kernel<<<1, 256, 0, st>>>((const unsigned char*)ix->mflag.p, (int)nq, ids, cnt, nullptr, 0, nullptr, gate_above);
    mips::gather_rows_kernel<<<grid_for(n_pad * (int64_t)(row_bytes / 16), 256), 256, 0, st>>>((const unsigned char*)ix->qbuf.p, ids, 0, n_pad, (int)row_bytes,
                                                                                              (unsigned char*)ix->qbuf2.p, cnt);
    if (ix->plane > 0) {
        const size_t rb32 = (size_t)ix->plane * sizeof(float);
        rc = ix->qf32b.ensure((size_t)n_pad * rb32);
        if (rc) return rc;
        mips::gather_rows_kernel<<<grid_for(n_pad * (int64_t)(rb32 / 16), 256), 256, 0, st>>>((const unsigned char*)ix->qf32.p, ids, 0, n_pad, (int)rb32,
                                                                                            (unsigned char*)ix->qf32b.p, cnt);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(ix->gthr.p, 0, (size_t)(n_pad * 8 + 4) * sizeof(unsigned), st)); // insert bounds, error word, flag counter
    std::swap(ix->qbuf, ix->qbuf2);
    std::swap(ix->qf32, ix->qf32b);
    const bool armed = ix->timing_armed;
    char name_keep[sizeof ix->last_kernel];
    std::memcpy(name_keep, ix->last_kernel, sizeof name_keep);
    ix->timing_armed = false;
    ix->rescan_depth = 1;
    ix->nq_dev = cnt;
    const int ns_keep = ix->opt_nsplit;
    if (ns_keep == 0 && nq <= 8192) ix->opt_nsplit = nq <= 4096 ? 128 : 64; // the flagged queries are few: spread each of their tiles over many CUs
    float* ts = (float*)ix->tmp_s.p;
    int64_t* ti = (int64_t*)ix->tmp_i.p;
    if (wide == 32) rc = launch_search<32>(ix, nq, k, ts, ti, packed ? ti : nullptr, idx_offset, st, nullptr, false);
    else rc = launch_search<16>(ix, nq, k, ts, ti, packed ? ti : nullptr, idx_offset, st, nullptr, false);
    ix->opt_nsplit = ns_keep;
    ix->nq_dev = nullptr;
    ix->rescan_depth = 0;
    ix->timing_armed = armed;
    std::memcpy(ix->last_kernel, name_keep, sizeof name_keep);
    std::swap(ix->qbuf, ix->qbuf2);
    std::swap(ix->qf32, ix->qf32b);
    if (rc) return rc;
    if (packed) {
        mips::scatter_i64_kernel<<<grid_for(nq * 2 * k, 256), 256, 0, st>>>(ti, ids, 0, 2 * k, d_i, cnt);
    } else {
        mips::scatter_i64_kernel<<<grid_for(nq * k, 256), 256, 0, st>>>(ti, ids, 0, k, d_i, cnt);
        mips::scatter_f32_kernel<<<grid_for(nq * k, 256), 256, 0, st>>>(ts, ids, 0, k, d_s, cnt);
    }
    HIP_TRY(hipGetLastError());
    if (gate_above >= 0) { // statistics stay the exact pass's; if this re-scan ran, what IT still flags is what is unresolved
        mips::adopt_rescan_count_kernel<<<1, 1, 0, st>>>(cnt, ix->last_nflag_dev, exact_unres);
        HIP_TRY(hipGetLastError());
        ix->first_nflag_dev = first_keep;
        ix->last_nflag_dev = exact_unres;
        ix->last_fallback = true;
        return MIPS_OK;
    }
    ix->first_nflag_dev = (const int*)cnt; // last_nflag_dev: the re-scan's own counter (still flagged on the widest lists)
    return MIPS_OK;
}

// Margin check, host side.  The re-score flagged every query whose k-th exact score is within the MFMA error bound of
// what the candidate pool may have excluded (aux_kernels.hpp).  When the call may synchronise (host buffers, or
// "margin_check" = 2) the flagged queries are re-scanned with the widest lists (K' = 32; 16 on an fp8 index): their
// staged rows are gathered into a compact query buffer, searched again, and the rows scattered over the first
// results.  Queries still flagged after that are counted as unresolved (mips_index_margin_stats).
template <int KL>
int finish_margin(mips_index* ix, int64_t nq, int k, float* d_s, int64_t* d_i, bool packed, int64_t idx_offset, bool out_dev,
                  hipStream_t st, bool fast_first = false) {
    ix->last_flagged = -1;
    ix->last_rescanned = 0;
    ix->last_unresolved = 0;
    ix->first_nflag_dev = nullptr;
    if (ix->opt_margin == 0 || ix->rescan_depth != 0) return MIPS_OK;
    if (out_dev && ix->opt_margin != 2 && ix->opt_margin != 3) return MIPS_OK; // counted on the device only
    // flagged queries are settled exactly, by brute force on the canonical scores (rows of up to 1024 columns; beyond that
    // -- and with "resolve" = 0 -- by the re-scan with the widest lists below)
    if (ix->opt_resolve != 0 && (ix->plane > 0 ? ix->plane : ix->ld) <= 1024) {
        const bool stream_ordered = out_dev && ix->opt_margin == 3;
        const int r = resolve_flagged(ix, nq, k, d_s, d_i, packed, idx_offset, st, !stream_ordered);
        if (r == MIPS_OK && stream_ordered && fast_first) {
            const int max_n = ix->resolve_budget > 0 ? std::min(ix->resolve_budget, mips::RESOLVE_MAX) : mips::RESOLVE_MAX;
            if (nq > max_n) return rescan_on_stream<KL>(ix, nq, k, d_s, d_i, packed, idx_offset, st, fast_first, max_n);
        }
        if (r != kUseRescan) return r;
        ix->first_nflag_dev = nullptr; // (more flagged than the exact pass takes: the tile re-scan below, which synchronises)
    } else if (out_dev && ix->opt_margin == 3) {
        return rescan_on_stream<KL>(ix, nq, k, d_s, d_i, packed, idx_offset, st, fast_first);
    }
    if (!ix->nflag_host) HIP_TRY(hipHostMalloc((void**)&ix->nflag_host, 64, hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(ix->nflag_host, ix->last_nflag_dev, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t n = (int64_t)ix->nflag_host[0];
    ix->last_flagged = n;
    if (n == 0) return MIPS_OK;
    constexpr int WIDE = 32;
    const bool f8 = ix->esize == 1;
    // (fast_first: the scan just done was stage 1 of the two-stage fp32 search -- the re-scan is the three-segment scan)
    const int wide = f8 ? (KL < 16 ? 16 : 0) : (KL < WIDE || fast_first ? WIDE : 0);
    if (wide == 0) { // already on the widest lists this storage type has
        ix->last_unresolved = n;
        return MIPS_OK;
    }
    // flagged query numbers
    std::string flags((size_t)nq, '\0');
    HIP_TRY(hipMemcpyAsync(&flags[0], ix->mflag.p, (size_t)nq, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::string idbuf((size_t)n * sizeof(int), '\0');
    int* ids_h = reinterpret_cast<int*>(&idbuf[0]);
    int64_t w = 0;
    for (int64_t q = 0; q < nq && w < n; ++q)
        if (flags[(size_t)q]) ids_h[w++] = (int)q;
    if (w != n) return fail(MIPS_E_HIP, "margin check: flag count %lld does not match the flag array (%lld)", (long long)n, (long long)w);
    int rc = ix->ids.ensure((size_t)n * sizeof(int));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ix->ids.p, ids_h, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    const int* ids = (const int*)ix->ids.p;
    const int64_t n_pad = query_pad(ix, n);
    const size_t row_bytes = (size_t)ix->ld * ix->qsize;
    rc = ix->qbuf2.ensure((size_t)n_pad * row_bytes);
    if (rc) return rc;
    mips::gather_rows_kernel<<<grid_for(n_pad * (int64_t)(row_bytes / 16), 256), 256, 0, st>>>((const unsigned char*)ix->qbuf.p, ids, n, n_pad,
                                                                                              (int)row_bytes, (unsigned char*)ix->qbuf2.p);
    if (ix->plane > 0) {
        const size_t rb32 = (size_t)ix->plane * sizeof(float);
        rc = ix->qf32b.ensure((size_t)n_pad * rb32);
        if (rc) return rc;
        mips::gather_rows_kernel<<<grid_for(n_pad * (int64_t)(rb32 / 16), 256), 256, 0, st>>>((const unsigned char*)ix->qf32.p, ids, n, n_pad, (int)rb32,
                                                                                            (unsigned char*)ix->qf32b.p);
    }
    HIP_TRY(hipGetLastError());
    rc = ix->tmp_s.ensure((size_t)n * k * sizeof(float));
    if (rc) return rc;
    rc = ix->tmp_i.ensure((size_t)n * k * sizeof(int64_t) * 2);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ix->gthr.p, 0, (size_t)(n_pad * 8 + 4) * sizeof(unsigned), st)); // insert bounds, error word, flag counter
    std::swap(ix->qbuf, ix->qbuf2);
    std::swap(ix->qf32, ix->qf32b);
    const bool armed = ix->timing_armed;
    char name_keep[sizeof ix->last_kernel];
    std::memcpy(name_keep, ix->last_kernel, sizeof name_keep);
    ix->timing_armed = false; // the bench's event window times the first scan only
    ix->rescan_depth = 1;
    // few flagged queries = few query tiles: spread each tile's scan over many more splits than the automatic choice makes
    // (it stops at 64; one 128-query tile of the three-segment scan on 64 workgroups took 17 ms at 2^20 x 768)
    const int ns_keep = ix->opt_nsplit;
    if (ns_keep == 0 && n <= 1024) ix->opt_nsplit = (int)std::max<int64_t>(64, std::min<int64_t>(256, round_up(512 / ((n + 127) / 128), 8)));
    float* ts = (float*)ix->tmp_s.p;
    int64_t* ti = (int64_t*)ix->tmp_i.p;
    if (wide == 32) rc = launch_search<32>(ix, n, k, ts, ti, packed ? ti : nullptr, idx_offset, st);
    else rc = launch_search<16>(ix, n, k, ts, ti, packed ? ti : nullptr, idx_offset, st);
    ix->opt_nsplit = ns_keep;
    ix->rescan_depth = 0;
    ix->timing_armed = armed;
    std::memcpy(ix->last_kernel, name_keep, sizeof name_keep);
    std::swap(ix->qbuf, ix->qbuf2);
    std::swap(ix->qf32, ix->qf32b);
    if (rc) return rc;
    if (packed) {
        mips::scatter_i64_kernel<<<grid_for(n * 2 * k, 256), 256, 0, st>>>(ti, ids, n, 2 * k, d_i);
    } else {
        mips::scatter_i64_kernel<<<grid_for(n * k, 256), 256, 0, st>>>(ti, ids, n, k, d_i);
        mips::scatter_f32_kernel<<<grid_for(n * k, 256), 256, 0, st>>>(ts, ids, n, k, d_s);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ix->nflag_host, ix->last_nflag_dev, 4, hipMemcpyDeviceToHost, st)); // still flagged on the widest lists
    HIP_TRY(hipStreamSynchronize(st));
    ix->last_rescanned = n;
    ix->last_unresolved = (int64_t)ix->nflag_host[0];
    return MIPS_OK;
}

// One scan + select + exact re-score + margin finish at list length KL.  fast: stage 1 of the two-stage search of an
// fp32-exact index -- for the duration of the launch the index is viewed as the bf16 index rows_hi (pitch hp) with the
// bf16 queries qhi; the re-score and the margin check still run on the fp32 rows (launch_search: fast_f32).  Queries the
// widened margin cannot certify are re-scanned by finish_margin on the three-segment scan with K' = 32 lists.
template <int KL>
int scan_and_finish(mips_index* ix, int64_t nq, int k, float* d_s, int64_t* d_i, bool packed, int64_t idx_offset, bool out_dev, hipStream_t st,
                    hipStream_t tail_st, bool split, bool fast, bool optimistic = false) {
    int rc;
    ix->optimistic = fast || optimistic;
    if (fast) {
        uint8_t* rows_keep = ix->rows;
        const int ld_keep = ix->ld, plane_keep = ix->plane;
        ix->rows = ix->rows_hi;
        ix->ld = ix->hp;
        ix->plane = 0;
        ix->plane_keep = plane_keep;
        ix->fast_f32 = true;
        std::swap(ix->qbuf, ix->qhi);
        rc = launch_search<KL>(ix, nq, k, d_s, d_i, packed ? d_i : nullptr, idx_offset, st, tail_st, split);
        std::swap(ix->qbuf, ix->qhi);
        ix->rows = rows_keep;
        ix->ld = ld_keep;
        ix->plane = plane_keep;
        ix->fast_f32 = false;
    } else {
        rc = launch_search<KL>(ix, nq, k, d_s, d_i, packed ? d_i : nullptr, idx_offset, st, tail_st, split);
    }
    const bool first_was_optimistic = ix->optimistic;
    ix->optimistic = false;
    if (rc) return rc;
    if (split) { // scan on st, tail on tail_st: the certificate, when asked for, is part of the tail
        ix->last_flagged = -1;
        ix->first_nflag_dev = nullptr;
        if (ix->opt_margin == 3 && out_dev && ix->opt_resolve != 0 && (ix->plane > 0 ? ix->plane : ix->ld) <= 1024) {
            rc = resolve_flagged(ix, nq, k, d_s, d_i, packed, idx_offset, tail_st, false);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ix->tail_done[ix->cur_set], tail_st)); // (supersedes the record behind the re-score: the exact
            ix->tail_pending[ix->cur_set] = true;                         // pass reads this set's staged queries)
        }
        return MIPS_OK;
    }
    rc = finish_margin<KL>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, first_was_optimistic);
    // the optimistic scan pays while few queries need the second one: after a call that sent more than a quarter there, skip
    // it for a while
    if (!rc && first_was_optimistic && ix->opt_f32_fast != 2 && ix->last_flagged >= 64 && ix->last_flagged * 8 > nq) ix->fast_skip = 8;
    return rc;
}

// "margin_check" as the caller set it -> the mode the code below acts on, for the duration of one call:
//   1 (default, "auto")  device outputs: 3 = certify on the stream; host buffers: they synchronise anyway and certify
//   4 ("count only")     1 in the code below: device outputs count flagged queries, nothing more
// 0 / 2 / 3 as they are.  Restored when the call returns.
struct MarginScope {
    mips_index* ix;
    int keep;
    MarginScope(mips_index* ix_, bool out_dev) : ix(ix_), keep(ix_->opt_margin) {
        if (keep == 1 && out_dev) ix->opt_margin = 3;
        else if (keep == 4) ix->opt_margin = 1;
    }
    ~MarginScope() { ix->opt_margin = keep; }
};

} // namespace

extern "C" {

int mips_abi_version(void) { return MIPS_ABI_VERSION; }

const char* mips_last_error(void) { return g_err.c_str(); }

int mips_index_create(mips_index_t** out, int device, int64_t d, int doc_dtype, int metric) {
    if (!out) return fail(MIPS_E_INVALID, "mips_index_create: out is NULL");
    *out = nullptr;
    if (d <= 0 || d > (1 << 20)) return fail(MIPS_E_INVALID, "mips_index_create: bad dimension %lld", (long long)d);
    if (metric != MIPS_METRIC_IP && metric != MIPS_METRIC_L2)
        return fail(MIPS_E_INVALID, "mips_index_create: metric must be 0 (inner product) or 1 (L2), got %d", metric);
    if (doc_dtype != MIPS_DTYPE_BF16 && doc_dtype != MIPS_DTYPE_FP8_E4M3 && doc_dtype != MIPS_DTYPE_F32 && doc_dtype != MIPS_DTYPE_FP8_E4M3_DOCS)
        return fail(MIPS_E_INVALID, "mips_index_create: index storage dtype must be BF16, FP8_E4M3, FP8_E4M3_DOCS or F32, got %d", doc_dtype);
    const bool f8_storage = doc_dtype == MIPS_DTYPE_FP8_E4M3 || doc_dtype == MIPS_DTYPE_FP8_E4M3_DOCS;
    if (f8_storage && d > 1024)
        return fail(MIPS_E_UNSUPPORTED, "mips_index_create: fp8 e4m3 storage supports d <= 1024 in this build");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(MIPS_E_INVALID, "mips_index_create: no HIP device %d (have %d)", device, count);
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    mips_index* ix = new (std::nothrow) mips_index();
    if (!ix) return fail(MIPS_E_NOMEM, "out of host memory");
    ix->device = device;
    ix->d = d;
    ix->esize = f8_storage ? 1 : 2;
    ix->mixed = doc_dtype == MIPS_DTYPE_FP8_E4M3_DOCS;
    ix->qsize = ix->mixed ? 2 : ix->esize;
    // Row pitch.  bf16: the query-stationary kernels exist for pitches of 128 .. 768 (multiples of 128) and
    // 1024, so every d <= 1024 is padded to one of those (zero columns); beyond that the generic kernel
    // takes multiples of 64.  fp8: multiples of 256 bytes.  fp32-exact: generic kernel over two planes.
    if (f8_storage) ix->ld = (int)round_up(d, 256);
    else if (doc_dtype == MIPS_DTYPE_F32 || d > 1024) ix->ld = (int)round_up(d, mips::BK);
    else ix->ld = d > 768 ? 1024 : (int)round_up(d, 128);
    if (doc_dtype == MIPS_DTYPE_F32) {
        ix->plane = ix->ld;
        ix->ld = 2 * ix->plane;
        // a pitch at which the query-stationary kernels serve pools of 32: 256 (true K' = 32 lists), 384 .. 768 (16x16x32
        // kernel, optimistic pools); 1024: K' <= 10 only
        if (d <= 1024) ix->hp = d <= 256 ? 256 : d <= 768 ? (int)round_up(d, 128) : 1024;
    }
    ix->doc_dtype = doc_dtype;
    ix->metric = metric;
    for (int e = 0; e < mips_index::kEvRing; ++e)
        if (hipEventCreate(&ix->ev0[e]) != hipSuccess || hipEventCreate(&ix->ev1[e]) != hipSuccess) {
            mips_index_destroy(ix);
            return fail(MIPS_E_HIP, "hipEventCreate failed");
        }
    if (hipEventCreateWithFlags(&ix->busy, hipEventDisableTiming) != hipSuccess) {
        mips_index_destroy(ix);
        return fail(MIPS_E_HIP, "hipEventCreate failed");
    }
    if (hipEventCreateWithFlags(&ix->scan_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ix->tail_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ix->tail_done[1], hipEventDisableTiming) != hipSuccess) {
        mips_index_destroy(ix);
        return fail(MIPS_E_HIP, "hipEventCreate failed");
    }
    // the sticky scan-error word: pinned host memory the device writes to (coherent, mapped)
    if (hipHostMalloc((void**)&ix->sticky_host, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&ix->sticky_dev, ix->sticky_host, 0) != hipSuccess) {
        mips_index_destroy(ix);
        return fail(MIPS_E_HIP, "hipHostMalloc for the scan-error word failed");
    }
    for (int w = 0; w < 16; ++w) ix->sticky_host[w] = 0u; // [0] scan error, [2..3] resolve statistics of the last stream-ordered search
    *out = ix;
    return MIPS_OK;
}

int mips_index_destroy(mips_index_t* ix) {
    if (!ix) return MIPS_OK;
    DeviceGuard g(ix->device);
    (void)hipDeviceSynchronize();
    if (ix->rows) (void)hipFree(ix->rows);
    ix->qbuf.release();
    ix->part_s.release();
    ix->part_i.release();
    ix->stage.release();
    ix->out_s.release();
    ix->out_i.release();
    ix->scalar.release();
    ix->gthr.release();
    ix->cand.release();
    ix->qf32.release();
    ix->mbnd.release();
    ix->mflag.release();
    ix->qbuf2.release();
    ix->qf32b.release();
    ix->tmp_s.release();
    ix->tmp_i.release();
    ix->ids.release();
    if (ix->xmax2_dev) (void)hipFree(ix->xmax2_dev);
    if (ix->nflag_host) (void)hipHostFree(ix->nflag_host);
    if (ix->tiny_words) (void)hipFree(ix->tiny_words);
    if (ix->rows_f32) (void)hipFree(ix->rows_f32);
    if (ix->rows_hi) (void)hipFree(ix->rows_hi);
    if (ix->dres2_dev) (void)hipFree(ix->dres2_dev);
    ix->qhi.release();
    ix->qerr2.release();
    ix->keyk.release();
    ix->qqv.release();
    ix->hit_d.release();
    ix->hit_i.release();
    ix->hit_n.release();
    ix->qnorm.release();
    for (int e = 0; e < mips_index::kEvRing; ++e) {
        if (ix->ev0[e]) (void)hipEventDestroy(ix->ev0[e]);
        if (ix->ev1[e]) (void)hipEventDestroy(ix->ev1[e]);
    }
    if (ix->busy) (void)hipEventDestroy(ix->busy);
    if (ix->scan_done) (void)hipEventDestroy(ix->scan_done);
    for (int e = 0; e < 2; ++e)
        if (ix->tail_done[e]) (void)hipEventDestroy(ix->tail_done[e]);
    ix->alt_qbuf.release();
    ix->alt_qf32.release();
    ix->alt_gthr.release();
    ix->alt_part_s.release();
    ix->alt_part_i.release();
    ix->alt_cand.release();
    ix->alt_mbnd.release();
    ix->alt_mflag.release();
    if (ix->sticky_host) (void)hipHostFree(ix->sticky_host);
    delete ix;
    return MIPS_OK;
}

int mips_index_reserve(mips_index_t* ix, int64_t n) {
    if (!ix || n < 0) return fail(MIPS_E_INVALID, "mips_index_reserve: bad argument");
    if (n > (int64_t)0x7fffff00) return fail(MIPS_E_UNSUPPORTED, "mips_index_reserve: more than 2^31 rows on one GPU");
    DeviceGuard g(ix->device);
    ORDER_ON(ix, nullptr);
    return grow(ix, n, nullptr, /*exact=*/true);
}

int mips_index_add(mips_index_t* ix, const void* rows, int64_t n, int src_dtype, int src_is_device, void* hip_stream) {
    if (!ix) return fail(MIPS_E_INVALID, "mips_index_add: index is NULL");
    if (n < 0 || (n > 0 && !rows)) return fail(MIPS_E_INVALID, "mips_index_add: bad rows / n");
    if (!src_dtype_ok(ix, src_dtype))
        return fail(MIPS_E_INVALID, "mips_index_add: src_dtype must be F32 or BF16 (or FP8_E4M3 bytes for an fp8 index)");
    if (n == 0) return MIPS_OK;
    if (ix->ntotal + n > (int64_t)0x7fffff00) return fail(MIPS_E_UNSUPPORTED, "mips_index_add: more than 2^31 rows on one GPU");
    DeviceGuard g(ix->device);
    hipStream_t st = (hipStream_t)hip_stream;
    ORDER_ON(ix, st);
    int rc = grow(ix, ix->ntotal + n, st);
    if (rc) return rc;
    rc = convert_into(ix, rows, n, src_dtype, src_is_device, ix->rows + (size_t)ix->ntotal * ix->ld * ix->esize, st,
                      ix->plane > 0 ? ix->rows_f32 + (size_t)ix->ntotal * ix->plane : nullptr);
    if (rc) return rc;
    ix->ntotal += n;
    if (!ix->phi_override) ix->phi_valid = false;
    ix->xmax2_valid = false;
    ix->dres2_valid = false;
    return MIPS_OK;
}

int mips_index_reset(mips_index_t* ix) {
    if (!ix) return fail(MIPS_E_INVALID, "mips_index_reset: index is NULL");
    ix->ntotal = 0;
    ix->hi_rows = 0;
    ix->phi_valid = false;
    ix->phi_override = false;
    ix->xmax2_valid = false;
    ix->dres2_valid = false;
    return MIPS_OK;
}

int64_t mips_index_ntotal(const mips_index_t* ix) { return ix ? ix->ntotal : -1; }
int64_t mips_index_dim(const mips_index_t* ix) { return ix ? ix->d : -1; }
int mips_index_metric(const mips_index_t* ix) { return ix ? ix->metric : -1; }

int mips_index_phi(mips_index_t* ix, double* out_phi, void* hip_stream) {
    if (!ix || !out_phi) return fail(MIPS_E_INVALID, "mips_index_phi: bad argument");
    DeviceGuard g(ix->device);
    ORDER_ON(ix, (hipStream_t)hip_stream);
    int rc = compute_phi(ix, (hipStream_t)hip_stream);
    if (rc) return rc;
    *out_phi = ix->phi;
    return MIPS_OK;
}

int mips_index_set_phi(mips_index_t* ix, double phi) {
    if (!ix || phi != phi) return fail(MIPS_E_INVALID, "mips_index_set_phi: bad argument");
    if (phi < 0.0) { // drop the override: the next mips_index_phi / L2 search recomputes the local maximum
        ix->phi_override = false;
        ix->phi_valid = false;
        return MIPS_OK;
    }
    ix->phi = phi;
    ix->phi_valid = true;
    ix->phi_override = true;
    return MIPS_OK;
}

int mips_index_read_rows(mips_index_t* ix, int64_t row0, int64_t n, void* out_host_u16, void* hip_stream) {
    if (!ix || row0 < 0 || n < 0 || row0 + n > ix->ntotal || (n > 0 && !out_host_u16))
        return fail(MIPS_E_INVALID, "mips_index_read_rows: bad range [%lld, +%lld) of %lld", (long long)row0, (long long)n,
                    ix ? (long long)ix->ntotal : -1LL);
    if (n == 0) return MIPS_OK;
    DeviceGuard g(ix->device);
    hipStream_t st = (hipStream_t)hip_stream;
    ORDER_ON(ix, st);
    if (ix->plane > 0) { // fp32-exact mode: the fp32 originals
        HIP_TRY(hipMemcpy2DAsync(out_host_u16, (size_t)ix->d * 4, ix->rows_f32 + (size_t)row0 * ix->plane, (size_t)ix->plane * 4,
                                 (size_t)ix->d * 4, (size_t)n, hipMemcpyDeviceToHost, st));
    } else {
        const size_t es = (size_t)ix->esize;
        HIP_TRY(hipMemcpy2DAsync(out_host_u16, (size_t)ix->d * es, ix->rows + (size_t)row0 * ix->ld * es, (size_t)ix->ld * es,
                                 (size_t)ix->d * es, (size_t)n, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return MIPS_OK;
}

int mips_index_add_synthetic(mips_index_t* ix, int64_t n, int64_t row0, uint64_t seed, int kind, void* hip_stream) {
    if (!ix || n < 0 || row0 < 0) return fail(MIPS_E_INVALID, "mips_index_add_synthetic: bad argument");
    if (kind < 0 || kind > 2) return fail(MIPS_E_INVALID, "mips_index_add_synthetic: unknown kind %d", kind);
    if (n == 0) return MIPS_OK;
    if (ix->ntotal + n > (int64_t)0x7fffff00) return fail(MIPS_E_UNSUPPORTED, "more than 2^31 rows on one GPU");
    DeviceGuard g(ix->device);
    hipStream_t st = (hipStream_t)hip_stream;
    ORDER_ON(ix, st);
    int rc = grow(ix, ix->ntotal + n, st);
    if (rc) return rc;
    if (ix->plane > 0) {
        float* f32 = ix->rows_f32 + (size_t)ix->ntotal * ix->plane;
        const int64_t items = n * (ix->plane / 8);
        mips::synth_fill_kernel<<<grid_for(items, 256), 256, 0, st>>>(f32, n, (int)ix->d, ix->plane, row0, seed, kind, 1);
        mips::split_rows_kernel<float><<<grid_for(items, 256), 256, 0, st>>>(
            f32, n, (int)ix->d, ix->plane, (uint16_t*)(ix->rows + (size_t)ix->ntotal * ix->ld * 2), ix->plane, nullptr);
    } else {
        const int64_t items = n * (ix->ld / 8);
        mips::synth_fill_kernel<<<grid_for(items, 256), 256, 0, st>>>(ix->rows + (size_t)ix->ntotal * ix->ld * ix->esize, n,
                                                                      (int)ix->d, ix->ld, row0, seed, kind, ix->esize == 1 ? 2 : 0);
    }
    HIP_TRY(hipGetLastError());
    ix->ntotal += n;
    if (!ix->phi_override) ix->phi_valid = false; // as mips_index_add: an override stays until the caller renews it
    ix->xmax2_valid = false;
    ix->dres2_valid = false;
    return MIPS_OK;
}

int mips_synth_fill(void* out_device, int64_t n, int64_t d, int64_t row0, uint64_t seed, int kind, int dtype, int device,
                    void* hip_stream) {
    if (!out_device || n < 0 || d <= 0 || d % 8 != 0)
        return fail(MIPS_E_INVALID, "mips_synth_fill: bad argument (d must be a positive multiple of 8)");
    if (kind < 0 || kind > 2) return fail(MIPS_E_INVALID, "mips_synth_fill: unknown kind %d", kind);
    if (dtype != MIPS_DTYPE_F32 && dtype != MIPS_DTYPE_BF16 && dtype != MIPS_DTYPE_FP8_E4M3)
        return fail(MIPS_E_INVALID, "mips_synth_fill: dtype must be F32, BF16 or FP8_E4M3");
    if (n == 0) return MIPS_OK;
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    const int64_t items = n * (d / 8);
    mips::synth_fill_kernel<<<grid_for(items, 256), 256, 0, (hipStream_t)hip_stream>>>(out_device, n, (int)d, (int)d, row0, seed,
                                                                                      kind, dtype == MIPS_DTYPE_F32 ? 1 : dtype == MIPS_DTYPE_FP8_E4M3 ? 2 : 0);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

static int search_impl(mips_index_t* ix, const void* q, int q_dtype, int64_t nq, int k, float* out_scores, int64_t* out_idx,
                       int64_t idx_offset, int flags, void* hip_stream, void* tail_stream);

int mips_search(mips_index_t* ix, const void* q, int q_dtype, int64_t nq, int k, float* out_scores, int64_t* out_idx,
                int64_t idx_offset, int flags, void* hip_stream) {
    return search_impl(ix, q, q_dtype, nq, k, out_scores, out_idx, idx_offset, flags, hip_stream, hip_stream);
}

int mips_search_split(mips_index_t* ix, const void* q, int q_dtype, int64_t nq, int k, float* out_scores, int64_t* out_idx,
                      int64_t idx_offset, int flags, void* scan_stream, void* tail_stream) {
    if (!(flags & MIPS_OUT_DEVICE)) return fail(MIPS_E_INVALID, "mips_search_split: device outputs only (MIPS_OUT_DEVICE)");
    return search_impl(ix, q, q_dtype, nq, k, out_scores, out_idx, idx_offset, flags, scan_stream, tail_stream);
}

static int search_impl(mips_index_t* ix, const void* q, int q_dtype, int64_t nq, int k, float* out_scores, int64_t* out_idx,
                       int64_t idx_offset, int flags, void* hip_stream, void* tail_stream) {
    if (!ix) return fail(MIPS_E_INVALID, "mips_search: index is NULL");
    if (nq < 0 || k < 0) return fail(MIPS_E_INVALID, "mips_search: negative nq or k");
    if (k > MIPS_MAX_K) return fail(MIPS_E_UNSUPPORTED, "mips_search: k = %d exceeds MIPS_MAX_K = %d", k, MIPS_MAX_K);
    if (!src_dtype_ok(ix, q_dtype, true)) return fail(MIPS_E_INVALID, "mips_search: q_dtype must be F32 or BF16 (or FP8_E4M3 bytes for an all-e4m3 index)");
    if (nq == 0 || k == 0) return MIPS_OK;
    if (!q || !out_idx || (!out_scores && !(flags & MIPS_OUT_PACKED))) return fail(MIPS_E_INVALID, "mips_search: NULL buffer");
    if (nq > (1 << 24)) return fail(MIPS_E_UNSUPPORTED, "mips_search: more than 2^24 queries in one call");
    DeviceGuard g(ix->device);
    hipStream_t st = (hipStream_t)hip_stream;
    {   // an earlier device-output search on this index whose scan timed out: report it now (no synchronisation:
        // the word lives in host memory), before anything new is enqueued
        const int prev = take_scan_error(ix, "mips_search");
        if (prev) return prev;
    }
    ORDER_ON(ix, st);
    const bool out_dev = (flags & MIPS_OUT_DEVICE) != 0;
    const bool packed = (flags & MIPS_OUT_PACKED) != 0;
    ix->call_metric = (flags & MIPS_FORCE_IP) ? MIPS_METRIC_IP : ix->metric;
    if (packed && !out_dev) return fail(MIPS_E_INVALID, "mips_search: MIPS_OUT_PACKED requires MIPS_OUT_DEVICE");
    // split-tail form: the other scratch set, so that this scan may run while the previous search's tail still reads
    // its lists; whichever set is about to be used, a tail still pending on it comes first
    hipStream_t tail_st = (hipStream_t)tail_stream;
    const bool split = tail_st != st;
    if (split) swap_scratch_sets(ix);
    if (ix->tail_pending[ix->cur_set]) {
        HIP_TRY(hipStreamWaitEvent(st, ix->tail_done[ix->cur_set], 0));
        ix->tail_pending[ix->cur_set] = false;
    }

    float* d_s = out_scores;
    int64_t* d_i = out_idx;
    if (!out_dev) {
        int rc = ix->out_s.ensure((size_t)nq * k * sizeof(float));
        if (rc) return rc;
        rc = ix->out_i.ensure((size_t)nq * k * sizeof(int64_t));
        if (rc) return rc;
        d_s = (float*)ix->out_s.p;
        d_i = (int64_t*)ix->out_i.p;
    }

    MarginScope margin_scope(ix, out_dev);
    const bool certifies = ix->opt_margin != 0 && (!out_dev || ix->opt_margin >= 2);
    bool done = false;
    if (ix->ntotal == 0) {
        const int64_t total = nq * k;
        mips::fill_empty_kernel<<<(int)((total + 255) / 256), 256, 0, st>>>(d_s, d_i, packed ? d_i : nullptr, total, ix->call_metric);
        HIP_TRY(hipGetLastError());
        done = true;
    } else if (tiny_eligible(ix, nq, k, certifies)) {
        // the reference's own call shape (<= 16 queries, small knowledge base): one launch (tiny_search.hpp)
        if (ix->call_metric == MIPS_METRIC_L2) {
            int rc = compute_phi(ix, st);
            if (rc) return rc;
        }
        const void* qd = q;
        if (!(flags & MIPS_Q_DEVICE)) {
            const size_t qbytes = (size_t)nq * ix->d * (q_dtype == MIPS_DTYPE_F32 ? 4 : 2);
            int rc = ix->stage.ensure(qbytes);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(ix->stage.p, q, qbytes, hipMemcpyHostToDevice, st));
            qd = ix->stage.p;
        }
        int rc = tiny_search(ix, qd, q_dtype, nq, k, k, 0, nullptr, d_s, d_i, packed, idx_offset, st, certifies);
        if (rc) return rc;
        done = true;
        if (certifies) {
            // the exact pass behind the one launch (the kernel, and with it its results, live on the scan stream even in the
            // split-tail form).  Device outputs: enqueued blind, it leaves at once when nothing was flagged; host buffers and
            // "margin_check" = 2 synchronise, read the count and skip it when that is 0
            rc = resolve_flagged(ix, nq, k, d_s, d_i, packed, idx_offset, st, /*certify_now=*/!(out_dev && ix->opt_margin == 3), /*skip_compact=*/true);
            if (rc < 0) return rc; // (kUseRescan cannot happen: <= 16 queries)
        }
    }
    if (!done) {
        if (ix->call_metric == MIPS_METRIC_L2) {
            int rc = compute_phi(ix, st);
            if (rc) return rc;
        }
        const int64_t nq_pad = query_pad(ix, nq);
        const size_t row_bytes = (size_t)ix->ld * ix->qsize;
        int rc = ix->qbuf.ensure((size_t)nq_pad * row_bytes);
        if (rc) return rc;
        uint8_t* qb = (uint8_t*)ix->qbuf.p;
        float* qkeep = nullptr;
        if (ix->plane > 0) {
            rc = ix->qf32.ensure((size_t)nq_pad * ix->plane * sizeof(float));
            if (rc) return rc;
            qkeep = (float*)ix->qf32.p;
        }
        // shared insert bounds of the scan: 8 class words per query + the error word
        const int64_t thr_words = nq_pad * 8 + 4;
        rc = ix->gthr.ensure((size_t)thr_words * sizeof(unsigned));
        if (rc) return rc;
        if (ix->plane > 0) { // fp32-exact staging has its own kernel: clear with plain memsets
            rc = convert_into(ix, q, nq, q_dtype, (flags & MIPS_Q_DEVICE) ? 1 : 0, qb, st, qkeep);
            if (rc) return rc;
            if (nq_pad > nq) HIP_TRY(hipMemsetAsync(qb + (size_t)nq * row_bytes, 0, (size_t)(nq_pad - nq) * row_bytes, st));
            HIP_TRY(hipMemsetAsync(ix->gthr.p, 0, (size_t)thr_words * sizeof(unsigned), st));
        } else { // one launch converts the queries, zero-pads to the tile multiple and clears the bounds
            rc = convert_into(ix, q, nq, q_dtype, (flags & MIPS_Q_DEVICE) ? 1 : 0, qb, st, qkeep, nq_pad - nq,
                              (uint32_t*)ix->gthr.p, thr_words, ix->qsize);
            if (rc) return rc;
        }
        // The MFMA scores only SELECT a pool of K' candidates that is then re-scored exactly.
        // Pool K' = 8 / 10 / 16 / 32 >= k + 3.  What is GUARANTEED to reach the pool per (query, split): the 32x32
        // kernels (scan_kernel_v3 / f8 / generic) keep lists of K' entries, so MFMA ranks 1 .. K'; the 16x16 kernels
        // (scan_kernel_v4 / f8x: bf16 pitch 384 .. 768 or fp8 pitch <= 768, k <= 5, more than one query tile) keep 4
        // sub-lists of 6, so MFMA ranks 1 .. 6 = k + 1 for certain and 7 .. 8 unless 6 better documents share the
        // sub-list (rows congruent mod 16 within a split).  Queries whose k-th exact score is too close to what the
        // pool may have lost are detected by the re-score and re-scanned with the widest lists (margin check below).
        // fp32-exact index: two-stage search when the call may synchronise anyway (see mips_index: rows_hi).  Stage 1 keeps
        // K' = 32 lists where the bf16 kernels have them (pitch 256 / 512 / 768): the pool's bound is then the ~33rd best
        // score, far enough below the k-th for the widened margin to certify nearly every query on well-separated data
        // (with K' = 8 pools 44 % of the queries of a Gaussian test set went to the second stage; pitch 1024 has no more).
        const bool hi_long = ix->hp > 0 && ix->hp <= 1024;
        // (round 3: row pitch 1024 has true K' = 32 lists as well -- with pools of 8, all it had before, 46 % of a Gaussian
        // test set went to the second stage and stage 1 did not pay)
        // k <= 13 only: the pool is 32 wide whatever k, and the widened margin (representation error of bf16(x) . bf16(q)) has to
        // fit between the k-th and the 33rd score -- at k = 20 .. 29 nearly every query of a Gaussian test set was flagged, and an
        // exact pass per 8 flagged queries costs far more than the three-segment scan saves
        bool fast = ix->plane > 0 && ix->hp > 0 && hi_long && k <= 13 && ix->opt_f32_fast != 0 && ix->opt_margin != 0 && !split &&
                    (ix->opt_f32_fast == 2 || !out_dev || ix->opt_margin >= 2);
        if (fast && ix->opt_f32_fast == 1 && ix->fast_skip > 0) {
            --ix->fast_skip;
            fast = false;
        }
        if (fast) { // bf16(q) at the bf16 kernels' row pitch, |q - bf16 q|^2, bf16 rows and residual bound up to date
            rc = ix->qhi.ensure((size_t)nq_pad * ix->hp * 2);
            if (rc) return rc;
            rc = ix->qerr2.ensure((size_t)nq * sizeof(double));
            if (rc) return rc;
            mips::convert_rows_kernel<float><<<grid_for(nq_pad * (int64_t)(ix->hp / 8), 256), 256, 0, st>>>((const float*)ix->qf32.p, nq, ix->plane,
                                                                                                           (uint16_t*)ix->qhi.p, ix->hp, nq_pad);
            mips::query_resid_kernel<<<(int)((nq + 3) / 4), 256, 0, st>>>((const float*)ix->qf32.p, nq, ix->plane, (double*)ix->qerr2.p);
            HIP_TRY(hipGetLastError());
            rc = ensure_hi(ix, st);
            if (rc) return rc;
            rc = ensure_xmax2(ix, st); // (on the fp32 rows: before the index is viewed as rows_hi)
            if (rc) return rc;
        }
        // (pools of 32, not 16: with the 16th best score as the bound 30 of 4096 Gaussian queries went to the second stage, and
        // a second stage of even one query tile costs more than stage 1 saved -- 22 ms per call instead of 5.4)
        // bf16 rows at pitch 1024, k <= 5, a large index, a call that certifies: pool of 16 (scan_kernel_k3 with every sub-list
        // vouching for its 2nd best).  The MFMA error bound at K = 1024 (0.14 on unit-variance data) reaches from the 5th exact score
        // to the 8th best approximate one for one Gaussian query in ~3000 at 2^22 rows, and each flagged query costs a pass over the
        // index (3 ms there); the 16th best is out of reach.  Small indexes keep the pool of 8 (their exact pass is cheap).
        bool deep_1024 = !fast && k <= 5 && ix->plane == 0 && ix->esize == 2 && !ix->mixed && ix->ld == 1024 && nq > 256 && ix->ntotal >= (1ll << 21) &&
                         ix->opt_margin != 0 && !split && (!out_dev || ix->opt_margin >= 2) && ix->opt_f32_fast != 0;
        if (deep_1024 && ix->fast_skip > 0) {
            --ix->fast_skip;
            deep_1024 = false;
        }
        if (fast && hi_long) {
            rc = scan_and_finish<32>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, tail_st, split, true);
        } else if (k <= 5 && deep_1024) {
            rc = scan_and_finish<16>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, tail_st, split, false, true);
        } else if (k <= 5) {
            rc = scan_and_finish<8>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, tail_st, split, fast);
        } else if (k <= 7) { // k + 1 = 6 is what Mips.search fetches for top_k = 5 with ignore_indexes (mips.py:388-398)
            rc = scan_and_finish<10>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, tail_st, split, fast);
        } else {
            // bf16 index, a call that certifies, 8 <= k <= 29: pool of 32 out of the 16x16x32 kernel's sub-lists (see
            // mips_index::optimistic; round 3: k = 14 .. 29 as well -- the class words vouch for 32 documents, what the
            // sub-lists of 6 may have dropped is bounded, the margin check decides per query).  Otherwise true K' = 16 / 32 lists.
            bool opt = ix->plane == 0 && ix->esize == 2 && !ix->mixed && ix->opt_margin != 0 && !split && (!out_dev || ix->opt_margin >= 2) &&
                       ix->opt_f32_fast != 0 && ((ix->ld % 128 == 0 && ix->ld >= 384 && ix->ld <= 768) || (ix->ld == 1024 && nq > 256));
            if (opt && ix->fast_skip > 0) {
                --ix->fast_skip;
                opt = false;
            }
            if (opt) rc = scan_and_finish<32>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, tail_st, split, false, true);
            else if (k <= 13) rc = scan_and_finish<16>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, tail_st, split, false);
            else rc = scan_and_finish<32>(ix, nq, k, d_s, d_i, packed, idx_offset, out_dev, st, tail_st, split, false);
        }
        if (rc) return rc;
    }
    if (!out_dev) {
        HIP_TRY(hipMemcpyAsync(out_scores, d_s, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(out_idx, d_i, (size_t)nq * k * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // host buffers: the stream is drained, so a timed-out scan of THIS call is known now
        const int bad = take_scan_error(ix, "mips_search");
        if (bad) return bad;
    }
    return MIPS_OK;
}

int mips_search_fused(mips_index_t* ix, const void* q_device, int q_dtype, int64_t nq, int k, int normalize, const int64_t* ignore_device,
                      float* out_scores_device, int64_t* out_idx_device, int64_t idx_offset, void* hip_stream) {
    if (!ix) return fail(MIPS_E_INVALID, "mips_search_fused: index is NULL");
    if (nq < 0 || k < 0) return fail(MIPS_E_INVALID, "mips_search_fused: negative nq or k");
    const int k_fetch = k + (ignore_device ? 1 : 0);
    if (k_fetch > MIPS_MAX_K) return fail(MIPS_E_UNSUPPORTED, "mips_search_fused: k = %d exceeds MIPS_MAX_K = %d", k_fetch, MIPS_MAX_K);
    if (!src_dtype_ok(ix, q_dtype, true)) return fail(MIPS_E_INVALID, "mips_search_fused: q_dtype must be F32 or BF16");
    if (normalize && q_dtype != MIPS_DTYPE_F32) return fail(MIPS_E_INVALID, "mips_search_fused: normalize needs float32 queries");
    if (nq == 0 || k == 0) return MIPS_OK;
    if (!q_device || !out_idx_device || !out_scores_device) return fail(MIPS_E_INVALID, "mips_search_fused: NULL buffer");
    DeviceGuard g(ix->device);
    hipStream_t st = (hipStream_t)hip_stream;
    {   // (both forms below touch the index's scratch: an earlier scan error is reported, and a call on another stream ordered,
        // before anything is enqueued)
        const int prev = take_scan_error(ix, "mips_search_fused");
        if (prev) return prev;
    }
    ORDER_ON(ix, st);
    {
        MarginScope margin_scope(ix, /*out_dev=*/true);
        const bool certifies = ix->opt_margin >= 2;
        if (tiny_eligible(ix, nq, k_fetch, certifies)) {
            ix->call_metric = ix->metric;
            if (ix->call_metric == MIPS_METRIC_L2) {
                int rc = compute_phi(ix, st);
                if (rc) return rc;
            }
            int rc = tiny_search(ix, q_device, q_dtype, nq, k_fetch, k, normalize, ignore_device, out_scores_device, out_idx_device, false,
                                 idx_offset, st, certifies);
            if (rc) return rc;
            if (!certifies) return MIPS_OK; // ("margin_check" = 4 / 0: flagged queries are counted, or not even that)
            // certified: the exact pass behind the one launch ranks the hits of a flagged query and applies the same ignore
            // filter -- without a synchronisation by default (it leaves at once when nothing was flagged); "margin_check" = 2
            // synchronises to read the counts
            rc = resolve_flagged(ix, nq, k_fetch, out_scores_device, out_idx_device, false, idx_offset, st, /*certify_now=*/ix->opt_margin == 2,
                                 /*skip_compact=*/true, ignore_device, k);
            return rc < 0 ? rc : MIPS_OK;
        }
    }
    // general form: the same steps as separate launches
    const void* qsrc = q_device;
    if (normalize) { // (a buffer of its own: the searches' scratch -- qf32 / qf32b -- is rewritten by the nested call)
        const size_t qbytes = (size_t)nq * ix->d * sizeof(float);
        int rc = ix->qnorm.ensure(qbytes);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(ix->qnorm.p, q_device, qbytes, hipMemcpyDeviceToDevice, st));
        rc = mips_l2_normalize((float*)ix->qnorm.p, nq, ix->d, ix->device, hip_stream);
        if (rc) return rc;
        qsrc = ix->qnorm.p;
    }
    if (!ignore_device)
        return mips_search(ix, qsrc, q_dtype, nq, k, out_scores_device, out_idx_device, idx_offset, MIPS_Q_DEVICE | MIPS_OUT_DEVICE, hip_stream);
    int rc = ix->out_s.ensure((size_t)nq * k_fetch * sizeof(float));
    if (rc) return rc;
    rc = ix->out_i.ensure((size_t)nq * k_fetch * sizeof(int64_t));
    if (rc) return rc;
    rc = mips_search(ix, qsrc, q_dtype, nq, k_fetch, (float*)ix->out_s.p, (int64_t*)ix->out_i.p, idx_offset, MIPS_Q_DEVICE | MIPS_OUT_DEVICE,
                     hip_stream);
    if (rc) return rc;
    return mips_filter_ignore((const float*)ix->out_s.p, (const int64_t*)ix->out_i.p, ignore_device, nq, k_fetch, k, out_scores_device,
                              out_idx_device, ix->device, hip_stream);
}

int mips_merge_topk(const float* cand_s, const int64_t* cand_i, int64_t nq, int parts, int k, int metric, float* out_s,
                    int64_t* out_i, int device, void* hip_stream) {
    if (nq < 0 || parts <= 0 || k < 0) return fail(MIPS_E_INVALID, "mips_merge_topk: bad sizes");
    if (nq == 0 || k == 0) return MIPS_OK;
    if (!cand_s || !cand_i || !out_s || !out_i) return fail(MIPS_E_INVALID, "mips_merge_topk: NULL buffer");
    if ((int64_t)parts * k > 65536) return fail(MIPS_E_UNSUPPORTED, "mips_merge_topk: parts * k too large");
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    mips::merge_topk_kernel<<<(int)nq, 64, 0, (hipStream_t)hip_stream>>>(cand_s, cand_i, parts * k, k, metric, out_s, out_i);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

int mips_merge_topk_packed(const int64_t* gathered, int64_t nq, int parts, int k, int metric, float* out_s,
                           int64_t* out_i, int device, void* hip_stream) {
    if (nq < 0 || parts <= 0 || k < 0) return fail(MIPS_E_INVALID, "mips_merge_topk_packed: bad sizes");
    if (nq == 0 || k == 0) return MIPS_OK;
    if (!gathered || !out_s || !out_i) return fail(MIPS_E_INVALID, "mips_merge_topk_packed: NULL buffer");
    if ((int64_t)parts * k > 65536) return fail(MIPS_E_UNSUPPORTED, "mips_merge_topk_packed: parts * k too large");
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    mips::merge_topk_packed_kernel<<<(int)nq, 64, 0, (hipStream_t)hip_stream>>>(gathered, nq, parts, k, metric, out_s, out_i);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

int mips_filter_ignore(const float* scores, const int64_t* idx, const int64_t* ignore, int64_t nq, int k_fetched, int k,
                       float* out_s, int64_t* out_i, int device, void* hip_stream) {
    if (nq < 0 || k < 0 || k_fetched < k) return fail(MIPS_E_INVALID, "mips_filter_ignore: bad sizes");
    if (nq == 0 || k == 0) return MIPS_OK;
    if (!scores || !idx || !ignore || !out_s || !out_i) return fail(MIPS_E_INVALID, "mips_filter_ignore: NULL buffer");
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    mips::filter_ignore_kernel<<<(int)((nq + 255) / 256), 256, 0, (hipStream_t)hip_stream>>>(scores, idx, ignore, nq, k_fetched,
                                                                                          k, out_s, out_i);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

static int cosine_rescore_impl(const char* who, const void* query, const void* cls, int dtype, int64_t b, int k, int64_t d,
                               float* out, int64_t mem_len, float* bias, int device, void* hip_stream) {
    if (b < 0 || k < 0 || d <= 0 || mem_len < 0) return fail(MIPS_E_INVALID, "%s: bad sizes", who);
    if (dtype != MIPS_DTYPE_F32 && dtype != MIPS_DTYPE_BF16) return fail(MIPS_E_INVALID, "%s: dtype must be F32 or BF16", who);
    if (b == 0 || k == 0) return MIPS_OK;
    if (!query || !cls || !out || (mem_len > 0 && !bias)) return fail(MIPS_E_INVALID, "%s: NULL buffer", who);
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    const int64_t pairs = b * k;
    const int grid = (int)((pairs + 3) / 4);
    float* bias_arg = mem_len > 0 ? bias : nullptr;
    if (dtype == MIPS_DTYPE_F32)
        mips::cosine_rescore_kernel<float><<<grid, 256, 0, (hipStream_t)hip_stream>>>((const float*)query, (const float*)cls, pairs, k, (int)d, out, mem_len, bias_arg);
    else
        mips::cosine_rescore_kernel<uint16_t><<<grid, 256, 0, (hipStream_t)hip_stream>>>((const uint16_t*)query, (const uint16_t*)cls, pairs, k, (int)d, out, mem_len, bias_arg);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

int mips_cosine_rescore(const void* query, const void* cls, int dtype, int64_t b, int k, int64_t d, float* out, int device,
                        void* hip_stream) {
    return cosine_rescore_impl("mips_cosine_rescore", query, cls, dtype, b, k, d, out, 0, nullptr, device, hip_stream);
}

int mips_cosine_rescore_bias(const void* query, const void* cls, int dtype, int64_t b, int k, int64_t d, float* out,
                             int64_t memory_seq_len, float* memory_bias, int device, void* hip_stream) {
    return cosine_rescore_impl("mips_cosine_rescore_bias", query, cls, dtype, b, k, d, out, memory_seq_len, memory_bias, device,
                               hip_stream);
}

int mips_cosine_rescore_backward(const void* query, const void* cls, int dtype, int64_t b, int k, int64_t d, const float* grad_scores,
                                 const float* grad_memory_bias, int64_t memory_seq_len, float* grad_query, float* grad_cls, int device,
                                 void* hip_stream) {
    if (b < 0 || k < 0 || d <= 0 || memory_seq_len < 0) return fail(MIPS_E_INVALID, "mips_cosine_rescore_backward: bad sizes");
    if (k > 64) return fail(MIPS_E_UNSUPPORTED, "mips_cosine_rescore_backward: k = %d > 64", k);
    if (dtype != MIPS_DTYPE_F32 && dtype != MIPS_DTYPE_BF16) return fail(MIPS_E_INVALID, "mips_cosine_rescore_backward: dtype must be F32 or BF16");
    if (b == 0 || k == 0) return MIPS_OK;
    if (!query || !cls || !grad_query || !grad_cls || (!grad_scores && !grad_memory_bias))
        return fail(MIPS_E_INVALID, "mips_cosine_rescore_backward: NULL buffer");
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    const float* gb = memory_seq_len > 0 ? grad_memory_bias : nullptr;
    if (dtype == MIPS_DTYPE_F32)
        mips::cosine_rescore_bwd_kernel<float><<<(int)b, 256, 0, (hipStream_t)hip_stream>>>((const float*)query, (const float*)cls, k, (int)d, grad_scores, gb, memory_seq_len, grad_query, grad_cls);
    else
        mips::cosine_rescore_bwd_kernel<uint16_t><<<(int)b, 256, 0, (hipStream_t)hip_stream>>>((const uint16_t*)query, (const uint16_t*)cls, k, (int)d, grad_scores, gb, memory_seq_len, grad_query, grad_cls);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

int mips_l2_normalize(float* x_device, int64_t n, int64_t d, int device, void* hip_stream) {
    if (n < 0 || d <= 0 || (n > 0 && !x_device)) return fail(MIPS_E_INVALID, "mips_l2_normalize: bad argument");
    if (n == 0) return MIPS_OK;
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    mips::l2_normalize_kernel<<<(int)((n + 3) / 4), 256, 0, (hipStream_t)hip_stream>>>(x_device, n, (int)d);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

int mips_rows_max_sumsq(const float* x_device, int64_t n, int64_t d, double* out_host, int device, void* hip_stream) {
    if (n < 0 || d <= 0 || (n > 0 && !x_device) || !out_host) return fail(MIPS_E_INVALID, "mips_rows_max_sumsq: bad argument");
    *out_host = 0.0;
    if (n == 0) return MIPS_OK;
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    hipStream_t st = (hipStream_t)hip_stream;
    // one 8-byte result slot per (host thread, device), allocated once: the rebuild path calls this per build
    // (mips.py:298-304), and a hipMalloc / hipFree pair per call serialises the device
    constexpr int kMaxDev = 64;
    thread_local unsigned long long* slots[kMaxDev] = {};
    if (device < 0 || device >= kMaxDev) return fail(MIPS_E_INVALID, "mips_rows_max_sumsq: device %d out of range", device);
    if (!slots[device]) HIP_TRY(hipMalloc((void**)&slots[device], 8));
    unsigned long long* slot = slots[device];
    hipError_t e = hipMemsetAsync(slot, 0, 8, st);
    if (e == hipSuccess) {
        mips::f32_rows_max_sumsq_kernel<<<(int)std::min<int64_t>((n + 3) / 4, 256 * 16), 256, 0, st>>>(x_device, n, (int)d, slot);
        e = hipGetLastError();
    }
    unsigned long long bits = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bits, slot, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(MIPS_E_HIP, "mips_rows_max_sumsq: %s", hipGetErrorString(e));
    std::memcpy(out_host, &bits, 8);
    return MIPS_OK;
}

int mips_rows_max_sumsq_device(const float* x_device, int64_t n, int64_t d, double* acc_device, int device, void* hip_stream) {
    if (n < 0 || d <= 0 || (n > 0 && !x_device) || !acc_device) return fail(MIPS_E_INVALID, "mips_rows_max_sumsq_device: bad argument");
    if (n == 0) return MIPS_OK;
    DeviceGuard g(device);
    if (!g.ok) return fail(MIPS_E_HIP, "hipSetDevice(%d) failed", device);
    // (non-negative doubles order like their bit patterns: the kernel's atomicMax on the 64-bit word IS the running maximum)
    mips::f32_rows_max_sumsq_kernel<<<(int)std::min<int64_t>((n + 3) / 4, 256 * 16), 256, 0, (hipStream_t)hip_stream>>>(
        x_device, n, (int)d, (unsigned long long*)acc_device);
    HIP_TRY(hipGetLastError());
    return MIPS_OK;
}

int mips_index_set_param(mips_index_t* ix, const char* name, int64_t value) {
    if (!ix || !name) return fail(MIPS_E_INVALID, "mips_index_set_param: bad argument");
    const std::string n(name);
    if (n == "nsplit") ix->opt_nsplit = (int)value;
    else if (n == "qgroups") ix->opt_qgroups = (int)value;
    else if (n == "variant") {
        ix->opt_variant = (int)value;
#ifndef MIPS_EXPERIMENTAL
        if (value == 5 || value == 6) ix->opt_variant = 0; // (kernels of the A/B build only: the automatic choice answers)
#endif
    }
    else if (n == "tiny") ix->opt_tiny = value == 2 ? 2 : value != 0 ? 1 : 0; // 2: one launch, fall-back paths forced (tests)
    else if (n == "resolve") ix->opt_resolve = value < 0 ? 0 : value > 2 ? 2 : (int)value; // 2: exact pass without the MFMA pre-filter
    else if (n == "f32_fast" || n == "optimistic") { // (one switch: two-stage fp32 search and optimistic pools, include/mips_hip.h)
        if (value < 0 || value > 2) return fail(MIPS_E_INVALID, "mips_index_set_param: f32_fast / optimistic must be 0, 1 or 2");
        ix->opt_f32_fast = (int)value;
        ix->fast_skip = 0;
    } else if (n == "margin_check") {
        if (value < 0 || value > 4) return fail(MIPS_E_INVALID, "mips_index_set_param: margin_check must be 0, 1, 2, 3 or 4");
        ix->opt_margin = (int)value;
    } else if (n == "resolve_budget") {
        if (value < 0) return fail(MIPS_E_INVALID, "mips_index_set_param: resolve_budget must be >= 0");
        ix->resolve_budget = (int)std::min<int64_t>(value, mips::RESOLVE_MAX);
    } else if (n == "spin_limit") ix->opt_spin_limit = (int)std::max<int64_t>(-1, std::min<int64_t>(value, 1 << 30));
    else if (n == "sub") {
#ifdef MIPS_EXPERIMENTAL
        ix->opt_sub = (int)value;
#else
        if (value != 0)
            return fail(MIPS_E_UNSUPPORTED, "mips_index_set_param: 'sub' selects experimental kernel instances that the shipped "
                                            "library does not contain (build with -DMIPS_EXPERIMENTAL, tools/ab.py)");
#endif
    } else return fail(MIPS_E_INVALID, "mips_index_set_param: unknown parameter '%s'", name);
    return MIPS_OK;
}

int mips_index_check_error(mips_index_t* ix, int synchronize, void* hip_stream) {
    if (!ix) return fail(MIPS_E_INVALID, "mips_index_check_error: index is NULL");
    if (synchronize) {
        DeviceGuard g(ix->device);
        HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
        for (int e = 0; e < 2; ++e) // split-tail searches raise the sticky word from their tail stream
            if (ix->tail_pending[e]) HIP_TRY(hipEventSynchronize(ix->tail_done[e]));
    }
    return take_scan_error(ix, "mips_index_check_error");
}

const char* mips_index_last_kernel(const mips_index_t* ix) { return ix ? ix->last_kernel : ""; }

int mips_index_margin_stats(mips_index_t* ix, int64_t* flagged, int64_t* rescanned, int64_t* unresolved, int synchronize,
                            void* hip_stream) {
    if (!ix) return fail(MIPS_E_INVALID, "mips_index_margin_stats: index is NULL");
    if (synchronize && ix->tail_pending[ix->cur_set]) { // a split-tail search writes its counters on the tail stream
        DeviceGuard g(ix->device);
        HIP_TRY(hipEventSynchronize(ix->tail_done[ix->cur_set]));
    }
    if (ix->last_flagged < 0 && synchronize && ix->first_nflag_dev != nullptr) {
        // the last search re-scanned its flagged queries on the stream: first count = flagged = re-scanned, second = unresolved
        DeviceGuard g(ix->device);
        unsigned n[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(&n[0], ix->first_nflag_dev, 4, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
        HIP_TRY(hipMemcpyAsync(&n[1], ix->last_nflag_dev, 4, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
        ix->last_flagged = (int64_t)n[0];
        ix->last_unresolved = (int64_t)n[1];
        // settled exactly (or re-scanned by the fall-back); a search over its budget whose first results stand settled nothing
        ix->last_rescanned = (ix->last_max_n > 0 && (int64_t)n[0] > ix->last_max_n && !ix->last_fallback) ? 0 : (int64_t)n[0];
    } else if (ix->last_flagged < 0 && synchronize && ix->opt_margin != 0 && ix->last_nflag_dev != nullptr) {
        // the last search only counted on the device: fetch the count now
        DeviceGuard g(ix->device);
        unsigned n = 0;
        HIP_TRY(hipMemcpyAsync(&n, ix->last_nflag_dev, 4, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
        ix->last_flagged = (int64_t)n;
        ix->last_rescanned = 0;
        ix->last_unresolved = (int64_t)n;
    }
    if (flagged) *flagged = ix->last_flagged;
    if (rescanned) *rescanned = ix->last_rescanned;
    if (unresolved) *unresolved = ix->last_unresolved;
    return MIPS_OK;
}

int mips_scan_timing(mips_index_t* ix, float* out_sum_ms, int* out_count, int reset) {
    if (!ix) return fail(MIPS_E_INVALID, "mips_scan_timing: index is NULL");
    DeviceGuard g(ix->device);
    float sum = 0.f;
    int n = 0;
    for (int t = 0; t < ix->ev_count; ++t) {
        const int slot = ((ix->ev_next - 1 - t) % mips_index::kEvRing + mips_index::kEvRing) % mips_index::kEvRing;
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, ix->ev0[slot], ix->ev1[slot]);
        if (e == hipErrorNotReady) return fail(MIPS_E_HIP, "mips_scan_timing: a scan is still running; synchronise the stream first");
        if (e != hipSuccess) return fail(MIPS_E_HIP, "hipEventElapsedTime failed: %s", hipGetErrorString(e));
        sum += ms;
        ++n;
    }
    if (out_sum_ms) *out_sum_ms = sum;
    if (out_count) *out_count = n;
    if (reset) {
        ix->ev_count = 0;
        ix->timing_armed = true;
    }
    return MIPS_OK;
}

} // extern "C"
