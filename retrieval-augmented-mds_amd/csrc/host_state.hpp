// Host side of libmips_hip.so, part 1 of 3 (included by mips_hip.hip, one translation unit): error reporting, device / stream
// helpers, the index object (storage in HBM, scratch, options, statistics) and what keeps it consistent (growth, phi, row norms,
// the bf16 image of an fp32-exact index, stream ordering).
#pragma once

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

} // namespace
