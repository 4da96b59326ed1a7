// C ABI of the MI355X MIPS backend (include/mips_hip.h), gfx950 only.  ONE translation unit: the kernels (*.hpp), the host side
// in three parts -- host_state.hpp (the index object), host_launch.hpp (which kernel answers which search), host_search.hpp (what a
// search does around its scan) -- and, below, the extern "C" entry points themselves.
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

#include "host_state.hpp"
#include "host_launch.hpp"
#include "host_search.hpp"

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
                         ix->opt_margin != 0 && !split && (!out_dev || ix->opt_margin >= 2) && ix->opt_f32_fast != 0 && ix->opt_sub == 0 &&
                         ix->opt_variant == 0; // (forced kernels and the A/B instances keep the pool of 8)
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
