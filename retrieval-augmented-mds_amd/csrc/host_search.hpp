// Host side of libmips_hip.so, part 3 of 3 (included by mips_hip.hip): what a search does around its scan -- the one-launch
// kernel for the reference's own call shape, the exact pass of flagged queries (resolve_flagged), the fall-back re-scans, the
// certificate (finish_margin), the optimistic / two-stage first scans (scan_and_finish) and the scope of the margin mode.
#pragma once

namespace {

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
