// Host side of libmips_hip.so, part 2 of 3 (included by mips_hip.hip): which scan kernel answers which search -- the instance
// tables, the configurations of scan_kernel_e8, and launch_search<K'>: geometry (splits, query tiles, XCD groups), scratch, the scan
// launch, and the tail behind it (merge_select -> rescore_rank with the margin check).  DESIGN.md section 4, "Dispatch".
#pragma once

namespace {

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
    const bool want_k3 = k3_opt || (ks_shape && (ix->opt_sub == 0 || (ix->opt_sub >= 61 && ix->opt_sub <= 74)) && (ix->opt_variant == 7 || (ix->opt_variant == 0 && nq > 256)));
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
    if (want_k3 && ix->opt_sub >= 72 && ix->opt_sub <= 74) list_len = 3; // scan_kernel_k3 with sub-lists of 3: A prefetch depth 2 / 3 / 4
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
            else if (ix->opt_sub == 72) rc3 = gok3(mips::scan_kernel_k3<3, 32, 2, 0, 1>);
            else if (ix->opt_sub == 73) rc3 = gok3(mips::scan_kernel_k3<3, 32, 3, 0, 1>);
            else if (ix->opt_sub == 74) rc3 = gok3(mips::scan_kernel_k3<3, 32, 4, 0, 1>);
            else if (ix->opt_sub == 70) { // cycle accounting of the waves' waits (same results): printed to stderr, synchronises
                static unsigned long long* dbg_dev = nullptr;
                if (!dbg_dev) HIP_TRY(hipMalloc((void**)&dbg_dev, 64));
                HIP_TRY(hipMemsetAsync(dbg_dev, 0, 64, st));
                a.nq_dev = reinterpret_cast<const int*>(dbg_dev);
                rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 10>);
                a.nq_dev = ix->nq_dev;
                unsigned long long h[8];
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(h, dbg_dev, 64, hipMemcpyDeviceToHost));
                if (h[5] == 0) fprintf(stderr, "k3 waits: nothing recorded (rc %d)\n", rc3);
                if (h[5] > 0)
                    fprintf(stderr, "k3 waits per wave and block (shader cycles; %llu waves, %.1f blocks each): DMA wait %.0f  block barrier %.0f  pair %.0f  of %.0f per block\n",
                            h[5], (double)h[4] / (double)h[5], (double)h[0] / (double)h[4], (double)h[1] / (double)h[4], (double)h[2] / (double)h[4],
                            (double)h[3] / (double)h[4]);
            } else if (ix->opt_sub == 69) rc3 = gok3(mips::scan_kernel_k3<K3_KLL, 32, 2, 9>); // ... three ahead, by one workgroup per document stream only
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
            else if (ix->ld == 768 && ix->opt_sub == 59) { // cycle accounting of the waves' explicit waits (same results): stderr, synchronises
                static unsigned long long* dbg_dev = nullptr;
                if (!dbg_dev) HIP_TRY(hipMalloc((void**)&dbg_dev, 64));
                HIP_TRY(hipMemsetAsync(dbg_dev, 0, 64, st));
                a.nq_dev = reinterpret_cast<const int*>(dbg_dev);
                rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 15>);
                a.nq_dev = ix->nq_dev;
                unsigned long long h[8];
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipMemcpy(h, dbg_dev, 64, hipMemcpyDeviceToHost));
                if (h[5] > 0)
                    fprintf(stderr, "v4 waits per wave and block (shader cycles; %llu waves, %.1f blocks each): DMA wait %.0f  block barrier %.0f  of %.0f per block\n",
                            h[5], (double)h[4] / (double)h[5], (double)h[0] / (double)h[4], (double)h[1] / (double)h[4], (double)h[3] / (double)h[4]);
            } else if (ix->ld == 768 && ix->opt_sub == 58) rc2 = go4(mips::scan_kernel_v4<V4_KLL, 24, 2, 14>); // DMA lane offset kept in a register
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

} // namespace
