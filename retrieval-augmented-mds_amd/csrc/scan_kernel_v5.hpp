// scan_kernel_v5: the query-stationary scan with 64 stationary queries per wave -- half the LDS read bytes per MFMA.
//
// Why (profiles/r2_mapping_ceiling): under this load the chip is power-limited, and what it sustains depends on the
// energy per MFMA.  Bare LDS-fed 16x16x32 loops on random bf16 data: one A fragment per TWO MFMAs (scan_kernel_v4's
// loop) 1.70-1.71 PFLOP/s at 1.78 GHz; one A fragment per FOUR MFMAs 1.86-1.87 PFLOP/s at 1.94 GHz; operands in
// registers only 2.0.  The A-fragment reads (every wave re-reads the whole document block from LDS) are the largest
// consumer next to the MFMAs themselves, and the only way to halve them is to let one fragment feed twice as many
// stationary queries.  64 queries x 768 k = 384 fragment registers, so: ONE wave per SIMD, 4 waves per workgroup
// (still 256 stationary queries per CU), the 512-register file, 240 of the fragment registers pinned in AGPRs (MFMA
// reads B operands from there directly).
//
// One wave per SIMD has nobody to hide behind, so the block structure differs from v3 / v4:
//   * top-K epilogue STAGGERED into the next chain: the pre-test of query block n (2 max + 1 compare + a branch) sits
//     right before the first MFMA that overwrites acc[n] (C = 0), i.e. under the MFMAs of the other query blocks;
//   * each wave's k-order is ROTATED so that a block starts on the 64-k slabs this wave's own LDS-DMA brought
//     (landed: proven by the wave's own counted vmcnt); the arrival poll for the other waves' slabs comes only after
//     those first 24 MFMAs.  The stationary fragments are loaded in the rotated order, so register indices stay static;
//   * the A-fragment prefetch ring runs across halves and blocks without a gap.
// Everything else is scan_kernel_v4's: 3-stage LDS-DMA ring of 32-document blocks, XOR-swizzled image, lane (c, g)
// holds documents 4 g .. 4 g + 3 of a 16-document half against queries 16 n + c, 4 sub-lists of KL = 6 per (query,
// split), class-maximum insert bounds re-read sparsely, strict '>' tie rule, split barrier on an LDS arrival counter.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "scan_kernel.hpp"
#include "scan_kernel_v3.hpp"
#include "scan_kernel_v4.hpp"

namespace mips {

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), every call inlined.  hipcc
// declines to fully unroll a `#pragma unroll` loop of the size of a chain step sequence (and a loop left rolled turns
// the fragment-register indices into scratch accesses).
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int KL, int KS32, int AD = 2, int TIMING_MODE = 0>
__global__ __launch_bounds__(256, 1) void scan_kernel_v5(ScanArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WAVES = 4;
    constexpr int NQB = 4;                          // 16-query column blocks per wave
    constexpr int TN = WAVES * NQB * 16;            // 256 queries per workgroup
    constexpr int STAGES = 3;
    constexpr int STAGE_BYTES = V3_DB * KS32 * 64;  // 32 rows x (32 KS32) k x 2 B
    constexpr int PIECES = STAGE_BYTES / 1024;      // 1-KiB DMA pieces per block: 4 row groups per 64-k slab
    constexpr int PPW = PIECES / WAVES;
    constexpr int SLABS = KS32 / 2;                 // 64-k slabs per block
    constexpr int SPW = SLABS / WAVES;              // slabs a wave's DMA brings
    constexpr int OWN = 2 * SPW;                    // k32-steps that touch only the wave's own slabs
    static_assert(SLABS % WAVES == 0, "every wave must own whole slabs");
    static_assert(KL <= 8, "8 class words vouch for 8 documents");
    static_assert(PPW == 4 * SPW, "pieces per wave");
    static_assert(AD >= 1 && AD < OWN, "the ring's first AD positions of a block must be own slabs, with a step left for the poll");
    constexpr int NFRAG = NQB * KS32;
    constexpr int NFRAG_A = NFRAG > 64 ? 64 : NFRAG / 2; // stationary fragments pinned in AGPRs (240 registers; the accumulators take the other 16)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int g = lane >> 4;

    const int xcd = blockIdx.x & 7;
    const int j0 = blockIdx.x >> 3;
    const int qt = (xcd % p.qgroups) + p.qgroups * (j0 % p.qt_per_group);
    const int split = (xcd / p.qgroups) * p.splits_per_group + j0 / p.qt_per_group;
    if (qt >= p.nqt) return;
    if (p.spin_limit < 0 && tid == 0) *p.err = 1u; // test-only: force the scan-error path (include/mips_hip.h, "spin_limit")

    const int b0 = split * p.tiles_per_split;
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // k-order of this wave: chain position j works on k32-step (OWN * wave + j) mod KS32, i.e. on slab
    // (SPW * wave + (j >> 1)) mod SLABS -- its own slabs first
    const int rot = OWN * wave;

    // ---- stationary query fragments, in the wave's k-order: lane holds Q[q0 + 16 n + c][32 s + 8 g .. +8)
    bf16x8 bq[NQB][KS32];
    // loaded in batches of 12 fragments, each batch pinned before the next is requested: left alone, hipcc issues all
    // 96 loads first (384 transient VGPRs), spills fragments in the prologue and reloads them -- behind a vmcnt(0) that
    // drains the LDS-DMA ring -- in every block
#pragma unroll
    for (int n = 0; n < NQB; ++n) {
        const uint16_t* qrow = p.qbuf + ((int64_t)qt * TN + wave * 64 + n * 16 + c) * p.ld + 8 * g;
#pragma unroll
        for (int jb = 0; jb < KS32; jb += 12) {
#pragma unroll
            for (int j = jb; j < jb + 12 && j < KS32; ++j) {
                int s = rot + j;
                if (s >= KS32) s -= KS32;
                bq[n][j] = *reinterpret_cast<const bf16x8*>(qrow + 32 * s);
            }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int j = jb; j < jb + 12 && j < KS32; ++j) {
                if (n * KS32 + j < NFRAG_A) asm volatile("" : "+a"(bq[n][j]) : : "memory");
                else asm volatile("" : "+v"(bq[n][j]) : : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }

    float ls[NQB][KL];
    int li[NQB][KL];
    float thr[NQB];
#pragma unroll
    for (int n = 0; n < NQB; ++n) {
        thr[n] = -INFINITY;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            ls[n][i] = -INFINITY;
            li[n][i] = IDX_NONE;
        }
    }

    // ---- shared insert bounds: p.gthr = [query tile][wave][64 queries][8 words]; 2 KiB per wave, two DMA operations
    constexpr unsigned THR_AREA = STAGES * STAGE_BYTES;
    constexpr unsigned THR_WAVE = 2048u;
    constexpr unsigned DUMP_AREA = THR_AREA + WAVES * THR_WAVE;
    static_assert(THR_AREA % 1024 == 0, "wave areas are 1-KiB aligned");
    auto thr_base_of = [&]() { return THR_AREA + wave * THR_WAVE; };
    {
        const unsigned a = thr_base_of() + lane * 16u;
        *reinterpret_cast<uint4*>(smem + a) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(smem + a + 1024u) = make_uint4(0u, 0u, 0u, 0u);
    }
    const __amdgpu_buffer_rsrc_t thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * (WAVES * THR_WAVE) - (int64_t)THR_AREA), 0,
        (int)(THR_AREA + WAVES * THR_WAVE), 0x00020000);
    auto refresh_thresholds = [&]() { // two 1-KiB pieces; blocks that do not refresh issue nothing (their counted wait differs)
        const unsigned voff = thr_base_of() + lane_id_here() * 16u;
#pragma unroll
        for (int h = 0; h < 2; ++h)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, (lds_void*)(smem + thr_base_of() + 1024u * h), 16, voff + 1024u * h, 0, 0, 16);
    };

    // ---- LDS-DMA map: piece (slab, rg) = rows 8 rg .. 8 rg + 7 of 64-k slab `slab`, at stage + (4 slab + rg) KiB;
    // wave w brings slabs SPW w .. SPW w + SPW - 1 (all four row groups): its i-th piece is slab SPW w + (i >> 2), rg i & 3
    const unsigned char* docs_b = reinterpret_cast<const unsigned char*>(p.docs);
    const int64_t row_bytes = (int64_t)p.ld * 2;
    // The lane part of the source address lives in two persistent VGPRs (even / odd row groups): with one wave per
    // SIMD every instruction of a piece is MFMA issue time, and re-deriving it per piece (scan_kernel_v4 does, it has
    // no register to spare and a SIMD partner to hide behind) costs ~5 dependent VALU operations x 14 pieces a block
    const unsigned lane_off0 = (unsigned)(lane >> 3) * (unsigned)row_bytes + ((((unsigned)lane & 7u) ^ (((unsigned)lane >> 4) & 7u)) << 4);
    const unsigned lane_off1 = lane_off0 ^ 64u;
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)(V3_DB * row_bytes), 0x00020000);
        const int slab = SPW * wave + (i >> 2), rg = i & 3;
        const int pc = slab * 4 + rg;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + stage * STAGE_BYTES + pc * 1024), 16,
                                                 (rg & 1) ? lane_off1 : lane_off0, rg * 8 * (int)row_bytes + slab * 128, 0, 0);
    };

    // ---- A-fragment read: row 16 half + c, chunk 4 (s & 1) + g of slab s >> 1, slot chunk ^ ((row >> 1) & 7)
    auto rd0_of = [&](unsigned ln) {
        const unsigned cc = ln & 15u, gg = ln >> 4;
        return (int)(cc * 128u + ((gg ^ ((cc >> 1) & 7u)) << 4));
    };
    // fragment of chain position j (k32-step rot + j, wrapped) of `half` in ring stage `stage`
    auto lds_frag = [&](int stage, int half, int j, int rd0) {
        int slab = SPW * wave + (j >> 1);
        if (slab >= SLABS) slab -= SLABS;
        const int off = stage * STAGE_BYTES + half * 2048 + slab * 4096 + ((j & 1) ? (rd0 ^ 64) : rd0);
        return *reinterpret_cast<const bf16x8*>(smem + off);
    };

    // ---- split barrier: arrival counter in LDS (scan_kernel_v3.hpp)
    const unsigned cnt_lds = (unsigned)(size_t)(lds_void*)(smem + DUMP_AREA + 1024);
    unsigned arrivals_needed = 0;
    constexpr int PER_BLOCK = PPW + 2; // VMEM operations a wave issues per block: 2 threshold pieces + its document pieces
    auto arrive = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(cnt_lds), "v"(1u) : "memory");
#endif
    };
    auto wait_all = [&]() {
        arrivals_needed += WAVES;
        for (int spin = 0;; ++spin) {
            unsigned v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(cnt_lds) : "memory");
#endif
            if (__builtin_amdgcn_readfirstlane(v) >= arrivals_needed) break;
            if (spin > p.spin_limit) {
                if (lane == 0) *p.err = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // ---- top-K epilogue of ONE query block n of a 16-document half: a = documents base .. base + 3 against query
    // 16 n + c.  Fast path: 2 max, 1 compare and one NOT-TAKEN branch (the insert code is laid out out of line:
    // with one wave per SIMD a taken branch is an instruction-fetch bubble in the MFMA stream; the first version of
    // this kernel, whose fast path jumped over the insert code 8 times per block and tested `ragged` separately, lost
    // 17 % to its epilogue where scan_kernel_v4 loses 5 %).
    auto epilogue_n = [&](const f32x4& a, int n, int blk, int half, bool ragged) __attribute__((always_inline)) {
        if (TIMING_MODE == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(a));
#endif
            return;
        }
        const float mx = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
        if (__builtin_expect(ragged || __ballot(mx > thr[n]) != 0ull, 0)) {
            const unsigned ln = lane_id_here();
            const int base = blk * V3_DB + 16 * half + 4 * (int)(ln >> 4);
            f32x4 v = a;
            if (ragged) { // last block of the index only (uniform)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((int64_t)(base + r) >= p.ntotal) v[r] = -INFINITY;
            }
            const float mark = ls[n][0];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = v[r];
                if (sc > thr[n]) {
                    list_insert<KL>(ls[n], li[n], sc, base + r);
                    thr[n] = fmaxf(thr[n], ls[n][KL - 1]);
                }
            }
            if (ls[n][0] > mark) { // new best of this sub-list: raise its class word, (4 split + g) & 7
                const unsigned cls = (4u * (unsigned)split + (ln >> 4)) & 7u;
                publish_umax(thr_encode(ls[n][0]), thr_base_of() + (16u * n + (ln & 15u)) * 32u + 4u * cls, thr_rsrc);
            }
        }
    };
    // apply the class words an earlier refresh brought: bound of query 16 n + c = minimum of its 8 words
    auto apply_bounds = [&]() {
        const unsigned a0 = (unsigned)(size_t)(lds_void*)smem + thr_base_of() + (lane_id_here() & 15u) * 32u;
#pragma unroll
        for (int n = 0; n < NQB; ++n) {
            u32x4 w0 = {0u, 0u, 0u, 0u}, w1 = w0;
#if defined(__HIP_DEVICE_COMPILE__)
            // inline asm: for an ordinary load of an LDS-DMA destination hipcc first drains vmcnt(0)
            if (n == 0) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
            else if (n == 1) asm volatile("ds_read_b128 %0, %2 offset:512\n\tds_read_b128 %1, %2 offset:528\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
            else if (n == 2) asm volatile("ds_read_b128 %0, %2 offset:1024\n\tds_read_b128 %1, %2 offset:1040\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
            else asm volatile("ds_read_b128 %0, %2 offset:1536\n\tds_read_b128 %1, %2 offset:1552\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
#endif
            const unsigned key = min(min(min(w0[0], w0[1]), min(w0[2], w0[3])), min(min(w1[0], w1[1]), min(w1[2], w1[3])));
            thr[n] = fmaxf(thr[n], key > 1u ? thr_decode(key - 1u) : -INFINITY);
        }
    };

    const unsigned char* first = docs_b + (int64_t)b0 * V3_DB * row_bytes;
    const unsigned char* last = docs_b + (int64_t)(b1 - 1) * V3_DB * row_bytes;
    const int64_t blk_bytes = V3_DB * row_bytes;
    constexpr int AHEAD = STAGES - 1;
    if (nb > 0) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) { // same operation sequence as steady-state blocks (vmcnt arithmetic)
            refresh_thresholds();
#pragma unroll
            for (int i = 0; i < PPW; ++i) issue_piece(a < nb ? first + a * blk_bytes : last, a, i);
        }
    }
    const unsigned char* pbase = nb > AHEAD ? first + AHEAD * blk_bytes : last;
    if (tid == 0) *reinterpret_cast<unsigned*>(smem + DUMP_AREA + 1024) = 0u;
    __syncthreads();
    if (nb == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // this wave's share of block 0 has landed: everything but the youngest block's worth of operations
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_BLOCK) : "memory");
        arrive();

        // ONE accumulator set (a second one does not fit: 384 fragment registers + lists leave ~28 registers for
        // everything else, hipcc spills fragments and reloads them behind vmcnt(0)): the pre-test of query block n sits
        // right before the chain-start MFMA that overwrites acc[n] (C = 0), under the MFMAs of the other query blocks
        f32x4 acc[NQB];
#pragma unroll
        for (int n = 0; n < NQB; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 ar[AD];
        {
            const int rd0 = rd0_of(lane_id_here());
#pragma unroll
            for (int t = 0; t < AD; ++t) ar[t] = lds_frag(0, 0, t, rd0); // own slabs of block 0
        }
        int stage = 0, pstage = AHEAD;
        int pend_blk = -1; // block whose second half still waits for its epilogue (-1: none)
        bool pend_refresh = false;
        // DMA piece schedule: PPW pieces of block i + 2, all after the arrival poll (which frees their stage):
        // half 0 at chain positions OWN + 1, OWN + 4, ... , half 1 spread evenly, the last one before position KS32 - 3
        constexpr int P0 = PPW / 2, P1 = PPW - P0;
        constexpr int STRIDE0 = (KS32 - OWN - 1) / P0 > 0 ? (KS32 - OWN - 1) / P0 : 1;
        constexpr int LASTJ = KS32 - AD - 1; // chain position of half 1 after which the ring reads the NEXT block
        constexpr int STRIDE1 = LASTJ / P1 > 0 ? LASTJ / P1 : 1;
        static_assert(OWN + 1 + (P0 - 1) * STRIDE0 < KS32, "half-0 piece schedule");
        static_assert((P1 - 1) * STRIDE1 < LASTJ, "half-1 piece schedule: all pieces before the counted wait");

        for (int i = 0; i < nb; ++i) {
            const int blk = b0 + i;
            const bool refresh = i < 8 || (i & 7) == 0; // refresh schedule: scan_kernel_v3.hpp
            const int nstage = stage == STAGES - 1 ? 0 : stage + 1;
            const bool ragged = (int64_t)(blk + 1) * V3_DB > p.ntotal;       // uniform
            const bool pragged = (int64_t)blk * V3_DB > p.ntotal;           // the pending block (blk - 1) is ragged
            if (refresh) refresh_thresholds(); // first VMEM operations of the block
            if (pend_blk >= 0 && pend_refresh && TIMING_MODE == 0) apply_bounds();
            const int rd0 = rd0_of(lane_id_here());
            __builtin_amdgcn_sched_barrier(0);
            // one 16-document half; HALF is a compile-time constant (two explicit instantiations: hipcc does not
            // unroll a loop over the halves around a body of this size, and a runtime `half` would turn the
            // fragment-register indices into scratch accesses)
            auto chain = [&](auto half_c) {
                constexpr int half = decltype(half_c)::value;
                static_for<KS32>([&](auto j_c) __attribute__((always_inline)) {
                    constexpr int j = decltype(j_c)::value;
                    constexpr int t = half * KS32 + j;
                    if (j == 0) {
#pragma unroll
                        for (int n = 0; n < NQB; ++n) {
                            if (half == 1) epilogue_n(acc[n], n, blk, 0, ragged);
                            else if (pend_blk >= 0) epilogue_n(acc[n], n, pend_blk, 1, pragged);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t % AD], bq[n][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int n = 0; n < NQB; ++n)
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t % AD], bq[n][j], acc[n], 0, 0, 0);
                    }
                    // prefetch ring, continuous across halves and blocks (the next block starts on own slabs)
                    {
                        const int tn = t + AD;
                        if (tn < 2 * KS32) ar[t % AD] = lds_frag(stage, tn / KS32, tn % KS32, rd0);
                        else ar[t % AD] = lds_frag(nstage, 0, tn - 2 * KS32, rd0);
                    }
                    // Before the prefetch ring reaches a FOREIGN slab (position OWN, read at step OWN - AD): the other
                    // waves' slabs of this block have landed, and everyone is done with the previous block (whose
                    // stage the pieces below overwrite)
                    if (half == 0 && j == OWN - AD - 1) wait_all();
                    if (half == 0 && j > OWN && (j - OWN - 1) % STRIDE0 == 0 && (j - OWN - 1) / STRIDE0 < P0)
                        issue_piece(pbase, pstage, (j - OWN - 1) / STRIDE0);
                    if (half == 1 && j % STRIDE1 == 0 && j / STRIDE1 < P1) issue_piece(pbase, pstage, P0 + j / STRIDE1);
                    if (half == 1 && j == LASTJ) {
                        // before the ring crosses into the next block: this wave's share of it has landed (everything
                        // but this block's own operations: its pieces, and two more in a block that refreshed)
                        if (refresh) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            chain(std::integral_constant<int, 0>{});
            chain(std::integral_constant<int, 1>{});
            arrive(); // all LDS reads of this block are issued and consumed; the second half's epilogue runs later
            pend_blk = blk;
            pend_refresh = refresh;
            if (i + AHEAD + 1 < nb) pbase += blk_bytes;
            stage = nstage;
            pstage = pstage == STAGES - 1 ? 0 : pstage + 1;
        }
        // drain: the last block's second half
        if (pend_refresh && TIMING_MODE == 0) apply_bounds();
        {
            const bool pragged = (int64_t)(pend_blk + 1) * V3_DB > p.ntotal;
#pragma unroll
            for (int n = 0; n < NQB; ++n) epilogue_n(acc[n], n, pend_blk, 1, pragged);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no DMA may outlive the workgroup's LDS allocation
    }

#pragma unroll
    for (int n = 0; n < NQB; ++n) {
        const int q = qt * TN + wave * 64 + n * 16 + c;
        const size_t o = (((size_t)q * p.nsplit + split) * 4 + g) * KL;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            p.part_s[o + i] = ls[n][i];
            p.part_i[o + i] = li[n][i];
        }
    }
}

} // namespace mips
