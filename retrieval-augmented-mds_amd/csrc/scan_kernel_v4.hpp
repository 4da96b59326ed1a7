// scan_kernel_v4: the query-stationary scan of scan_kernel_v3 on the 16x16x32 bf16 MFMA shape.
//
// Why: under this load the chip is power-limited.  tools/mfma_ceiling.hip (profiles/r1_class_maxima): a bare
// LDS-fed single-chain loop sustains 1.51-1.59 PFLOP/s on random bf16 data with 32x32x16 and 1.71-1.74 with
// 16x16x32 at the same bytes and flops per wave -- the chip holds a higher clock on the smaller shape.
//
// Mapping (v_mfma_f32_16x16x32_bf16: lane l -> c = l & 15, g = l >> 4; A[row c][k = 8 g + j],
// B[k = 8 g + j][col c], C/D col = c, row = 4 g + reg):
//   * a wave still owns 32 stationary queries = two 16-query column blocks n = 0, 1 (192 fragment VGPRs);
//   * a 32-document block is scored as two 16-document halves; per half and k32-step ONE A fragment feeds
//     two MFMAs (one per query block), 4 accumulator registers each -> 8 accumulator VGPRs instead of 16,
//     which pays for the second running list a lane now needs (lane (c, g) sees documents 4 g .. 4 g + 3 of
//     each half for queries c and 16 + c); the half-block epilogue is a handful of instructions unless a
//     document passes (16 accumulators and one epilogue per block do not fit: 13+ spilled registers);
//   * four lanes (g = 0..3) share a query, so a (query, split) pair has 4 sub-lists of KL entries.  A
//     document of global rank r can only be pushed out of its sub-list by KL better documents of the same
//     sub-list, so ranks 1 .. KL survive for certain (KL = 6 = k + 1 for k <= 5), the re-score pool is the
//     8 best of the union;
//   * shared insert bounds as scan_kernel_v3 TMODE 2: every sub-list publishes its best score (PUB = 1; its PUB-th best
//     in general) into class word (4 split + g) & 7 of its query, the bound is the minimum of the 8 words, re-read
//     sparsely: 8 PUB distinct documents score at least the bound, so nothing below it belongs to a pool of 8 PUB.
//     PUB = 4 serves the "optimistic" pools of 32 (mips_hip.hip): the lists still keep 6 entries each, the margin
//     check decides per query whether the pool selected from them was wide enough.
// Everything else (LDS-DMA ring, counted vmcnt, split barrier, strict-'>' tie rule) is scan_kernel_v3's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"
#include "scan_kernel_v3.hpp"

namespace mips {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// lane id the compiler cannot hoist out of a loop (so that values derived from it are re-derived where they
// are used instead of living in -- or being spilled from -- a register across the MFMA chain)
__device__ __forceinline__ unsigned lane_id_here() {
    unsigned ln = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
#endif
    return ln;
}

// NT_DOCS: non-temporal document DMA -- for searches of ONE query tile, where every document block has a single reader
template <int KL, int KS32, int AD, int TIMING_MODE = 0, bool NT_DOCS = false, int PUB = 1>
__global__ __launch_bounds__(512, 2) void scan_kernel_v4(ScanArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WAVES = 8;
    constexpr int TN = WAVES * 32;
    constexpr int STAGES = 3;
    constexpr int STAGE_BYTES = V3_DB * KS32 * 64; // 32 rows x (32 KS32) k x 2 B
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int PPW = PIECES / WAVES;
    static_assert(PIECES % WAVES == 0, "every wave must issue the same number of DMA pieces");
    static_assert(KL <= 8, "8 class words vouch for 8 documents");
    static_assert(PUB >= 1 && PUB <= KL, "a sub-list publishes one of its entries");
    constexpr int STEPS = 2 * KS32; // k32-steps per block (two halves)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int g = lane >> 4;

    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int qt = (xcd % p.qgroups) + p.qgroups * (j % p.qt_per_group);
    const int split = (xcd / p.qgroups) * p.splits_per_group + j / p.qt_per_group;
    if (qt >= p.nqt) return;
    if (p.spin_limit < 0 && tid == 0) *p.err = 1u; // test-only: force the scan-error path (include/mips_hip.h, "spin_limit")
    const bool idle_wave = (qt * TN + wave * 32) >= p.nq;
    // experiments (profiles/r3_v4_prio; results unchanged): TIMING_MODE 3 / 5 = static priority for the younger half of the
    // workgroup (waves 4 .. 7 lose every issue arbitration against their SIMD partners 0 .. 3: MI355X_MICROARCH.md, "Two waves per
    // SIMD", item 4); 4 / 5 = the arrival poll spins on s_nop instead of s_sleep 1 (64-cycle wake-up granularity)
    if ((TIMING_MODE == 3 || TIMING_MODE == 5) && wave >= 4) __builtin_amdgcn_s_setprio(1);

    const int b0 = split * p.tiles_per_split;
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // ---- stationary query fragments: lane holds Q[q0 + 16 n + c][32 s + 8 g .. +8)
    bf16x8 bq[2][KS32];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint16_t* qrow = p.qbuf + ((int64_t)qt * TN + wave * 32 + n * 16 + c) * p.ld + 8 * g;
#pragma unroll
        for (int s = 0; s < KS32; ++s) bq[n][s] = *reinterpret_cast<const bf16x8*>(qrow + 32 * s);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s = 0; s < KS32; ++s) asm volatile("" : "+v"(bq[n][s]));
#endif
    }

    float ls[2][KL];
    int li[2][KL];
    float thr[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        thr[n] = -INFINITY;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            ls[n][i] = -INFINITY;
            li[n][i] = IDX_NONE;
        }
    }

    // ---- shared insert bounds (scan_kernel_v3.hpp, TMODE 2): p.gthr = [query tile][wave][32 queries][8 words]
    constexpr unsigned THR_AREA = STAGES * STAGE_BYTES;
    constexpr unsigned THR_WAVE = 1024u;
    constexpr unsigned DUMP_AREA = THR_AREA + WAVES * THR_WAVE;
    static_assert(THR_AREA % 1024 == 0, "the wave areas are recovered from thr_addr by masking");
    // this lane's DMA chunk = LDS slot = voffset; re-derived from the lane id wherever it is needed
    auto thr_addr_of = [&](unsigned ln) { return THR_AREA + wave * THR_WAVE + ln * 16u; };
    *reinterpret_cast<uint4*>(smem + thr_addr_of(lane)) = make_uint4(0u, 0u, 0u, 0u);
    const __amdgpu_buffer_rsrc_t thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * (WAVES * THR_WAVE) - (int64_t)THR_AREA), 0,
        (int)(THR_AREA + WAVES * THR_WAVE), 0x00020000);
    auto refresh_thresholds = [&](bool real) { // !real: out-of-range dummy into the dump area (uniform vmcnt count)
        lds_void* dst = (lds_void*)(smem + (real ? THR_AREA + wave * THR_WAVE : DUMP_AREA));
        const unsigned thr_addr = thr_addr_of(lane_id_here());
        __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, dst, 16, real ? thr_addr : (thr_addr | 0x40000000u), 0, 0, 16);
    };

    // ---- LDS-DMA map (as v3): piece pc = slab * 4 + rg, 8 rows x 128 B
    const unsigned char* docs_b = reinterpret_cast<const unsigned char*>(p.docs);
    const int64_t row_bytes = (int64_t)p.ld * 2;
    // TIMING_MODE 14 (experiment, same results): the per-lane source offset lives in ONE register for the whole kernel instead of
    // being re-derived from the lane id for every piece (~10 vector instructions each, 6 pieces per block and wave)
    unsigned lane_off_kept = 0u;
    if (TIMING_MODE == 14) {
        const unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        lane_off_kept = (ln >> 3) * (unsigned)row_bytes + (((ln & 7u) ^ ((ln >> 4) & 7u)) << 4);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(lane_off_kept));
#endif
    }
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        if (TIMING_MODE == 6 || TIMING_MODE == 8) return; // diagnostic builds (results are wrong): the ring is never filled
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)(V3_DB * row_bytes), 0x00020000);
        const int pc = wave + WAVES * i;
        const int slab = pc >> 2, rg = pc & 3;
        // per-lane source offset, recomputed per piece from the lane id (the kernel has no VGPR to spare):
        // row lane >> 3 of the piece, chunk slot (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ ((4 rg + (lane >> 4)) & 7)
        unsigned lane_off0 = lane_off_kept;
        if (TIMING_MODE != 14) {
            const unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            lane_off0 = (ln >> 3) * (unsigned)row_bytes + (((ln & 7u) ^ ((ln >> 4) & 7u)) << 4);
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + stage * STAGE_BYTES + pc * 1024), 16,
                                                 (rg & 1) ? (lane_off0 ^ 64u) : lane_off0,
                                                 rg * 8 * (int)row_bytes + slab * 128, 0, NT_DOCS ? 2 : 0);
    };
    auto issue = [&](const unsigned char* blk_base, int stage) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(blk_base, stage, i);
    };

    // ---- A-fragment read addresses: row = 16 half + c, chunk 4 (s & 1) + g of slab s >> 1, slot chunk ^ swz
    auto rd0_of = [&](unsigned ln) { // (s & 1) == 0; the odd step is this ^ 64 (chunk + 4)
        const unsigned cc = ln & 15u, gg = ln >> 4;
        return (int)(cc * 128u + ((gg ^ ((cc >> 1) & 7u)) << 4));
    };

    // ---- split barrier (see scan_kernel_v3.hpp)
    const unsigned cnt_lds = (unsigned)(size_t)(lds_void*)(smem + DUMP_AREA + 1024);
    unsigned arrivals_needed = 0;
    constexpr int PER_BLOCK = PPW + 1;
    // TIMING_MODE 15 (diagnostic, same results): shader-clock cycles this wave spends in the DMA wait of its arrival and in the
    // block barrier's poll, summed over the launch into the words p.nq_dev points at (the launcher passes a scratch buffer there)
    // (32-bit sums of low words pinned to SGPRs: the kernel has no vector register for them)
    auto clk = [&]() { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__builtin_readcyclecounter()); };
    unsigned t_vm = 0u, t_bar = 0u;
    const unsigned t_start = TIMING_MODE == 15 ? clk() : 0u;
    auto arrive = [&]() {
        const unsigned t0 = TIMING_MODE == 15 ? clk() : 0u;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_BLOCK) : "memory");
        if (TIMING_MODE == 15) t_vm += clk() - t0;
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(cnt_lds), "v"(1u) : "memory");
#endif
    };
    auto wait_all = [&]() {
        arrivals_needed += WAVES;
        if (TIMING_MODE == 7) return; // diagnostic build (results are wrong): nobody waits for the block's arrivals
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
            if (TIMING_MODE == 4 || TIMING_MODE == 5) {
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("s_nop 15" ::: "memory");
#endif
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
    };

    // epilogue of one 16-document half: the 8 accumulator registers of a lane are documents base .. base + 3
    // against queries c (n = 0) and 16 + c (n = 1).  No synchronisation in here: at mid-block the partner
    // wave's MFMAs keep the matrix pipe busy while this wave runs the few instructions of the pre-test.
    auto epilogue_half = [&](f32x4 (&acc)[2], int blk, int half) {
        if (TIMING_MODE == 1 || TIMING_MODE == 8) { // diagnostic builds (results are wrong): no epilogue at all
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
#endif
            return;
        }
        // fast path: 6 max, 2 compares, one branch -- everything else is derived only if a document passes
        const float mx0 = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3]));
        const float mx1 = fmaxf(fmaxf(acc[1][0], acc[1][1]), fmaxf(acc[1][2], acc[1][3]));
        if (__ballot(mx0 > thr[0] || mx1 > thr[1]) != 0ull) {
            const unsigned thr_addr = thr_addr_of(lane_id_here());
            const int base = blk * V3_DB + 16 * half + (int)((thr_addr >> 6) & 12u); // + 4 g, from the lane bits
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float mark = ls[n][PUB - 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = acc[n][r];
                    if (s > thr[n]) {
                        list_insert<KL>(ls[n], li[n], s, base + r);
                        thr[n] = fmaxf(thr[n], ls[n][KL - 1]);
                    }
                }
                if (ls[n][PUB - 1] > mark) { // new PUB-th best of this sub-list: raise its class word, (4 split + g) & 7
                    const unsigned cls = (4u * (unsigned)split + ((thr_addr >> 8) & 3u)) & 7u;
                    publish_umax(thr_encode(ls[n][PUB - 1]), (thr_addr & ~0x3FFu) + ((thr_addr & 0xF0u) << 1) + 512u * n + 4u * cls, thr_rsrc);
                }
            }
        }
    };

    auto block = [&](bool refresh, int blk, int stage, const unsigned char* pbase, int pstage) {
        if (idle_wave) { // all of this wave's queries are padding (scan_kernel_v3.hpp): bring the documents, skip the arithmetic
            refresh_thresholds(false);
            issue(pbase, pstage);
            arrive();
            return;
        }
        const unsigned char* sa = smem + stage * STAGE_BYTES;
        const int rd0 = rd0_of(lane_id_here());
        // flattened step t = half * KS32 + s
        auto lds_frag = [&](int t) {
            const int half = t / KS32, s = t % KS32;
            const int off = half * 2048 + (s >> 1) * 4096 + ((s & 1) ? (rd0 ^ 64) : rd0);
            return *reinterpret_cast<const bf16x8*>(sa + off);
        };
        bf16x8 ar[AD];
#pragma unroll
        for (int t = 0; t < AD; ++t) ar[t] = lds_frag(t);
        refresh_thresholds(refresh); // first VMEM op of the block
        __builtin_amdgcn_sched_barrier(0);
        const bool ragged = (int64_t)(blk + 1) * V3_DB > p.ntotal; // uniform
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 acc[2];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS32; ++s) {
                const int t = half * KS32 + s;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t % AD], bq[0][s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t % AD], bq[1][s], acc[1], 0, 0, 0);
                if (t + AD < STEPS) ar[t % AD] = lds_frag(t + AD);
                // TIMING_MODE 9 (experiment, same results): the two waves of a SIMD (w, w + 4) issue their pieces half a period apart
                if (TIMING_MODE == 9) {
                    if ((t % (STEPS / PPW)) == ((STEPS / PPW) / 2) * (1 - (wave >> 2))) issue_piece(pbase, pstage, t / (STEPS / PPW));
                } else if ((t % (STEPS / PPW)) == (STEPS / PPW) / 2) issue_piece(pbase, pstage, t / (STEPS / PPW));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (half == 1) arrive(); // all LDS reads of this block are done; the epilogue runs un-synchronised
            if (half == 1 && refresh && TIMING_MODE != 1 && TIMING_MODE != 8) {
                // minimum of the 8 class words of queries c and 16 + c (what an earlier block's DMA brought, or 0);
                // inline asm, one query at a time: see scan_kernel_v3.hpp
                const unsigned thr_addr = thr_addr_of(lane_id_here());
                const unsigned a0 = (unsigned)(size_t)(lds_void*)smem + (thr_addr & ~0x3FFu) + ((thr_addr & 0xF0u) << 1);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    u32x4 w0 = {0u, 0u, 0u, 0u}, w1 = w0;
#if defined(__HIP_DEVICE_COMPILE__)
                    if (n == 0)
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
                    else
                        asm volatile("ds_read_b128 %0, %2 offset:512\n\tds_read_b128 %1, %2 offset:528\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
#endif
                    const unsigned key = min(min(min(w0[0], w0[1]), min(w0[2], w0[3])), min(min(w1[0], w1[1]), min(w1[2], w1[3])));
                    thr[n] = fmaxf(thr[n], key > 1u ? thr_decode(key - 1u) : -INFINITY);
                }
            }
            if (ragged) { // last block of the index only
                const int base = blk * V3_DB + 16 * half + 4 * (int)(lane_id_here() >> 4);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((int64_t)(base + r) >= p.ntotal) {
                        acc[0][r] = -INFINITY;
                        acc[1][r] = -INFINITY;
                    }
            }
            epilogue_half(acc, blk, half);
        }
    };

    const unsigned char* first = docs_b + (int64_t)b0 * V3_DB * row_bytes;
    const unsigned char* last = docs_b + (int64_t)(b1 - 1) * V3_DB * row_bytes;
    const int64_t blk_bytes = V3_DB * row_bytes;
    constexpr int AHEAD = STAGES - 1;
    if (nb > 0) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) { // same operation sequence as steady-state blocks (vmcnt arithmetic)
            refresh_thresholds(true);
            issue(a < nb ? first + a * blk_bytes : last, a);
        }
    }
    const unsigned char* pbase = nb > AHEAD ? first + AHEAD * blk_bytes : last;
    int stage = 0, pstage = AHEAD;
    if (tid == 0) *reinterpret_cast<unsigned*>(smem + DUMP_AREA + 1024) = 0u;
    __syncthreads();
    if (nb > 0) arrive();
    for (int i = 0; i < nb; ++i) {
        {
            const unsigned t0 = TIMING_MODE == 15 ? clk() : 0u;
            wait_all();
            if (TIMING_MODE == 15) t_bar += clk() - t0;
        }
        // TIMING_MODE 10 / 11 / 12 (experiment, same results): the second wave of each SIMD starts its block 64 / 128 / 192 cycles
        // late, so that the two waves' epilogues (an MFMA -> VALU dependency stall + ~10 dependent instructions per half, which both
        // reach at the same moment when they leave the barrier together) fall into each other's MFMA chains
        if (TIMING_MODE >= 10 && TIMING_MODE <= 12 && wave >= 4) __builtin_amdgcn_s_sleep(TIMING_MODE - 9);
        block(i < 8 || (i & 7) == 0, b0 + i, stage, pbase, pstage); // refresh schedule: scan_kernel_v3.hpp
        if (i + AHEAD + 1 < nb) pbase += blk_bytes;
        stage = stage == STAGES - 1 ? 0 : stage + 1;
        pstage = pstage == STAGES - 1 ? 0 : pstage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TIMING_MODE == 15 && p.nq_dev != nullptr && lane_id_here() == 0u && !idle_wave) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<int*>(p.nq_dev));
        atomicAdd(dbg + 0, (unsigned long long)t_vm);
        atomicAdd(dbg + 1, (unsigned long long)t_bar);
        atomicAdd(dbg + 3, (unsigned long long)(clk() - t_start));
        atomicAdd(dbg + 4, (unsigned long long)nb);
        atomicAdd(dbg + 5, 1ull);
    }

#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = qt * TN + wave * 32 + n * 16 + c;
        const size_t o = (((size_t)q * p.nsplit + split) * 4 + g) * KL;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            p.part_s[o + i] = ls[n][i];
            p.part_i[o + i] = li[n][i];
        }
    }
}

} // namespace mips
