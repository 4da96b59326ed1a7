// scan_kernel_ks: query-stationary scan for row pitch 1024 (BASELINE config 4, Longformer-large width) with the K
// dimension SPLIT over a wave pair.
//
// At pitch 1024 the stationary fragments of 32 queries are 256 registers: scan_kernel_v3's configuration for it runs ONE
// wave per SIMD (512-register file) on a 2-stage ring and is bound by its own stalls (1.16 PFLOP/s = 46 % in round 1).
// Here two waves share 32 queries: wave r of a pair keeps the fragments of k in [512 r, 512 r + 512) only (128
// registers), so 8 waves fit at TWO per SIMD again -- somebody always has MFMAs to issue while the other waits.
//   * per 32-document block a wave multiplies its K half of BOTH 16-document halves (2 x 16 k32-steps x 2 query blocks
//     = 64 MFMAs of 16x16x32) and reads only its K half of the LDS image (32 KiB instead of 64);
//   * the partial sums meet in LDS: wave r owns document half r.  It scores the FOREIGN half first, writes those 8
//     accumulator registers to its 2-KiB exchange slot (ds_write_b128 x 2) and bumps its counter, scores its OWN half,
//     waits for the partner's counter (written ~32 MFMAs earlier), adds the partner's partial sums and runs the top-K
//     epilogue of scan_kernel_v4 on its half.  The slot is single-buffered: a wave arrives at the block barrier only
//     AFTER reading its partner's slot, and nobody passes the next barrier before everyone has arrived;
//   * ring: 2 stages of 64 KiB (3 do not fit next to the exchange slots); the pieces of block i + 1 are issued at the
//     START of block i's chain (their stage was released by the barrier just passed) and must have landed at its end
//     (vmcnt(0) before the arrival) -- with 15 of 16 query tiles of an XCD served from L2 that is ~0.3 us of a ~1 us block.
// A (query, split) pair now has 8 sub-lists of 6 (4 lane groups x 2 document halves), same class words and the same
// strict-'>' tie rule as scan_kernel_v4; MFMA scores differ from the other kernels' in the last bits (two partial sums
// added), the exact results do not.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"
#include "scan_kernel_v3.hpp"
#include "scan_kernel_v4.hpp"

namespace mips {

template <int KL, int KS32, int AD, int TIMING_MODE = 0>
__global__ __launch_bounds__(512, 2) void scan_kernel_ks(ScanArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WAVES = 8;
    constexpr int PAIRS = 4;
    constexpr int TN = PAIRS * 32;                  // 128 queries per workgroup
    constexpr int STAGES = 2;
    constexpr int STAGE_BYTES = V3_DB * KS32 * 64;  // 32 rows x (32 KS32) k x 2 B = 64 KiB at pitch 1024
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int PPW = PIECES / WAVES;
    constexpr int KH = KS32 / 2;                    // k32-steps of one K half
    static_assert(KS32 % 4 == 0 && PIECES % WAVES == 0, "K halves must be whole 64-k slabs, DMA shares whole pieces");
    static_assert(PIECES / WAVES <= KS32 / 2, "one DMA piece per step of the foreign half");
    static_assert(KL <= 8, "8 class words vouch for 8 documents");
    constexpr int CHAIN = 2 * KH;                   // k-steps per block and wave (two document halves)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1;
    const int role = wave & 1;                      // K half AND owned document half
    const int c = lane & 15;
    const int g = lane >> 4;

    const int xcd = blockIdx.x & 7;
    const int j0 = blockIdx.x >> 3;
    const int qt = (xcd % p.qgroups) + p.qgroups * (j0 % p.qt_per_group);
    const int split = (xcd / p.qgroups) * p.splits_per_group + j0 / p.qt_per_group;
    if (qt >= p.nqt) return;
    if (p.spin_limit < 0 && tid == 0) *p.err = 1u; // test-only: force the scan-error path (include/mips_hip.h, "spin_limit")
    const bool idle_pair = (qt * TN + pair * 32) >= p.nq; // all 32 queries of the pair are padding (scan_kernel_v3.hpp)

    const int b0 = split * p.tiles_per_split;
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // ---- stationary fragments of this wave's K half: lane holds Q[q0 + 16 n + c][32 (KH role + j) + 8 g .. +8)
    bf16x8 bq[2][KH];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint16_t* qrow = p.qbuf + ((int64_t)qt * TN + pair * 32 + n * 16 + c) * p.ld + 32 * KH * role + 8 * g;
#pragma unroll
        for (int j = 0; j < KH; ++j) bq[n][j] = *reinterpret_cast<const bf16x8*>(qrow + 32 * j);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int j = 0; j < KH; ++j) asm volatile("" : "+v"(bq[n][j]));
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

    // ---- LDS map: ring | per-wave copies of the pair's class words (1 KiB each) | exchange slots (2 KiB each) | counters
    constexpr unsigned THR_AREA = STAGES * STAGE_BYTES;
    constexpr unsigned XCH_AREA = THR_AREA + WAVES * 1024u;
    constexpr unsigned CNT_AREA = XCH_AREA + WAVES * 2048u;
    // p.gthr = [query tile][pair][32 queries][8 words]: both waves of a pair read (and publish into) the same 1 KiB
    const __amdgpu_buffer_rsrc_t thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * (PAIRS * 1024)), 0, PAIRS * 1024, 0x00020000);
    *reinterpret_cast<uint4*>(smem + THR_AREA + wave * 1024u + lane * 16u) = make_uint4(0u, 0u, 0u, 0u);
    auto refresh_thresholds = [&]() {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, (lds_void*)(smem + THR_AREA + wave * 1024u), 16,
                                                 (unsigned)pair * 1024u + lane_id_here() * 16u, 0, 0, 16);
    };

    // ---- LDS-DMA map (as scan_kernel_v4): piece pc = slab * 4 + rg, 8 rows x 128 B
    const unsigned char* docs_b = reinterpret_cast<const unsigned char*>(p.docs);
    const int64_t row_bytes = (int64_t)p.ld * 2;
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)(V3_DB * row_bytes), 0x00020000);
        const int pc = wave + WAVES * i;
        const int slab = pc >> 2, rg = pc & 3;
        const unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const unsigned lane_off0 = (ln >> 3) * (unsigned)row_bytes + (((ln & 7u) ^ ((ln >> 4) & 7u)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + stage * STAGE_BYTES + pc * 1024), 16,
                                                 (rg & 1) ? (lane_off0 ^ 64u) : lane_off0,
                                                 rg * 8 * (int)row_bytes + slab * 128, 0, 0);
    };

    // ---- A-fragment read address of chain position (half, j): row 16 half + c, k32-step KH role + j
    auto rd0_of = [&](unsigned ln) {
        const unsigned cc = ln & 15u, gg = ln >> 4;
        return (int)(cc * 128u + ((gg ^ ((cc >> 1) & 7u)) << 4));
    };

    // ---- block barrier (split: arrive / wait) and the pair's exchange counters, all in LDS
    const unsigned cnt_lds = (unsigned)(size_t)(lds_void*)(smem + CNT_AREA);
    const unsigned xcnt_mine = cnt_lds + 16u + 4u * (unsigned)wave;
    const unsigned xcnt_partner = cnt_lds + 16u + 4u * (unsigned)(wave ^ 1);
    unsigned arrivals_needed = 0;
    auto arrive = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // 2-stage ring: this wave's share of the NEXT block has landed
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(cnt_lds), "v"(1u) : "memory");
#endif
    };
    auto poll = [&](unsigned addr, unsigned need) {
        for (int spin = 0;; ++spin) {
            unsigned v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
#endif
            if (__builtin_amdgcn_readfirstlane(v) >= need) break;
            if (spin > p.spin_limit) {
                if (lane == 0) *p.err = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // epilogue of the OWNED 16-document half (scan_kernel_v4's): acc[n] = documents base .. base + 3 vs query 16 n + c
    auto epilogue_half = [&](f32x4 (&acc)[2], int blk) {
        if (TIMING_MODE == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
#endif
            return;
        }
        const float mx0 = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3]));
        const float mx1 = fmaxf(fmaxf(acc[1][0], acc[1][1]), fmaxf(acc[1][2], acc[1][3]));
        if (__ballot(mx0 > thr[0] || mx1 > thr[1]) != 0ull) {
            const unsigned ln = lane_id_here();
            const int base = blk * V3_DB + 16 * role + 4 * (int)(ln >> 4);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float mark = ls[n][0];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = acc[n][r];
                    if (s > thr[n]) {
                        list_insert<KL>(ls[n], li[n], s, base + r);
                        thr[n] = fmaxf(thr[n], ls[n][KL - 1]);
                    }
                }
                if (ls[n][0] > mark) { // new best of this sub-list: raise its class word, (4 split + g) & 7
                    const unsigned cls = (4u * (unsigned)split + (ln >> 4)) & 7u;
                    publish_umax(thr_encode(ls[n][0]), (unsigned)pair * 1024u + (16u * n + (ln & 15u)) * 32u + 4u * cls, thr_rsrc);
                }
            }
        }
    };

    auto block = [&](bool refresh, int blk, int stage, const unsigned char* nbase, int nstage, bool have_next) {
        // pieces of the NEXT block first: their stage was released by the barrier just passed
        if (refresh) refresh_thresholds();
        if (idle_pair) {
            if (have_next) {
#pragma unroll
                for (int i = 0; i < PPW; ++i) issue_piece(nbase, nstage, i);
            }
            arrive();
            return;
        }
        // this wave's K half starts at slab (KH / 2) role of the image; rows of the foreign half first
        const unsigned char* sa = smem + stage * STAGE_BYTES + (KH / 2) * role * 4096;
        const int rd0 = rd0_of(lane_id_here());
        const int hoff0 = (1 - role) * 2048, hoff1 = role * 2048;
        // chain position t = hpos * KH + j: hpos 0 = the FOREIGN half (1 - role), hpos 1 = the OWNED half (role);
        // k32-step KH role + j has the parity of j (KH is even)
        auto lds_frag = [&](int t) {
            const int hpos = t / KH, j = t % KH;
            return *reinterpret_cast<const bf16x8*>(sa + (hpos == 0 ? hoff0 : hoff1) + (j >> 1) * 4096 + ((j & 1) ? (rd0 ^ 64) : rd0));
        };
        bf16x8 ar[AD];
#pragma unroll
        for (int t = 0; t < AD; ++t) ar[t] = lds_frag(t);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][2];
#pragma unroll
        for (int hpos = 0; hpos < 2; ++hpos) {
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[hpos][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                const int t = hpos * KH + j;
                acc[hpos][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t % AD], bq[0][j], acc[hpos][0], 0, 0, 0);
                acc[hpos][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t % AD], bq[1][j], acc[hpos][1], 0, 0, 0);
                if (t + AD < CHAIN) ar[t % AD] = lds_frag(t + AD);
                if (have_next && hpos == 0 && j < PPW) issue_piece(nbase, nstage, j); // early: they must land by the end of this chain
                __builtin_amdgcn_sched_barrier(0);
            }
            if (hpos == 0) {
                // hand the foreign half's partial sums to the partner: 2 KiB slot [n][lane][4], then the counter (a
                // wave's LDS operations complete in order)
                unsigned char* slot = smem + XCH_AREA + wave * 2048u + lane_id_here() * 16u;
                *reinterpret_cast<f32x4*>(slot) = acc[0][0];
                *reinterpret_cast<f32x4*>(slot + 1024) = acc[0][1];
#if defined(__HIP_DEVICE_COMPILE__)
                if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(xcnt_mine), "v"(1u) : "memory");
#endif
            }
        }
        // the partner's partial sums for MY half
        poll(xcnt_partner, (unsigned)(blk - b0) + 1u);
        f32x4 own[2];
        {
            const unsigned char* slot = smem + XCH_AREA + (wave ^ 1) * 2048u + lane_id_here() * 16u;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(slot);
            const f32x4 p1 = *reinterpret_cast<const f32x4*>(slot + 1024);
            own[0] = acc[1][0] + p0; // (K-low partial) + (K-high partial)
            own[1] = acc[1][1] + p1;
        }
        arrive(); // chain done, the partner's slot read, this wave's share of the next block landed
        if (refresh && TIMING_MODE == 0) {
            // minimum of the 8 class words of queries c and 16 + c (what an earlier refresh brought, or 0)
            const unsigned a0 = (unsigned)(size_t)(lds_void*)smem + THR_AREA + wave * 1024u + (lane_id_here() & 15u) * 32u;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                u32x4 w0 = {0u, 0u, 0u, 0u}, w1 = w0;
#if defined(__HIP_DEVICE_COMPILE__)
                if (n == 0) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
                else asm volatile("ds_read_b128 %0, %2 offset:512\n\tds_read_b128 %1, %2 offset:528\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
#endif
                const unsigned key = min(min(min(w0[0], w0[1]), min(w0[2], w0[3])), min(min(w1[0], w1[1]), min(w1[2], w1[3])));
                thr[n] = fmaxf(thr[n], key > 1u ? thr_decode(key - 1u) : -INFINITY);
            }
        }
        if ((int64_t)(blk + 1) * V3_DB > p.ntotal) { // ragged last block of the index (uniform)
            const int base = blk * V3_DB + 16 * role + 4 * (int)(lane_id_here() >> 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((int64_t)(base + r) >= p.ntotal) {
                    own[0][r] = -INFINITY;
                    own[1][r] = -INFINITY;
                }
        }
        epilogue_half(own, blk);
    };

    const unsigned char* first = docs_b + (int64_t)b0 * V3_DB * row_bytes;
    const int64_t blk_bytes = V3_DB * row_bytes;
    if (tid < 16) reinterpret_cast<unsigned*>(smem + CNT_AREA)[tid] = 0u;
    if (nb > 0) {
        refresh_thresholds();
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(first, 0, i);
    }
    __syncthreads();
    if (nb > 0) arrive(); // this wave's share of block 0 has landed
    for (int i = 0; i < nb; ++i) {
        arrivals_needed += WAVES;
        poll(cnt_lds, arrivals_needed); // every share of block i landed; everyone is done with block i - 1
        const bool have_next = i + 1 < nb;
        block(i < 8 || (i & 7) == 0, b0 + i, i & 1, first + (int64_t)(i + 1) * blk_bytes, (i + 1) & 1, have_next);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // lists: [q][nsplit][8 = 2 document halves x 4 lane groups][KL]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int q = qt * TN + pair * 32 + n * 16 + c;
        const size_t o = (((size_t)q * p.nsplit + split) * 8 + role * 4 + g) * KL;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            p.part_s[o + i] = ls[n][i];
            p.part_i[o + i] = li[n][i];
        }
    }
}

} // namespace mips
