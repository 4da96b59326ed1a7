// scan_kernel_k3: row pitch 1024 with 192 STATIONARY QUERIES per CU (scan_kernel_ks's wave pairs, three 16-query blocks each).
//
// Why (profiles/r2_pitch1024, profiles/r3_pitch1024): at pitch 1024 a 32-document block is 64 KiB.  With 128 stationary queries
// per CU (all the 1024-wide fragments the register file holds at 32 queries per wave) the MFMAs of a block take 2048 cycles per
// SIMD at full rate -- 32 B/clk/CU of L2 -> LDS fill, above what a CU's LDS-DMA path sustains (~28 B/clk): scan_kernel_v3's 4-wave
// configuration and scan_kernel_ks both sit at 0.45-0.47 of the MFMA peak whatever the wave schedule.  The cure is more flops
// per streamed byte: a wave PAIR splits K (512 columns each, as in scan_kernel_ks) but shares 48 queries instead of 32 -- three
// 16-query column blocks x 16 k32-steps = 48 fragments = 192 registers per wave, what scan_kernel_v4 carries at pitch 768.
// 4 pairs = 192 queries per workgroup: one A fragment feeds THREE MFMAs (a third less LDS read per flop as well), a block is
// 3072 MFMA cycles per SIMD, the fill needed drops to 21 B/clk/CU and the ring's latency budget grows by half.
//   * the third accumulator set is paid for with sub-lists of KL = 4 (8 sub-lists per (query, split): 2 document halves x 4 lane
//     groups); what a full sub-list drops is bounded by its last entry, the margin check decides per query, flagged queries are
//     settled exactly (resolve_kernels.hpp) -- results are the oracle's bits like every other kernel's;
//   * class words: ONE copy per pair in LDS (1.5 KiB = 48 queries x 8 words; the role-0 wave refreshes it, both read it: a
//     stale word is merely a weaker bound), which is what lets ring + copies + exchange slots fit 160 KiB;
//   * query tiles are 192 wide: the host pads the staged queries / insert bounds / lists to whole tiles (mips_hip.hip).
// Everything else -- ring, exchange of the foreign half's partial sums through LDS, split block barrier -- is scan_kernel_ks's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"
#include "scan_kernel_v3.hpp"
#include "scan_kernel_v4.hpp"

namespace mips {

// PUB: every sub-list publishes (vouches for) its PUB-th best entry, so the 8 class words of a query stand for 8 PUB documents:
// the pool selected from the lists may be 8 PUB deep (8 / 16 / 32 candidates: mips_hip.hip, "optimistic" pools and the first
// stage of the fp32-exact search at row pitch 1024), as scan_kernel_v4's PUB.
template <int KL, int KS32, int AD, int TIMING_MODE = 0, int PUB = 1>
__global__ __launch_bounds__(512, 2) void scan_kernel_k3(ScanArgs p) {
    static_assert(PUB >= 1 && PUB <= KL, "a sub-list vouches for one of its own entries");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WAVES = 8;
    constexpr int PAIRS = 4;
    constexpr int NCB = 3;                          // 16-query column blocks per pair
    constexpr int TN = PAIRS * 16 * NCB;            // 192 queries per workgroup
    constexpr int STAGES = 2;
    constexpr int STAGE_BYTES = V3_DB * KS32 * 64;  // 32 rows x (32 KS32) k x 2 B = 64 KiB at pitch 1024
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int PPW = PIECES / WAVES;
    constexpr int KH = KS32 / 2;                    // k32-steps of one K half
    static_assert(KS32 % 4 == 0 && PIECES % WAVES == 0, "K halves must be whole 64-k slabs, DMA shares whole pieces");
    static_assert(2 * (PIECES / WAVES) <= KS32 / 2, "the DMA pieces are issued inside the foreign half, one per two steps");
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
    const bool pf_leader = (j0 % p.qt_per_group) == 0;
    // TIMING_MODE 10 (diagnostic, same results): shader-clock cycles this wave spends in the DMA wait of its arrival, in the block
    // barrier's poll and in the pair's poll, summed over the launch into the five words p.nq_dev points at (the launcher passes a
    // scratch buffer there: tools/ab.py prints them)
    unsigned long long t_vm = 0ull, t_bar = 0ull, t_pair = 0ull;
    const unsigned long long t_start = TIMING_MODE == 10 ? __builtin_readcyclecounter() : 0ull;
    if (p.spin_limit < 0 && tid == 0) *p.err = 1u; // test-only: force the scan-error path (include/mips_hip.h, "spin_limit")
    const bool idle_pair = (qt * TN + pair * 16 * NCB) >= p.nq; // all 48 queries of the pair are padding (scan_kernel_v3.hpp)

    const int b0 = split * p.tiles_per_split;
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // ---- stationary fragments of this wave's K half: lane holds Q[q0 + 16 n + c][32 (KH role + j) + 8 g .. +8)
    bf16x8 bq[NCB][KH];
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
        const uint16_t* qrow = p.qbuf + ((int64_t)qt * TN + pair * 16 * NCB + n * 16 + c) * p.ld + 32 * KH * role + 8 * g;
#pragma unroll
        for (int j = 0; j < KH; ++j) bq[n][j] = *reinterpret_cast<const bf16x8*>(qrow + 32 * j);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int j = 0; j < KH; ++j) asm volatile("" : "+v"(bq[n][j]));
#endif
    }

    float ls[NCB][KL];
    int li[NCB][KL];
    float thr[NCB];
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
        thr[n] = -INFINITY;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            ls[n][i] = -INFINITY;
            li[n][i] = IDX_NONE;
        }
    }

    // ---- LDS map: ring | ONE copy per pair of its class words (1.5 KiB: 48 queries x 8 words) | exchange slots (3 KiB per wave) | counters
    constexpr unsigned THR_PAIR = 16u * NCB * 32u;  // bytes of class words per pair, in LDS and in p.gthr
    constexpr unsigned THR_AREA = STAGES * STAGE_BYTES;
    constexpr unsigned XCH_AREA = THR_AREA + PAIRS * THR_PAIR;
    constexpr unsigned XCH_WAVE = NCB * 1024u;
    constexpr unsigned CNT_AREA = XCH_AREA + WAVES * XCH_WAVE;
    constexpr unsigned PF_AREA = CNT_AREA + 64;     // 256 B nobody reads: where the L2 prefetch of a later block lands
    constexpr bool PREFETCH = TIMING_MODE == 7 || TIMING_MODE == 8 || TIMING_MODE == 9;
    static_assert(PF_AREA + 256 <= 160 * 1024, "LDS budget");
    // p.gthr = [query tile][pair][48 queries][8 words]: both waves of a pair publish into the same 1.5 KiB
    const __amdgpu_buffer_rsrc_t thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * (PAIRS * THR_PAIR)), 0, PAIRS * THR_PAIR, 0x00020000);
    if (role == 0) {
        *reinterpret_cast<uint4*>(smem + THR_AREA + pair * THR_PAIR + lane * 16u) = make_uint4(0u, 0u, 0u, 0u);
        if (lane < 32) *reinterpret_cast<uint4*>(smem + THR_AREA + pair * THR_PAIR + 1024u + lane * 16u) = make_uint4(0u, 0u, 0u, 0u);
    }
    auto refresh_thresholds = [&]() { // role 0 only: 1 KiB + 512 B (the second operation with lanes 0 .. 31 active)
        if (role != 0) return;
        const unsigned ln = lane_id_here();
        __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, (lds_void*)(smem + THR_AREA + pair * THR_PAIR), 16,
                                                 (unsigned)pair * THR_PAIR + ln * 16u, 0, 0, 16);
        if (ln < 32u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, (lds_void*)(smem + THR_AREA + pair * THR_PAIR + 1024u), 16,
                                                     (unsigned)pair * THR_PAIR + 1024u + ln * 16u, 0, 0, 16);
    };

    // ---- LDS-DMA map (as scan_kernel_v4): piece pc = slab * 4 + rg, 8 rows x 128 B
    const unsigned char* docs_b = reinterpret_cast<const unsigned char*>(p.docs);
    const int64_t row_bytes = (int64_t)p.ld * 2;
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)(V3_DB * row_bytes), 0x00020000);
        const int pc = wave + WAVES * i;
        const int slab = pc >> 2, rg = pc & 3;
        const unsigned ln = lane_id_here(); // (opaque: re-derived per piece, nothing lane-dependent lives across the MFMA chain)
        const unsigned lane_off0 = (ln >> 3) * (unsigned)(p.ld * 2) + (((ln & 7u) ^ ((ln >> 4) & 7u)) << 4);
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
    // (the counter addresses are wave-uniform: they live in SGPRs and are moved into a VGPR inside the asm that uses them -- this
    // kernel has no VGPR to keep them in across the MFMA chain)
    const unsigned cnt_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_void*)(smem + CNT_AREA));
    const unsigned xcnt_mine = cnt_lds + 16u + 4u * (unsigned)wave;
    const unsigned xcnt_partner = cnt_lds + 16u + 4u * (unsigned)(wave ^ 1);
    unsigned arrivals_needed = 0;
    auto bump = [&](unsigned saddr) { // ds_add_u32 [saddr], 1 by lane 0
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned ta, tb;
        if (lane_id_here() == 0u)
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, 1\n\tds_add_u32 %0, %1" : "=&v"(ta), "=&v"(tb) : "s"(saddr) : "memory");
#endif
    };
    auto arrive = [&](bool in_loop = true) {
        // 2-stage ring: this wave's share of the NEXT block has landed (the one younger operation that may still be in flight is
        // the block's L2 prefetch, issued after the pieces)
        const unsigned long long t0 = TIMING_MODE == 10 ? __builtin_readcyclecounter() : 0ull;
        if (PREFETCH && in_loop) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (TIMING_MODE == 10) t_vm += __builtin_readcyclecounter() - t0;
        bump(cnt_lds);
    };
    // One dword of each 128-B line of a block PF_DIST blocks ahead, 64 lines per wave, into the dummy area: the L2 miss of that
    // line is taken here, 2+ blocks before its DMA piece is issued, instead of inside the 2-stage ring's one-block window.
    constexpr int PF_DIST = TIMING_MODE == 8 ? 6 : 3;
    auto prefetch_block = [&](const unsigned char* blk_base) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)(V3_DB * row_bytes), 0x00020000);
        // TIMING_MODE 9: only ONE of the workgroups that share a document stream (the first query tile of its XCD group) touches
        // memory; the others issue the same operation out of range (no access, zeros into the dummy area; uniform vmcnt arithmetic)
        const unsigned voff = (TIMING_MODE != 9 || pf_leader) ? lane_id_here() * 128u : (0x40000000u | (lane_id_here() * 4u));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + PF_AREA), 4, voff, wave * 8192, 0, 0);
    };
    auto poll = [&](unsigned addr, unsigned need) {
        for (int spin = 0;; ++spin) {
            unsigned v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("v_mov_b32 %0, %1\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "s"(addr) : "memory");
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
    auto epilogue_half = [&](f32x4 (&acc)[NCB], int blk) {
        if (TIMING_MODE == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
#endif
            return;
        }
        const float mx0 = fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3]));
        const float mx1 = fmaxf(fmaxf(acc[1][0], acc[1][1]), fmaxf(acc[1][2], acc[1][3]));
        const float mx2 = fmaxf(fmaxf(acc[2][0], acc[2][1]), fmaxf(acc[2][2], acc[2][3]));
        if (__ballot(mx0 > thr[0] || mx1 > thr[1] || mx2 > thr[2]) != 0ull) {
            const unsigned ln = lane_id_here();
            const int base = blk * V3_DB + 16 * role + 4 * (int)(ln >> 4);
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
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
                    const unsigned cls = (4u * (unsigned)split + (ln >> 4)) & 7u;
                    publish_umax(thr_encode(ls[n][PUB - 1]), (unsigned)pair * THR_PAIR + (16u * n + (ln & 15u)) * 32u + 4u * cls, thr_rsrc);
                }
            }
        }
    };

    auto block = [&](bool refresh, int blk, int stage, const unsigned char* nbase, int nstage, const unsigned char* pbase) {
        // pieces of the NEXT block first: their stage was released by the barrier just passed
        if (refresh) refresh_thresholds();
        if (idle_pair) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) issue_piece(nbase, nstage, i);
            if (PREFETCH) prefetch_block(pbase);
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
        f32x4 acc[2][NCB];
#pragma unroll
        for (int hpos = 0; hpos < 2; ++hpos) {
#pragma unroll
            for (int n = 0; n < NCB; ++n) acc[hpos][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < KH; ++j) {
                const int t = hpos * KH + j;
#pragma unroll
                for (int n = 0; n < NCB; ++n) acc[hpos][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[t % AD], bq[n][j], acc[hpos][n], 0, 0, 0);
                if (t + AD < CHAIN) ar[t % AD] = lds_frag(t + AD);
                // The pieces of the next block: early (they must have landed by the end of this block), but NOT by both waves of a
                // SIMD at once -- the two partners (waves w and w + 4) pass the barrier together, and a piece costs its wave 100+ cycles
                // of issue that only the partner's MFMAs can cover (the diagnostic build without the DMA, TIMING_MODE 2, runs 17 %
                // faster: profiles/r3_pitch1024).  One piece every other step of the foreign half, the partners on alternate steps.
                // The pieces are issued UNCONDITIONALLY (the last block re-fetches itself, see the main loop): with a second branch
                // per site the 256-VGPR allocation spills a query fragment and reloads it in every block.
                // TIMING_MODE 4 / 5 / 6 (diagnostics, same results): every wave in steps 0 .. PPW - 1 (the first build) / waves
                // 0 .. 3 in steps 0 .. PPW - 1 and waves 4 .. 7 in steps PPW .. 2 PPW - 1 / every wave on the even steps.
                if (PREFETCH && hpos == 1 && j == 0) prefetch_block(pbase);
                if (TIMING_MODE == 4) {
                    if (hpos == 0 && j < PPW) issue_piece(nbase, nstage, j);
                } else if (TIMING_MODE == 5) {
                    if (hpos == 0 && j < 2 * PPW && (j / PPW) == (wave >> 2)) issue_piece(nbase, nstage, j % PPW);
                } else if (TIMING_MODE == 6) {
                    if (hpos == 0 && j < 2 * PPW && (j & 1) == 0) issue_piece(nbase, nstage, j >> 1);
                } else if (TIMING_MODE != 2) {
                    if (hpos == 0 && j < 2 * PPW && ((j ^ (wave >> 2)) & 1) == 0) issue_piece(nbase, nstage, j >> 1); // (this spelling allocates without a spill: tools/kernel_regs.py)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (hpos == 0) {
                // hand the foreign half's partial sums to the partner: 2 KiB slot [n][lane][4], then the counter (a
                // wave's LDS operations complete in order)
                unsigned char* slot = smem + XCH_AREA + wave * XCH_WAVE + lane_id_here() * 16u;
#pragma unroll
                for (int n = 0; n < NCB; ++n) *reinterpret_cast<f32x4*>(slot + 1024 * n) = acc[0][n];
                if (TIMING_MODE != 3) bump(xcnt_mine); // (TIMING_MODE 3: diagnostic build without the pair's counter hand-shake)
            }
        }
        // the partner's partial sums for MY half
        {
            const unsigned long long t0 = TIMING_MODE == 10 ? __builtin_readcyclecounter() : 0ull;
            if (TIMING_MODE != 3) poll(xcnt_partner, (unsigned)(blk - b0) + 1u);
            if (TIMING_MODE == 10) t_pair += __builtin_readcyclecounter() - t0;
        }
        f32x4 own[NCB];
        {
            const unsigned char* slot = smem + XCH_AREA + (wave ^ 1) * XCH_WAVE + lane_id_here() * 16u;
#pragma unroll
            for (int n = 0; n < NCB; ++n) own[n] = acc[1][n] + *reinterpret_cast<const f32x4*>(slot + 1024 * n); // (own K half) + (the partner's)
        }
        arrive(); // chain done, the partner's slot read, this wave's share of the next block landed
        if (refresh && TIMING_MODE == 0) {
            // minimum of the 8 class words of queries c, 16 + c and 32 + c (what an earlier refresh of the pair's copy brought, or 0)
            const unsigned a0 = (unsigned)(size_t)(lds_void*)smem + THR_AREA + pair * THR_PAIR + (lane_id_here() & 15u) * 32u;
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
                u32x4 w0 = {0u, 0u, 0u, 0u}, w1 = w0;
#if defined(__HIP_DEVICE_COMPILE__)
                if (n == 0) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
                else if (n == 1) asm volatile("ds_read_b128 %0, %2 offset:512\n\tds_read_b128 %1, %2 offset:528\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
                else asm volatile("ds_read_b128 %0, %2 offset:1024\n\tds_read_b128 %1, %2 offset:1040\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
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
#pragma unroll
                    for (int n = 0; n < NCB; ++n) own[n][r] = -INFINITY;
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
    if (nb > 0) arrive(false); // this wave's share of block 0 has landed
    for (int i = 0; i < nb; ++i) {
        arrivals_needed += WAVES;
        {
            const unsigned long long t0 = TIMING_MODE == 10 ? __builtin_readcyclecounter() : 0ull;
            poll(cnt_lds, arrivals_needed); // every share of block i landed; everyone is done with block i - 1
            if (TIMING_MODE == 10) t_bar += __builtin_readcyclecounter() - t0;
        }
        // (the last block of the range re-fetches ITSELF into the free stage instead of a next block: the pieces are issued
        // unconditionally, the MFMA chain stays one basic block)
        const unsigned char* nbase = first + (int64_t)(i + 1 < nb ? i + 1 : i) * blk_bytes;
        const int ipf = i + PF_DIST < nb ? i + PF_DIST : nb - 1;
        block(i < 8 || (i & 7) == 0, b0 + i, i & 1, nbase, (i + 1) & 1, first + (int64_t)ipf * blk_bytes);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (TIMING_MODE == 10 && p.nq_dev != nullptr && lane_id_here() == 0u && !idle_pair) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(const_cast<int*>(p.nq_dev));
        atomicAdd(dbg + 0, t_vm);
        atomicAdd(dbg + 1, t_bar);
        atomicAdd(dbg + 2, t_pair);
        atomicAdd(dbg + 3, (unsigned long long)(__builtin_readcyclecounter() - t_start));
        atomicAdd(dbg + 4, (unsigned long long)nb);
        atomicAdd(dbg + 5, 1ull);
    }

    // lists: [q][nsplit][8 = 2 document halves x 4 lane groups][KL]
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
        const unsigned ln = lane_id_here(); // (re-derived: c and g do not survive the main loop in a register)
        const int q = qt * TN + pair * 16 * NCB + n * 16 + (int)(ln & 15u);
        const size_t o = (((size_t)q * p.nsplit + split) * 8 + role * 4 + (int)(ln >> 4)) * KL;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            p.part_s[o + i] = ls[n][i];
            p.part_i[o + i] = li[n][i];
        }
    }
}

} // namespace mips
