// scan_kernel_f8x: the fp8 (e4m3) scan on v_mfma_f32_16x16x128_f8f6f4 with 64-document blocks -- the fast
// path of BASELINE config 5 for k <= 5 and row pitches up to 768 bytes (scan_kernel_f8 keeps the rest).
//
// Two things it changes against scan_kernel_f8, both measured first on bare loops (tools/mfma_ceiling.hip,
// random e4m3 bytes, LDS-fed): the 16x16x128 shape sustains 3.78-3.93 PFLOP/s where 32x32x64 sustains 3.34
// (the chip holds a higher clock on it), and scan_kernel_f8 reached only 77 % of its shape's ceiling because a
// 32-document fp8 block is just 12 MFMAs per wave, so the per-block costs (arrival counter, epilogue, first
// LDS round trip) weigh twice what they do in the bf16 kernels.  Here a block is 64 documents (48 KiB at
// d = 768, the bf16 kernels' stage size): 48 MFMAs per wave per block, one epilogue over 32 accumulators.
//
// Mapping (16x16x128: lane l -> c = l & 15, g = l >> 4; the lane supplies the 32 k-bytes [128 s + 32 g, +32)
// of row c for A and of column c for B -- the instruction pairs A and B bytes position by position, so any k
// assignment used on both sides gives the full dot product; C/D col = c, row = 4 g + reg):
//   * a wave owns 32 stationary queries = two 16-query column blocks n = 0, 1: 8 VGPRs per (n, k-step),
//     96 at d = 768;
//   * a block is four 16-document tiles; one A fragment (two ds_read_b128) feeds two MFMAs; the 32
//     accumulator registers of a lane are documents 16 tl + 4 g .. + 3 (tl = 0..3) against queries c, 16 + c;
//   * lists, shared insert bounds, ring, counted waits, split barrier: as scan_kernel_v4 (4 sub-lists of KL per
//     (query, split), class word (4 split + g) & 7, sparse re-read, dummy refresh).
// LDS image: one 128-byte slab per k-step, [64 rows][128 B]; 16-byte chunk ch of row r at slot
// ch ^ ((r >> 1) & 7), filled by 8-row x 128-B LDS-DMA pieces whose per-lane SOURCE address carries the swizzle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"
#include "scan_kernel_f8.hpp"
#include "scan_kernel_v3.hpp"
#include "scan_kernel_v4.hpp"

namespace mips {

constexpr int F8X_DB = 64; // documents per block

// NT_DOCS: non-temporal document DMA for searches of ONE query tile (every block has a single reader)
template <int KL, int LD, int AD, int TIMING_MODE = 0, bool NT_DOCS = false>
__global__ __launch_bounds__(512, 2) void scan_kernel_f8x(ScanArgsF8 pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ScanArgs& p = pa.c;
    constexpr int WAVES = 8;
    constexpr int TN = WAVES * 32;
    constexpr int STAGES = 3;
    constexpr int KS = LD / 128;                 // MFMA k-steps per document tile
    constexpr int TILES = F8X_DB / 16;           // 16-document tiles per block
    constexpr int STEPS = TILES * KS;            // A fragments per block
    constexpr int STAGE_BYTES = F8X_DB * LD;     // 64 rows x LD bytes
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int PPW = PIECES / WAVES;
    static_assert(LD % 128 == 0 && PIECES % WAVES == 0, "row length must be a multiple of 128 bytes");
    static_assert(STEPS % PPW == 0, "the DMA pieces are spread evenly over the chain");
    static_assert(KL <= 8, "8 class words vouch for 8 documents");
    static_assert(STAGES * STAGE_BYTES + WAVES * 1024 + 1024 + 16 <= 160 * 1024, "ring does not fit the LDS");

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

    const int b0 = split * p.tiles_per_split; // "tiles" are 64-document blocks here
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // ---- stationary query fragments: lane holds bytes [128 s + 32 g, +32) of query q0 + 16 n + c
    v8i32 bq[2][KS];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint8_t* qrow = pa.qbuf + ((int64_t)qt * TN + wave * 32 + n * 16 + c) * LD + 32 * g;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i32 lo = *reinterpret_cast<const v4i32*>(qrow + 128 * s);
            const v4i32 hi = *reinterpret_cast<const v4i32*>(qrow + 128 * s + 16);
            bq[n][s] = v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(bq[n][s]));
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
    const unsigned thr_addr = THR_AREA + wave * THR_WAVE + lane * 16; // this lane's DMA chunk = LDS slot = voffset
    *reinterpret_cast<uint4*>(smem + thr_addr) = make_uint4(0u, 0u, 0u, 0u);
    const __amdgpu_buffer_rsrc_t thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * (WAVES * THR_WAVE) - (int64_t)THR_AREA), 0,
        (int)(THR_AREA + WAVES * THR_WAVE), 0x00020000);
    auto refresh_thresholds = [&](bool real) { // !real: out-of-range dummy into the dump area (uniform vmcnt count)
        lds_void* dst = (lds_void*)(smem + (real ? THR_AREA + wave * THR_WAVE : DUMP_AREA));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, dst, 16, real ? thr_addr : (thr_addr | 0x40000000u), 0, 0, 16);
    };

    // ---- LDS-DMA map: piece pc = slab * 8 + rg covers rows 8 rg .. 8 rg + 7 of the 128-byte slab `slab`;
    // lane -> row 8 rg + (lane >> 3), slot lane & 7, source chunk slot ^ ((row >> 1) & 7)
    // = slot ^ ((4 rg + (lane >> 4)) & 7): depends on rg only through rg & 1 (byte offset ^ 64)
    const unsigned lane_off0 = (unsigned)((lane >> 3) * LD + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
    constexpr int64_t blk_bytes = (int64_t)F8X_DB * LD;
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)blk_bytes, 0x00020000);
        const int pc = wave + WAVES * i;
        const int slab = pc >> 3, rg = pc & 7;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + stage * STAGE_BYTES + pc * 1024), 16,
                                                 (rg & 1) ? (lane_off0 ^ 64u) : lane_off0, rg * 8 * LD + slab * 128, 0, NT_DOCS ? 2 : 0);
    };
    auto issue = [&](const unsigned char* blk_base, int stage) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(blk_base, stage, i);
    };

    // ---- A-fragment read address: row 16 tl + c of slab s, chunks 2 g and 2 g + 1 at slots chunk ^ ((c >> 1) & 7)
    const int rd0 = c * 128 + (((2 * g) ^ ((c >> 1) & 7)) << 4); // the second chunk is this ^ 16

    // ---- split barrier (see scan_kernel_v3.hpp)
    const unsigned cnt_lds = (unsigned)(size_t)(lds_void*)(smem + DUMP_AREA + 1024);
    unsigned arrivals_needed = 0;
    constexpr int PER_BLOCK = PPW + 1;
    auto arrive = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_BLOCK) : "memory");
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

    auto block = [&](bool refresh, int blk, int stage, const unsigned char* pbase, int pstage) {
        if (idle_wave) { // all of this wave's queries are padding (scan_kernel_v3.hpp): bring the documents, skip the arithmetic
            refresh_thresholds(false);
            issue(pbase, pstage);
            arrive();
            return;
        }
        const unsigned char* sa = smem + stage * STAGE_BYTES;
        // flattened step t = tl * KS + s: tile tl = rows 16 tl .. 16 tl + 15 (2048 B further down a slab), slab s
        auto lds_frag = [&](int t) {
            const int o = (t / KS) * 2048 + (t % KS) * (F8X_DB * 128);
            const v4i32 lo = *reinterpret_cast<const v4i32*>(sa + o + rd0);
            const v4i32 hi = *reinterpret_cast<const v4i32*>(sa + o + (rd0 ^ 16));
            return v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        f32x4 acc[TILES][2];
#pragma unroll
        for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tl][n][r] = 0.f;
        v8i32 ar[AD];
#pragma unroll
        for (int t = 0; t < AD; ++t) ar[t] = lds_frag(t);
        refresh_thresholds(refresh); // first VMEM op of the block
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            const int tl = t / KS, s = t % KS;
            acc[tl][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ar[t % AD], bq[0][s], acc[tl][0], 0, 0, 0, 0, 0, 0);
            acc[tl][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ar[t % AD], bq[1][s], acc[tl][1], 0, 0, 0, 0, 0, 0);
            if (t + AD < STEPS) ar[t % AD] = lds_frag(t + AD);
            if ((t % (STEPS / PPW)) == (STEPS / PPW) / 2) issue_piece(pbase, pstage, t / (STEPS / PPW));
            __builtin_amdgcn_sched_barrier(0);
        }
        arrive(); // chain done, share of the next block landed; the epilogue below runs un-synchronised
        if (TIMING_MODE == 1) { // diagnostic build (results are wrong): no epilogue at all
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) asm volatile("" ::"v"(acc[tl][0]), "v"(acc[tl][1]));
#endif
            return;
        }
        const int base = blk * F8X_DB + (int)((thr_addr >> 6) & 12u); // + 4 g, from the lane bits of thr_addr
        if ((int64_t)(blk + 1) * F8X_DB > p.ntotal) { // last block of the index only
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((int64_t)(base + 16 * tl + r) >= p.ntotal) {
                        acc[tl][0][r] = -INFINITY;
                        acc[tl][1][r] = -INFINITY;
                    }
        }
        if (refresh) { // minimum of the 8 class words of queries c and 16 + c (inline asm: see scan_kernel_v3.hpp)
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
        float mx[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            mx[n] = fmaxf(fmaxf(acc[0][n][0], acc[0][n][1]), fmaxf(acc[0][n][2], acc[0][n][3]));
#pragma unroll
            for (int tl = 1; tl < TILES; ++tl)
                mx[n] = fmaxf(mx[n], fmaxf(fmaxf(acc[tl][n][0], acc[tl][n][1]), fmaxf(acc[tl][n][2], acc[tl][n][3])));
        }
        if (__ballot(mx[0] > thr[0] || mx[1] > thr[1]) != 0ull) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float mark = ls[n][0];
#pragma unroll
                for (int tl = 0; tl < TILES; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s = acc[tl][n][r];
                        if (s > thr[n]) {
                            list_insert<KL>(ls[n], li[n], s, base + 16 * tl + r);
                            thr[n] = fmaxf(thr[n], ls[n][KL - 1]);
                        }
                    }
                if (ls[n][0] > mark) { // new best of this sub-list: raise its class word, (4 split + g) & 7
                    const unsigned cls = (4u * (unsigned)split + ((thr_addr >> 8) & 3u)) & 7u;
                    publish_umax(thr_encode(ls[n][0]), (thr_addr & ~0x3FFu) + ((thr_addr & 0xF0u) << 1) + 512u * n + 4u * cls, thr_rsrc);
                }
            }
        }
    };

    const unsigned char* docs_b = pa.docs;
    const unsigned char* first = docs_b + (int64_t)b0 * blk_bytes;
    const unsigned char* last = docs_b + (int64_t)(b1 - 1) * blk_bytes;
    constexpr int AHEAD = STAGES - 1;
    if (nb > 0) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) { // same operation sequence as AHEAD steady-state blocks (vmcnt arithmetic)
            refresh_thresholds(true);
            issue(a < nb ? first + a * blk_bytes : last, a);
        }
    }
    const unsigned char* pbase = nb > AHEAD ? first + AHEAD * blk_bytes : last;
    int stage = 0, pstage = AHEAD;
    if (tid == 0) *reinterpret_cast<unsigned*>(smem + DUMP_AREA + 1024) = 0u;
    __syncthreads(); // the one real barrier: arrival counter initialised
    if (nb > 0) arrive();
    for (int i = 0; i < nb; ++i) {
        wait_all();
        block(i < 8 || (i & 7) == 0, b0 + i, stage, pbase, pstage); // refresh schedule: scan_kernel_v3.hpp
        if (i + AHEAD + 1 < nb) pbase += blk_bytes;
        stage = stage == STAGES - 1 ? 0 : stage + 1;
        pstage = pstage == STAGES - 1 ? 0 : pstage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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
