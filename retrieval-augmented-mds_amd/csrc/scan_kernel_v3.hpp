// scan_kernel_v3: query-stationary fused score + top-K scan (the fast path for d <= 768).
//
// This problem is a GEMM with a tiny K (d = 768) and an enormous M (the index): instead of tiling it
// like a square GEMM, every wave keeps the MFMA B-fragments of ITS 32 queries for the WHOLE K in
// registers (KS16 = d/16 fragments x 4 VGPRs = 192 VGPRs at d = 768) for the lifetime of the
// workgroup.  8 waves = 256 stationary queries per workgroup.  Only documents move:
//   HBM/L2 -> LDS  by global_load_lds_dwordx4 (LDS-DMA), 32-document blocks (32 x d x 2 B = 48 KiB) in
//                  a 3-deep ring, two blocks ahead, completion by counted s_waitcnt vmcnt + raw s_barrier;
//   LDS -> MFMA    one ds_read_b128 (A fragment: 32 docs x 16 k) per v_mfma_f32_32x32x16_bf16.
// Per CU and per 32-document block this is 48 KiB of fill for 2 x 48 x 32 = 3072 MFMA cycles per SIMD
// = 16 B/clk, half of what a 256 x 256 GEMM tile needs and inside what the L2 -> LDS path of a CU
// sustains (~28 B/clk); queries are never re-read and never occupy L2 during the scan.
//
// Epilogue per block (same contract as scan_kernel.hpp): the 16 accumulator registers of a lane are 16
// documents scored against the lane's query; a max pre-test against the lane's current K-th best
// guards the rare sorted insert.  Two waves share a SIMD, so one wave's epilogue VALU work runs under
// the other wave's MFMAs.
//
// LDS image of a block: 64-k slabs of [32 rows][128 B]; a DMA piece (one wave-instruction, 1 KiB) is
// 8 rows x 128 B, full cache lines on the global side.  Chunk c of row r is stored at slot
// c ^ ((r >> 1) & 7) (applied to the SOURCE address; the read applies the same XOR), which makes the
// ds_read_b128 of 32 rows x one chunk conflict-free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"

namespace mips {

constexpr int V3_TN = 256;     // queries per workgroup
constexpr int V3_DB = 32;      // documents per block
constexpr int V3_STAGES = 3;
// NQB = 32-query blocks per wave: 1 -> 8 waves x 32 queries, two waves per SIMD (256 registers each);
//                                 2 -> 4 waves x 64 queries, one wave per SIMD (512 registers).
template <int NQB> struct V3Cfg {
    static constexpr int WAVES = 8 / NQB;
    static constexpr int THREADS = 64 * WAVES;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int KL, int KS16, int NQB, int AD>
__global__ __launch_bounds__(V3Cfg<NQB>::THREADS, NQB == 1 ? 2 : 1) void scan_kernel_v3(ScanArgs p) {
    constexpr int WAVES = V3Cfg<NQB>::WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE_BYTES = V3_DB * KS16 * 32; // 32 rows x (KS16 * 16) k x 2 B
    constexpr int PIECES = KS16;                   // 1 KiB pieces per block
    constexpr int PPW = PIECES / WAVES;            // pieces per wave
    static_assert(PIECES % WAVES == 0, "every wave must issue the same number of DMA pieces (vmcnt accounting)");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int l31 = lane & 31;

    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int qt = (xcd % p.qgroups) + p.qgroups * (j % p.qt_per_group);
    const int split = (xcd / p.qgroups) * p.splits_per_group + j / p.qt_per_group;
    if (qt >= p.nqt) return;

    const int b0 = split * p.tiles_per_split; // "tiles" are 32-document blocks here
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // ---- stationary query fragments: lane holds Q[q0 + l31][16 s + 8 h .. +8) for every k16-step s
    bf16x8 bq[NQB][KS16];
#pragma unroll
    for (int n = 0; n < NQB; ++n) {
        const uint16_t* qrow = p.qbuf + ((int64_t)qt * V3_TN + (wave * NQB + n) * 32 + l31) * p.ld + 8 * h;
#pragma unroll
        for (int s = 0; s < KS16; ++s) bq[n][s] = *reinterpret_cast<const bf16x8*>(qrow + 16 * s);
        // Consume the fragments here: the compiler's wait for these ordinary loads then sits BEFORE the
        // pipeline instead of inside the loop (where a vmcnt(0) would drain the LDS-DMA queue every
        // block), and the values stay opaque register residents.
#pragma unroll
        for (int s = 0; s < KS16; ++s) asm volatile("" : "+v"(bq[n][s]));
    }

    float ls[NQB][KL];
    int li[NQB][KL];
#pragma unroll
    for (int n = 0; n < NQB; ++n)
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            ls[n][i] = -INFINITY;
            li[n][i] = IDX_NONE;
        }

    // ---- LDS-DMA map: piece pc = slab * 4 + rg covers rows 8 rg .. 8 rg + 7 of 64-k slab `slab`;
    // lane -> row 8 rg + (lane >> 3), slot lane & 7, source chunk slot ^ ((row >> 1) & 7)
    // = slot ^ ((4 rg + (lane >> 4)) & 7): depends on rg only through rg & 1.
    const int lrow = lane >> 3;
    const unsigned lane_off0 = (unsigned)(lrow * p.ld * 2 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
    const unsigned lane_off1 = (unsigned)(lrow * p.ld * 2 + (((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4));
    const unsigned char* docs_b = reinterpret_cast<const unsigned char*>(p.docs);
    const int64_t row_bytes = (int64_t)p.ld * 2;

    auto issue = [&](int blk, int stage) {
        const unsigned char* bbase = docs_b + (int64_t)blk * V3_DB * row_bytes;
        unsigned char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int pc = wave + WAVES * i;
            const int slab = pc >> 2, rg = pc & 3;
            const unsigned char* src = bbase + (int64_t)rg * 8 * row_bytes + slab * 128 + ((rg & 1) ? lane_off1 : lane_off0);
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sbase + pc * 1024), 16, 0, 0);
        }
    };

    const int rd_row = l31 * 128;
    const int rd_swz = (l31 >> 1) & 7;

    auto block = [&](int blk, int stage) {
        const unsigned char* sa = smem + stage * STAGE_BYTES + rd_row;
        auto lds_frag = [&](int s) {
            return *reinterpret_cast<const bf16x8*>(sa + (s >> 2) * 4096 + (((2 * (s & 3) + h) ^ rd_swz) << 4));
        };
        f32x16 acc[NQB];
#pragma unroll
        for (int n = 0; n < NQB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        // A fragments run AD k-steps ahead of the MFMAs that consume them: an explicit ring, with a
        // sched_barrier after every step so the machine scheduler cannot sink each ds_read back next to
        // its consumer (which it does to save registers, exposing the full LDS latency per MFMA).
        bf16x8 ar[AD];
#pragma unroll
        for (int s = 0; s < AD; ++s) ar[s] = lds_frag(s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS16; ++s) {
#pragma unroll
            for (int n = 0; n < NQB; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[s % AD], bq[n][s], acc[n], 0, 0, 0);
            if (s + AD < KS16) ar[s % AD] = lds_frag(s + AD);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int base = blk * V3_DB + 4 * h;
        if ((int64_t)(blk + 1) * V3_DB > p.ntotal) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((int64_t)(base + (r & 3) + 8 * (r >> 2)) >= p.ntotal) {
#pragma unroll
                    for (int n = 0; n < NQB; ++n) acc[n][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int n = 0; n < NQB; ++n) {
            float mx = acc[n][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[n][r]);
            if (__ballot(mx > ls[n][KL - 1]) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = acc[n][r];
                    if (s > ls[n][KL - 1]) list_insert<KL>(ls[n], li[n], s, base + (r & 3) + 8 * (r >> 2));
                }
            }
        }
    };

    // block i lives in ring stage i % 3 and is issued two blocks ahead
    if (nb > 0) issue(b0, 0);
    if (nb > 1) issue(b0 + 1, 1);
    int stage = 0, pstage = 2;
    for (int i = 0; i < nb; ++i) {
        // this wave's pieces of block i have landed once at most block i+1's remain in flight
        if (i + 1 < nb) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier(); // all shares of block i landed; everyone is done with block i-1
        if (i + 2 < nb) issue(b0 + i + 2, pstage);
        block(b0 + i, stage);
        stage = stage == 2 ? 0 : stage + 1;
        pstage = pstage == 2 ? 0 : pstage + 1;
    }

#pragma unroll
    for (int n = 0; n < NQB; ++n) {
        const int q = qt * V3_TN + (wave * NQB + n) * 32 + l31;
        const size_t o = (((size_t)q * p.nsplit + split) * 2 + h) * KL;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            p.part_s[o + i] = ls[n][i];
            p.part_i[o + i] = li[n][i];
        }
    }
}

} // namespace mips
