// scan_kernel_e8: e4m3 DOCUMENTS x bf16 QUERIES -- BASELINE config 5 as it is worded ("fp8 e4m3 doc embeddings").
//
// The other fp8 kernels (scan_kernel_f8 / f8x) quantise the queries to e4m3 as well and run the fp8 MFMA at twice the bf16
// rate; here only the index is e4m3 -- half the HBM bytes of a bf16 index -- and the queries stay bf16: every product
// (e4m3 document element) x (bf16 query element) is exact in fp32, the MFMA accumulates in fp32, and the canonical score the
// results are re-scored with is the sequential fp64 sum of those exact products (oracle: search_exact on the decoded rows and the
// bf16-rounded queries).  It is meant for the regime where BYTES bind -- few queries per pass over the index -- so it is built
// around the conversion e4m3 -> bf16, which costs vector instructions (8 per 16 x 32 A fragment: 4 v_cvt_pk_f32_fp8 + 4
// v_perm_b32; e4m3 -> bf16 is exact), not around MFMA throughput:
//   * the 8 waves of a workgroup split K, not the queries: wave w multiplies columns [w K/8, (w + 1) K/8) of a 32-document
//     block against ALL the tile's queries (16 NCB <= 64), so every A fragment is read from LDS and converted ONCE per block
//     (a query-split kernel would convert it once per wave: 8 x the vector work, VALU-bound at a third of the HBM rate);
//   * the partial sums (2 document halves x NCB query blocks x 16 x 16 per wave) meet in LDS: every wave writes its 2 NCB
//     accumulator tiles to its slot, a block barrier, then wave w sums tile w over the 8 slots IN SLOT ORDER (the MFMA score is
//     the same number whoever asks) and runs scan_kernel_v4's top-K epilogue on it -- one sub-list of KL = 6 per lane, 8
//     sub-lists per (query, split): 2 halves x 4 lane groups;
//   * the ring holds RAW e4m3 blocks (32 rows x K bytes: 24 KiB at K = 768), so 3-4 stages are in flight per CU where a bf16
//     block of the same documents would be 48 KiB -- the point of the exercise; LDS image, XOR swizzle and LDS-DMA pieces are
//     scan_kernel_v4's with 128-BYTE slabs (128 k instead of 64); lane (c, g) reads the 8 bytes of row c, k = 32 s + 8 g with
//     one ds_read_b64 (conflict-free: 16-byte chunk slot = chunk ^ ((row >> 1) & 7), the two 8-byte halves side by side);
//   * counter barriers in LDS (inline asm, bounded polls).  PIPE (tiles of <= 32 queries, where two slot buffers fit): ONE barrier
//     per block -- block i is multiplied into slot buffer i & 1 while the reducing waves sum block i - 1 out of the other buffer;
//     all-arrived(i) = everybody's share of block i has landed, everybody's partial sums of block i - 1 are written, everybody is
//     done reading buffer i & 1 (the sums of block i - 2).  Otherwise two per block: A = landed + done with the previous block's
//     slots, B = partial sums written.
// Shared insert bounds: 8 class words per query, class (4 split + g) & 7, as scan_kernel_v4; every sub-list publishes its PUB-th
// best, so the words vouch for 8 PUB documents: PUB = 1 for pools of 8, 2 for pools of 10 / 16, 4 for pools of 32 (the margin
// check decides per query whether a pool selected from sub-lists of 6 was wide enough, as for scan_kernel_v4's optimistic pools).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"
#include "scan_kernel_v3.hpp"
#include "scan_kernel_v4.hpp"

namespace mips {

struct ScanArgsE8 {
    const uint8_t* docs;  // [capacity][ld] e4m3 bytes
    ScanArgs c;           // qbuf = bf16 queries [nq_pad][ld]; everything else as for the bf16 kernels
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 8 e4m3 bytes (two dwords, element j in byte j) -> 8 bf16 (exact)
__device__ __forceinline__ bf16x8 e4m3x8_to_bf16x8(unsigned lo, unsigned hi) {
    u32x4 w = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
    const f32x2 f01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), f23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const f32x2 f45 = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), f67 = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    // v_perm_b32: D = {S0.hi16, S1.hi16}: the bf16 of an e4m3 value is the upper half of its float32 (<= 4 significant bits)
    w[0] = __builtin_amdgcn_perm(__float_as_uint(f01[1]), __float_as_uint(f01[0]), 0x07060302u);
    w[1] = __builtin_amdgcn_perm(__float_as_uint(f23[1]), __float_as_uint(f23[0]), 0x07060302u);
    w[2] = __builtin_amdgcn_perm(__float_as_uint(f45[1]), __float_as_uint(f45[0]), 0x07060302u);
    w[3] = __builtin_amdgcn_perm(__float_as_uint(f67[1]), __float_as_uint(f67[0]), 0x07060302u);
#else
    (void)lo; (void)hi;
#endif
    bf16x8 r;
    __builtin_memcpy(&r, &w, 16);
    return r;
}

// KL: entries per sub-list; LDB: row pitch in bytes = padded K (256 .. 1024); NCB: 16-query column blocks per tile (1, 2 or 4);
// STAGES: ring depth; NT_DOCS: non-temporal document DMA (one query tile: every block has a single reader); PIPE: see above
// KW: K parts.  8: wave w multiplies K part w of BOTH 16-document halves (2 NCB partial tiles per wave).  4: wave w multiplies K
// part w & 3 of document half w >> 2 (NCB partial tiles per wave): the same fragments converted and the same MFMAs per wave, but
// half the partial sums through LDS and half the slot bytes -- which is what lets tiles of 64 queries keep two slot buffers
// (PIPE) beside a 3-stage ring.
template <int KL, int LDB, int NCB, int STAGES, bool NT_DOCS, int PUB = 1, bool PIPE = false, int KW = 8>
__global__ __launch_bounds__(512, 2) void scan_kernel_e8(ScanArgsE8 pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ScanArgs& p = pa.c;
    constexpr int WAVES = 8;
    constexpr int TN = 16 * NCB;                    // queries per workgroup
    static_assert(KW == 8 || KW == 4, "K parts");
    constexpr int KS = LDB / (32 * KW);             // k32-steps per wave (K / KW columns)
    constexpr int HPW = KW == 8 ? 2 : 1;            // 16-document halves per wave
    constexpr int STAGE_BYTES = V3_DB * LDB;        // 32 rows x K bytes
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int PPW = PIECES / WAVES;
    constexpr int TILES = 2 * NCB;                  // 16 x 16 accumulator tiles per block: (document half, query block)
    static_assert(LDB % 256 == 0 && LDB >= 256 && LDB <= 1024, "row pitch");
    static_assert(PIECES % WAVES == 0, "every wave issues the same number of DMA pieces");
    static_assert(TILES <= WAVES, "one reducing wave per accumulator tile");
    static_assert(KL <= 8 && STAGES >= 3 && PUB >= 1 && PUB <= KL, "");

    // ---- LDS map: ring | exchange slots [K part][tile][64 lanes][4 floats] | class words of the tile's queries (+ dump) | counters
    constexpr unsigned XCH_AREA = STAGES * STAGE_BYTES;
    constexpr unsigned XCH_WAVE = TILES * 1024u;                 // one K part's partial tiles
    constexpr unsigned XCH_BUF = KW * XCH_WAVE;                  // one buffer of slots; PIPE keeps two
    constexpr unsigned THR_AREA = XCH_AREA + (PIPE ? 2u : 1u) * XCH_BUF; // 16 NCB queries x 32 B (<= 2 KiB)
    constexpr unsigned THR_BYTES = TN * 32u;
    constexpr unsigned DUMP_AREA = THR_AREA + 2048u;             // 1 KiB: where the dummy refresh "lands"
    constexpr unsigned CNT_AREA = DUMP_AREA + 1024u;
    static_assert(CNT_AREA + 64 <= 160 * 1024, "LDS budget");

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

    const int kp = KW == 8 ? wave : (wave & 3);     // this wave's K part ...
    const int dh = KW == 8 ? 0 : (wave >> 2);       // ... and (KW = 4) its document half
    // ---- stationary query fragments of this wave's K slice: lane holds Q[q0 + 16 n + c][32 (KS kp + j) + 8 g .. +8)
    bf16x8 bq[NCB][KS];
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
        const uint16_t* qrow = p.qbuf + ((int64_t)qt * TN + n * 16 + c) * p.ld + 32 * KS * kp + 8 * g;
#pragma unroll
        for (int j = 0; j < KS; ++j) bq[n][j] = *reinterpret_cast<const bf16x8*>(qrow + 32 * j);
    }

    // the one sub-list of this lane: documents 4 g .. 4 g + 3 of half `th` against query 16 tn + c (waves >= TILES keep none)
    const int th = wave / NCB, tn = wave % NCB;
    const bool reducer = wave < TILES;
    float ls[KL];
    int li[KL];
    float thr = -INFINITY;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
        ls[i] = -INFINITY;
        li[i] = IDX_NONE;
    }

    // ---- shared insert bounds: p.gthr = [query][8 words]; the tile's TN queries are THR_BYTES contiguous bytes
    const __amdgpu_buffer_rsrc_t thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * THR_BYTES), 0, (int)THR_BYTES, 0x00020000);
    for (int t = tid; t < (int)(THR_BYTES / 16u); t += 512) *reinterpret_cast<uint4*>(smem + THR_AREA + t * 16) = make_uint4(0u, 0u, 0u, 0u);
    // every wave issues ONE such operation per block (uniform vmcnt arithmetic): when `real`, wave w < THR_PIECES brings KiB w of
    // the tile's class words; lanes past the words, and every lane of the other waves, point out of range (no memory access,
    // zeros into the dump area).  Readers never wait for these: a word that has not arrived yet is merely an older, weaker bound.
    constexpr int THR_PIECES = (int)((THR_BYTES + 1023u) / 1024u);
    static_assert(THR_PIECES <= 2, "");
    auto refresh_thresholds = [&](bool real) {
        const unsigned ln = lane_id_here();
        const bool mine = real && wave < THR_PIECES;
        const unsigned at = (unsigned)wave * 1024u + ln * 16u;
        lds_void* dst = (lds_void*)(smem + (mine ? THR_AREA + (unsigned)wave * 1024u : DUMP_AREA));
        const unsigned voff = (mine && at < THR_BYTES) ? at : (0x40000000u | (ln * 16u));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, dst, 16, voff, 0, 0, 16);
    };

    // ---- LDS-DMA map (scan_kernel_v4's with 128-byte slabs): piece pc = slab * 4 + rg = rows 8 rg .. + 7 of slab `slab`
    const unsigned char* docs_b = pa.docs;
    constexpr int row_bytes = LDB;
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, V3_DB * row_bytes, 0x00020000);
        const int pc = wave + WAVES * i;
        const int slab = pc >> 2, rg = pc & 3;
        const unsigned ln = lane_id_here();
        const unsigned lane_off0 = (ln >> 3) * (unsigned)row_bytes + (((ln & 7u) ^ ((ln >> 4) & 7u)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + stage * STAGE_BYTES + pc * 1024), 16,
                                                 (rg & 1) ? (lane_off0 ^ 64u) : lane_off0, rg * 8 * row_bytes + slab * 128, 0, NT_DOCS ? 2 : 0);
    };
    constexpr int PER_BLOCK = PPW + 1; // VMEM operations per wave and block: its pieces + one threshold operation

    // ---- counter barriers in LDS (inline asm: see scan_kernel_v3.hpp, "split barrier")
    const unsigned cnt_a = (unsigned)(size_t)(lds_void*)(smem + CNT_AREA), cnt_b = cnt_a + 16u;
    auto bump = [&](unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane_id_here() == 0u) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
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

    const unsigned char* first = docs_b + (int64_t)b0 * V3_DB * row_bytes;
    const unsigned char* last = docs_b + (int64_t)(b1 - 1) * V3_DB * row_bytes;
    constexpr int64_t blk_bytes = (int64_t)V3_DB * row_bytes;
    constexpr int AHEAD = STAGES - 1;
    if (tid < 16) reinterpret_cast<unsigned*>(smem + CNT_AREA)[tid] = 0u;
    if (nb > 0) {
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) { // same operation sequence as steady-state blocks (vmcnt arithmetic)
            refresh_thresholds(true);
#pragma unroll
            for (int i = 0; i < PPW; ++i) issue_piece(a < nb ? first + a * blk_bytes : last, a, i);
        }
    }
    __syncthreads();
    const unsigned char* pbase = nb > AHEAD ? first + AHEAD * blk_bytes : last;
    int stage = 0, pstage = AHEAD;

    // this wave's K slice of the block in ring stage `stg`: convert + multiply, partial sums -> its slot of buffer `buf`
    auto multiply = [&](int stg, unsigned buf) {
        const unsigned char* sa = smem + stg * STAGE_BYTES;
        f32x4 acc[HPW][NCB];
#pragma unroll
        for (int half = 0; half < HPW; ++half)
#pragma unroll
            for (int n = 0; n < NCB; ++n) acc[half][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned ln = lane_id_here();
        const unsigned cc = ln & 15u, gg = ln >> 4;
        const unsigned rowoff = cc * 128u + 8u * (gg & 1u) + (unsigned)dh * 2048u; // (rows 16 dh + c: 16 rows x 128 B per slab half)
        const unsigned swz = (cc >> 1) & 7u;
        // all KS fragment reads of this wave are issued up front (2 HPW registers each) and waited for one k-step at a time: with a
        // read + lgkmcnt(0) per step the LDS round trip (~100 cycles) was exposed KS times per block
        // (inline asm: behind an ordinary load of an LDS-DMA destination hipcc first drains vmcnt(0) -- the whole ring)
        u32x2 raw[KS][2]; // (a native vector type: the "+v" ties of the waits below are not honoured for HIP's uint2 struct)
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int sg = KS * kp + j;                // global k32-step (wave-uniform)
            const unsigned chunk = 2u * (unsigned)(sg & 3) + (gg >> 1);
            const unsigned off = (unsigned)(sg >> 2) * 4096u + rowoff + ((chunk ^ swz) << 4);
            raw[j][0] = raw[j][1] = u32x2{0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
            const unsigned a0 = (unsigned)(size_t)(lds_void*)sa + off;
            if (HPW == 2) asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:2048" : "=&v"(raw[j][0]), "=&v"(raw[j][1]) : "v"(a0) : "memory");
            else asm volatile("ds_read_b64 %0, %1" : "=&v"(raw[j][0]) : "v"(a0) : "memory");
#endif
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
            // the reads of steps 0 .. j have returned once at most HPW (KS - 1 - j) are outstanding (LDS returns in order)
            if (HPW == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(raw[j][0]), "+v"(raw[j][1]) : "n"(2 * (KS - 1 - j)) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(raw[j][0]) : "n"(KS - 1 - j) : "memory");
#endif
#pragma unroll
            for (int half = 0; half < HPW; ++half) {
                const bf16x8 af = e4m3x8_to_bf16x8(raw[j][half][0], raw[j][half][1]);
#pragma unroll
                for (int n = 0; n < NCB; ++n) acc[half][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bq[n][j], acc[half][n], 0, 0, 0);
            }
        }
        // partial tile (document half h, query block n) of K part kp: slot [kp][h NCB + n]
        unsigned char* slot = smem + XCH_AREA + buf * XCH_BUF + (unsigned)kp * XCH_WAVE + ln * 16u;
#pragma unroll
        for (int half = 0; half < HPW; ++half)
#pragma unroll
            for (int n = 0; n < NCB; ++n) *reinterpret_cast<f32x4*>(slot + ((HPW == 2 ? half : dh) * NCB + n) * 1024) = acc[half][n];
    };
    // (reducing waves) tile `wave` of block `blk`: the KW slots of buffer `buf` summed in slot order, then the top-K epilogue
    auto reduce = [&](int blk, unsigned buf, bool refresh) {
        const unsigned ln = lane_id_here();
        const unsigned char* src = smem + XCH_AREA + buf * XCH_BUF + (unsigned)wave * 1024u + ln * 16u;
        f32x4 sum = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
        for (int w = 1; w < KW; ++w) sum = sum + *reinterpret_cast<const f32x4*>(src + w * XCH_WAVE);
        if (refresh) { // minimum of the 8 class words of query 16 tn + c (what an earlier refresh brought, or 0)
            const unsigned a0 = (unsigned)(size_t)(lds_void*)smem + THR_AREA + ((unsigned)tn * 16u + (ln & 15u)) * 32u;
            u32x4 w0 = {0u, 0u, 0u, 0u}, w1 = w0;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0), "=&v"(w1) : "v"(a0) : "memory");
#endif
            const unsigned key = min(min(min(w0[0], w0[1]), min(w0[2], w0[3])), min(min(w1[0], w1[1]), min(w1[2], w1[3])));
            thr = fmaxf(thr, key > 1u ? thr_decode(key - 1u) : -INFINITY);
        }
        const int base = blk * V3_DB + 16 * th + 4 * (int)(ln >> 4);
        if ((int64_t)(blk + 1) * V3_DB > p.ntotal) { // ragged last block of the index (uniform)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if ((int64_t)(base + r) >= p.ntotal) sum[r] = -INFINITY;
        }
        const float mx = fmaxf(fmaxf(sum[0], sum[1]), fmaxf(sum[2], sum[3]));
        if (__ballot(mx > thr) != 0ull) {
            const float mark = ls[PUB - 1];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sc = sum[r];
                if (sc > thr) {
                    list_insert<KL>(ls, li, sc, base + r);
                    thr = fmaxf(thr, ls[KL - 1]);
                }
            }
            if (ls[PUB - 1] > mark) { // new PUB-th best of this sub-list: raise its class word, (4 split + g) & 7
                const unsigned cls = (4u * (unsigned)split + (ln >> 4)) & 7u;
                publish_umax(thr_encode(ls[PUB - 1]), ((unsigned)tn * 16u + (ln & 15u)) * 32u + 4u * cls, thr_rsrc);
            }
        }
    };
    auto refresh_of = [](int i) { return i < 8 || (i & 7) == 0; };

    if (PIPE) {
        // one barrier per block; iteration i multiplies block i and reduces block i - 1 (one more iteration for the last block)
        for (int i = 0; i <= nb && nb > 0; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PER_BLOCK) : "memory"); // this wave's share of block i has landed
            bump(cnt_a);
            poll(cnt_a, (unsigned)(i + 1) * WAVES);
            refresh_thresholds(refresh_of(i)); // (issued in the extra iteration too: uniform vmcnt arithmetic)
#pragma unroll
            for (int t = 0; t < PPW; ++t) issue_piece(pbase, pstage, t); // block i + AHEAD into the stage block i - 1 left
            if (i < nb) multiply(stage, (unsigned)(i & 1));
            if (i > 0 && reducer) reduce(b0 + i - 1, (unsigned)((i - 1) & 1), refresh_of(i - 1));
            if (i + AHEAD + 1 < nb) pbase += blk_bytes;
            stage = stage == STAGES - 1 ? 0 : stage + 1;
            pstage = pstage == STAGES - 1 ? 0 : pstage + 1;
        }
    } else {
        for (int i = 0; i < nb; ++i) {
            // barrier A: this wave's share of block i has landed (everything but the youngest AHEAD - 1 blocks' operations) ...
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AHEAD - 1) * PER_BLOCK) : "memory");
            bump(cnt_a);
            poll(cnt_a, (unsigned)(i + 1) * WAVES); // ... and everybody's; everybody is also done with block i - 1 (its stage, the slots)
            refresh_thresholds(refresh_of(i));
#pragma unroll
            for (int t = 0; t < PPW; ++t) issue_piece(pbase, pstage, t); // block i + AHEAD into the stage block i - 1 just left
            multiply(stage, 0u);
            bump(cnt_b); // (a wave's LDS operations complete in order: the counter is visible after the slot)
            if (reducer) {
                poll(cnt_b, (unsigned)(i + 1) * WAVES);
                reduce(b0 + i, 0u, refresh_of(i));
            }
            if (i + AHEAD + 1 < nb) pbase += blk_bytes;
            stage = stage == STAGES - 1 ? 0 : stage + 1;
            pstage = pstage == STAGES - 1 ? 0 : pstage + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no DMA may outlive the workgroup's LDS allocation

    // lists: [q][nsplit][8 = 2 document halves x 4 lane groups][KL]
    if (reducer) {
        const unsigned ln = lane_id_here();
        const int q = qt * TN + tn * 16 + (int)(ln & 15u);
        const size_t o = (((size_t)q * p.nsplit + split) * 8 + th * 4 + (int)(ln >> 4)) * KL;
#pragma unroll
        for (int i = 0; i < KL; ++i) {
            p.part_s[o + i] = ls[i];
            p.part_i[o + i] = li[i];
        }
    }
}

} // namespace mips
