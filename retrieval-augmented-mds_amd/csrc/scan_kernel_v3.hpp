// scan_kernel_v3: query-stationary fused score + top-K scan on v_mfma_f32_32x32x16_bf16.
// The structure every scan kernel of this library shares (scan_kernel_v4 = the same scan on the 16x16x32 shape, the
// default for multi-tile searches with k <= 5 at row pitches 384 .. 768; scan_kernel_f8 / f8x = fp8).  This instance
// serves single-tile searches (HBM-bound, non-temporal document DMA), k = 6 .. 29, row pitches 128 / 256 and 1024.
//
// This problem is a GEMM with a tiny K (d = 768) and an enormous M (the index): instead of tiling it
// like a square GEMM, every wave keeps the MFMA B-fragments of ITS 32 queries for the WHOLE K in
// registers (KS16 = d/16 fragments x 4 VGPRs = 192 VGPRs at d = 768) for the lifetime of the
// workgroup.  8 waves = 256 stationary queries per workgroup.  Only documents move:
//   HBM/L2 -> LDS  by global_load_lds_dwordx4 (LDS-DMA), 32-document blocks (32 x d x 2 B = 48 KiB) in
//                  a 3-deep ring, two blocks ahead, completion by counted s_waitcnt vmcnt + an LDS arrival counter;
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

constexpr int V3_DB = 32; // documents per block
// Configurations (queries per workgroup = WAVES * NQB * 32):
//   d <= 768 : WAVES = 8, NQB = 1, STAGES = 3  -- two waves per SIMD, 256 registers each (192 fragment VGPRs at d = 768)
//   d = 1024 : WAVES = 4, NQB = 1, STAGES = 2  -- one wave per SIMD with the 512-register file (256 fragment VGPRs),
//                                                 64 KiB blocks, so only two ring stages fit the 160 KiB LDS
//   (WAVES = 4, NQB = 2 is the 64-queries-per-wave experiment of profiles/r1_v3_query_stationary.)
// inline asm is device-only: the host pass parses kernel bodies too and must not see GPU constraints
__device__ __forceinline__ void keep_alive(const f32x16& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#else
    (void)v;
#endif
}

__device__ __forceinline__ void keep_alive_f(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"v"(v));
#else
    (void)v;
#endif
}

// no-return buffer_atomic_umax through the (shifted) threshold descriptor: the slot offset is the VGPR the
// lane already holds, so publishing needs no 64-bit address registers (the kernel sits at the 256-VGPR
// cap).  No VGPR destination => nothing for hipcc's waitcnt bookkeeping to miss; it only makes the
// counted vmcnt waits more conservative.  s_nop 4: SGPR-written-by-SALU -> VMEM descriptor read hazard.
__device__ __forceinline__ void publish_umax(unsigned key, unsigned voff, __amdgpu_buffer_rsrc_t rsrc) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 4\n\tbuffer_atomic_umax %0, %1, %2, 0 offen" ::"v"(key), "v"(voff), "s"(rsrc) : "memory");
#else
    (void)key; (void)voff; (void)rsrc;
#endif
}

typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void gbl_void;

// GLOBAL_THR: how the lists of one query (in different workgroups) share an insert bound --
//   0 none, 1 every list publishes its K'-th best (one word per lane), 2 every list publishes its BEST into
//   one of 8 class words of its query and the bound is the minimum of the 8 (see "shared thresholds" below).
template <int KL, int KS16, int NQB, int AD, bool DMA_SPREAD, int TIMING_MODE = 0, int GLOBAL_THR = 2,
          int WAVES = 8 / NQB, int STAGES = 3, bool SPLIT_BAR = true, bool NT_DOCS = false, int THR_PERIOD = 8>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void scan_kernel_v3(ScanArgs p) {
    constexpr int V3_TN = WAVES * NQB * 32;
    constexpr int TMODE = NQB == 1 ? GLOBAL_THR : 0;
    // TMODE 2 publishes a list's PUB-th best: 8 classes x PUB documents prove 8 PUB >= K' documents above the bound
    constexpr int PUB = (KL + 7) / 8;
    static_assert(8 * PUB >= KL && PUB <= KL, "the class words must prove at least K' documents");
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
    const int nq_run = p.nq_dev != nullptr ? *p.nq_dev : p.nq; // (stream-ordered re-scan: the count is on the device)
    if (qt * V3_TN >= nq_run) return;
    if (p.spin_limit < 0 && tid == 0) *p.err = 1u; // test-only: force the scan-error path (include/mips_hip.h, "spin_limit")

    const int b0 = split * p.tiles_per_split; // "tiles" are 32-document blocks here
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // A wave whose queries are ALL padding (small searches: 8 real queries leave 7 of the 8 waves of the one query tile
    // without work) still brings its share of every document block and takes part in the block barrier, but skips
    // the MFMA chain and the epilogue: fewer issue slots and less power spent next to the document stream, which is all
    // that matters in this HBM-bound regime.
    const bool idle_wave = (qt * V3_TN + wave * NQB * 32) >= nq_run;

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
        // (One wave per SIMD: the fragments exceed or crowd the 256 architectural VGPRs; part of them is pinned
        // in the accumulation half of the unified file -- MFMA reads B operands from AGPRs directly --
        // otherwise hipcc treats AGPRs as spill space and copies 4 registers back before every MFMA.)
#pragma unroll
        for (int s = 0; s < KS16; ++s) {
            if (WAVES == 4 && (n * KS16 + s) < (NQB == 2 ? 64 : KS16 / 2)) asm volatile("" : "+a"(bq[n][s]));
            else asm volatile("" : "+v"(bq[n][s]));
        }
    }

    float ls[NQB][KL];
    int li[NQB][KL];
    // Insert threshold of a list: max(own K'-th best, bound from the shared per-query thresholds below).
    // (Exchanging bounds between lanes l and l + 32 -- same query, other rows -- was measured: the max of
    // the two K'-th bests is neutral, the exact K'-th of the pair union costs more than it saves.)
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

    // ---- shared per-query thresholds.  The lists of one query live in different workgroups (16+ splits x 2
    // lane halves); left alone, each warms up on its own: ~K' ln(n / K') inserts per LIST.  They share a
    // bound through p.gthr instead (order-preserving keys, atomic umax, re-read once per block by LDS-DMA).
    // Any value g such that at least K' distinct documents of the query score >= g is a valid bound -- a
    // document below g cannot be among the query's K' best -- and the insert test becomes
    //     s > max(own K'-th, nextbelow(g))      i.e.  s > own  and  s >= g
    // (ties with g are kept, the strict rule applies to the own list only; a stale g is merely weaker).
    //   TMODE 1: g = some list's K'-th best.  One word per lane ([query tile][wave][lane]; the same lane of
    //            other splits shares it), 256 B per wave.
    //   TMODE 2: every list belongs to one of 8 classes, (2 split + lane half) & 7, and publishes its BEST
    //            score (K' = 8; its K'/8-th best for longer lists) into the class word of its query; lists
    //            hold disjoint documents, so the 8 class words vouch for 8 (K') distinct documents and g =
    //            their minimum.  A list's best over the
    //            documents of ALL lists of a class converges like the true top of the scan rather than like
    //            one list's K'-th: the bound is useful after one insert per class instead of K' per list
    //            (measured: 2^20 rows, Q = 256 -> 256 splits of 128 blocks: 0.575 ms with TMODE 1, of which
    //            0.22 ms was this warm-up).  [query tile][wave][query][8 words], 1 KiB per wave = one
    //            buffer_load_dwordx4 ... lds per block.
    // thr_addr is this lane's chunk of the wave's LDS threshold area and, with the descriptor base shifted
    // by the area offset, also the DMA's voffset -- one persistent VGPR for both.
    constexpr unsigned THR_AREA = STAGES * STAGE_BYTES;
    constexpr unsigned THR_WAVE = TMODE == 2 ? 1024u : 256u;
    static_assert(THR_AREA % 1024 == 0, "the wave areas are recovered from thr_addr by masking");
    const unsigned thr_addr = THR_AREA + wave * THR_WAVE + lane * (TMODE == 2 ? 16 : 4);
    __amdgpu_buffer_rsrc_t thr_rsrc;
    if (TMODE) {
        if (TMODE == 2) *reinterpret_cast<uint4*>(smem + thr_addr) = make_uint4(0u, 0u, 0u, 0u);
        else *reinterpret_cast<unsigned*>(smem + thr_addr) = 0u;
        thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * (WAVES * THR_WAVE) - (int64_t)THR_AREA), 0,
            (int)(THR_AREA + WAVES * THR_WAVE), 0x00020000);
    }

    // ---- LDS-DMA map: piece pc = slab * 4 + rg covers rows 8 rg .. 8 rg + 7 of 64-k slab `slab`;
    // lane -> row 8 rg + (lane >> 3), slot lane & 7, source chunk slot ^ ((row >> 1) & 7)
    // = slot ^ ((4 rg + (lane >> 4)) & 7): depends on rg only through rg & 1.
    const int lrow = lane >> 3;
    // odd row groups: slot ^ 4, i.e. byte offset ^ 64 (row pitch is a multiple of 128 B) -- derived per
    // piece instead of held in a second register
    const unsigned lane_off0 = (unsigned)(lrow * p.ld * 2 + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
    const unsigned char* docs_b = reinterpret_cast<const unsigned char*>(p.docs);
    const int64_t row_bytes = (int64_t)p.ld * 2;

    // one DMA piece (i-th of this wave) of document block `blk` into ring stage `stage`:
    // buffer_load_dwordx4 ... offen lds -- descriptor (rebased per block, 48 KiB window, so the hardware
    // range check also fences the last block) and piece offset in SGPRs, the lane part in ONE VGPR.
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)(V3_DB * row_bytes), 0x00020000);
        unsigned char* sbase = smem + stage * STAGE_BYTES;
        const int pc = wave + WAVES * i;
        const int slab = pc >> 2, rg = pc & 3;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(sbase + pc * 1024), 16,
                                                 (rg & 1) ? (lane_off0 ^ 64u) : lane_off0,
                                                 rg * 8 * (int)row_bytes + slab * 128, 0, NT_DOCS ? 2 : 0);
    };
    auto issue = [&](const unsigned char* blk_base, int stage) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(blk_base, stage, i);
    };

    const int rd_row = l31 * 128;
    const int rd_swz = (l31 >> 1) & 7;

    // One LDS-DMA (sc1: served by L2, not by this CU's L1) re-reads the wave's threshold words.  `real` =
    // false issues the same instruction with an out-of-range offset into a dump area: no memory access (the
    // range check answers with zeros), but still one operation in the wave's vmcnt sequence, which keeps the
    // counted waits uniform while the expensive re-read happens only every few blocks (schedule: main loop).
    constexpr unsigned DUMP_AREA = THR_AREA + WAVES * THR_WAVE; // 1 KiB shared by all waves, never read
    auto refresh_thresholds = [&](bool real) {
        lds_void* dst = (lds_void*)(smem + (real ? THR_AREA + wave * THR_WAVE : DUMP_AREA));
        const unsigned voff = real ? thr_addr : (thr_addr | 0x40000000u);
        if (TMODE == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, dst, 16, voff, 0, 0, 16);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, dst, 4, voff, 0, 0, 16);
    };

    // ---- split barrier (SPLIT_BAR): gfx950's s_barrier is arrive-and-wait in one instruction, so one wave
    // taking the divergent insert path at the end of a block holds up all eight at the next barrier.  Here a
    // wave ARRIVES right after its MFMA chain (one ds_add on an LDS counter, after the counted vmcnt wait that
    // proves its DMA share of the NEXT block has landed), then runs its epilogue, and only WAITS -- polls
    // for all WAVES arrivals -- before its next chain.  All-arrived(m) means: every share of block m + 1 has
    // landed (RAW) and every wave has finished reading block m, whose stage the DMA of block m + 3 reuses
    // (WAR).  The spin is bounded; giving up sets p.err (results are then invalid, the kernel still ends).
    // The counter traffic is inline asm: for a volatile / atomic LDS access hipcc first drains vmcnt(0) (it
    // cannot prove the word does not alias an in-flight LDS-DMA destination), which would stall the ring.
    unsigned* arrive_cnt = reinterpret_cast<unsigned*>(smem + DUMP_AREA + 1024);
    const unsigned cnt_lds = (unsigned)(size_t)(lds_void*)arrive_cnt; // 32-bit LDS byte address
    unsigned arrivals_needed = 0;
    // VMEM operations of one block, in issue order: the threshold refresh (real or dummy, see below), then its
    // PPW DMA pieces (of block i + STAGES - 1): the same count every block, so that after chain i the operations
    // younger than this wave's share of block i + 1 are exactly (STAGES - 2) blocks' worth.
    constexpr int PER_BLOCK_OPS = PPW + (TMODE ? 1 : 0);
    auto arrive = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_BLOCK_OPS) : "memory");
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

    // scores block `blk` (ring stage `stage`); when pblk >= 0 the DMA pieces of block pblk are issued
    // one at a time between the MFMAs (spread over the chain: the partner wave keeps the matrix pipe busy
    // during an issue, which right after the barrier it could not, both waves being there together)
    // `refresh`: does this block re-read the shared threshold words (and apply them in its epilogue)?
    auto block = [&](bool refresh, int blk, int stage, const unsigned char* pbase, int pstage) {
        if (idle_wave) { // (uniform) the same VMEM operation sequence as a working wave: refresh, then the pieces
            if (TMODE) refresh_thresholds(false);
            if (DMA_SPREAD) issue(pbase, pstage);
            if (SPLIT_BAR) arrive();
            return;
        }
        // Fragment address = stage + slab (s >> 2) * 4096 + row * 128 + ((2 (s & 3) + h) ^ swz) * 16.  2 j + h = 2 j ^ h
        // (h is bit 0), so the lane part is a0 ^ (j << 5) with ONE per-lane value a0 = row * 128 + (h ^ swz) * 16 -- the
        // four variants are derived where they are used (an opaque v_xor: hipcc would otherwise keep all four in
        // registers for the whole block, which together with the 192 fragment registers cost the 8-wave configuration a
        // spilled B fragment, reloaded at the top of every block behind a vmcnt(0) that drained the LDS-DMA ring).
        const unsigned char* sa = smem + stage * STAGE_BYTES;
        const unsigned a0 = (unsigned)(rd_row + ((h ^ rd_swz) << 4));
        auto lds_frag = [&](int s) {
            unsigned av = a0;
#if defined(__HIP_DEVICE_COMPILE__)
            if (WAVES == 8 && (s & 3) != 0) asm volatile("v_xor_b32 %0, %2, %1" : "=v"(av) : "v"(a0), "n"((s & 3) << 5)); // (a literal must be src0 of the VOP2 form)
            else
#endif
                av = a0 ^ (unsigned)((s & 3) << 5);
            return *reinterpret_cast<const bf16x8*>(sa + (s >> 2) * 4096 + av);
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
        if (TMODE) refresh_thresholds(refresh); // first VMEM op of the block
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS16; ++s) {
#pragma unroll
            for (int n = 0; n < NQB; ++n)
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[s % AD], bq[n][s], acc[n], 0, 0, 0);
            if (s + AD < KS16) ar[s % AD] = lds_frag(s + AD);
            if (DMA_SPREAD && (s % (KS16 / PPW)) == (KS16 / PPW) / 2) issue_piece(pbase, pstage, s / (KS16 / PPW));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (SPLIT_BAR) arrive(); // chain done + this wave's share of the next block landed; the epilogue runs un-synchronised
        if (TIMING_MODE == 1) { // diagnostic builds only (results are wrong): 1 = no epilogue at all
#pragma unroll
            for (int n = 0; n < NQB; ++n) keep_alive(acc[n]);
            return;
        }
        // + 4 * (lane >> 5), from the lane bits of thr_addr
        const int base = blk * V3_DB + (int)((thr_addr >> (TMODE == 2 ? 7 : 5)) & 4u);
        if ((int64_t)(blk + 1) * V3_DB > p.ntotal) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((int64_t)(base + (r & 3) + 8 * (r >> 2)) >= p.ntotal) {
#pragma unroll
                    for (int n = 0; n < NQB; ++n) acc[n][r] = -INFINITY;
                }
        }
        // the words an earlier block's DMA brought (or 0)
        if (TMODE == 1 && refresh) {
            const unsigned key = *reinterpret_cast<const unsigned*>(smem + thr_addr);
            thr[0] = fmaxf(thr[0], key > 1u ? thr_decode(key - 1u) : -INFINITY);
        } else if (TMODE == 2 && refresh) {
            // wave area + 32 B * (lane & 31).  Read in inline asm: for an ordinary load of a DMA destination
            // hipcc first drains vmcnt(0), i.e. the whole document ring, once per block.
            const unsigned qwords = (unsigned)(size_t)(lds_void*)smem + (thr_addr & ~0x3FFu) + ((thr_addr & 0x1F0u) << 1);
            u32x4 c0 = {0u, 0u, 0u, 0u}, c1 = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(c0), "=&v"(c1)
                         : "v"(qwords)
                         : "memory");
#endif
            const unsigned key = min(min(min(c0[0], c0[1]), min(c0[2], c0[3])), min(min(c1[0], c1[1]), min(c1[2], c1[3])));
            thr[0] = fmaxf(thr[0], key > 1u ? thr_decode(key - 1u) : -INFINITY);
        }
#pragma unroll
        for (int n = 0; n < NQB; ++n) {
            float mx = acc[n][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[n][r]);
            if (TIMING_MODE == 2) { // 2 = pre-test only, slow path never taken
                if (__ballot(mx > thr[n]) != 0ull) keep_alive_f(mx); // branch taken as often as the real one
                continue;
            }
            if (__ballot(mx > thr[n]) != 0ull) {
                const float mark = TMODE == 2 ? ls[n][PUB - 1] : thr[n]; // what a publication must beat
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = acc[n][r];
                    if (s > thr[n]) {
                        list_insert<KL>(ls[n], li[n], s, base + (r & 3) + 8 * (r >> 2));
                        thr[n] = fmaxf(thr[n], ls[n][KL - 1]);
                    }
                }
                if (TMODE == 1 && ls[n][KL - 1] > mark) // own K'-th now beats every bound seen: publish
                    publish_umax(thr_encode(ls[n][KL - 1]), thr_addr, thr_rsrc);
                if (TMODE == 2 && ls[n][PUB - 1] > mark) { // this list vouches for more: raise its class word
                    const unsigned cls = (2u * (unsigned)split + ((thr_addr >> 9) & 1u)) & 7u;
                    publish_umax(thr_encode(ls[n][PUB - 1]), (thr_addr & ~0x3FFu) + ((thr_addr & 0x1F0u) << 1) + 4u * cls, thr_rsrc);
                }
            }
        }
    };

    // Block i lives in ring stage i % STAGES and is issued STAGES - 1 blocks ahead.  The issue is
    // unconditional: past the end of the split the LAST block is fetched again into a stage nobody reads any
    // more, which keeps the MFMA chain free of branches and the vmcnt arithmetic uniform.
    const unsigned char* first = docs_b + (int64_t)b0 * V3_DB * row_bytes;
    const unsigned char* last = docs_b + (int64_t)(b1 - 1) * V3_DB * row_bytes;
    const int64_t blk_bytes = V3_DB * row_bytes;
    constexpr int AHEAD = STAGES - 1;
    if (nb > 0) {
        // The prologue issues exactly what AHEAD steady-state blocks would (threshold refresh, then the
        // pieces), so that "everything but the (STAGES - 2) * PER_BLOCK youngest operations has landed"
        // means the same at block 0 as later.  (Leaving the refresh out here made vmcnt one too lax for
        // the first block: a cold-start race caught by test_mips_facade_end_to_end.)
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            if (TMODE) refresh_thresholds(true);
            issue(a < nb ? first + a * blk_bytes : last, a);
        }
    }
    const unsigned char* pbase = nb > AHEAD ? first + AHEAD * blk_bytes : last;
    int stage = 0, pstage = AHEAD;
    // VMEM operations a wave issues per block: its DMA pieces and the threshold refresh; at the top of
    // block i everything older than the (STAGES - 2) youngest blocks' worth must have landed
    constexpr int PER_BLOCK = PPW + (TMODE ? 1 : 0);
    if (SPLIT_BAR) {
        if (tid == 0) *arrive_cnt = 0u;
        __syncthreads(); // the one real barrier: counter initialised (its vmcnt(0) also settles the prologue's loads)
        if (nb > 0) arrive();   // prologue arrival: this wave's share of block 0 has landed
    }
    for (int i = 0; i < nb; ++i) {
        if (SPLIT_BAR) {
            wait_all();
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * PER_BLOCK) : "memory");
            __builtin_amdgcn_s_barrier(); // all shares of block i landed; everyone is done with block i-1
        }
        if (!DMA_SPREAD) issue(pbase, pstage);
        // Threshold refresh schedule: every block while the lists warm up, every 8th afterwards.  All the
        // workgroups of a query tile re-read the same few cache lines, which the publications keep evicting
        // from L2 (atomics drop the line): refreshed every block by 256 workgroups those lines saturate, and
        // the refresh, completing in order with the document DMA, stalls the ring (2^20 rows, Q = 256: sharing
        // was SLOWER than no sharing, 0.477 vs 0.447 ms).
        block(THR_PERIOD == 1 || i < 8 || (i % THR_PERIOD) == 0, b0 + i, stage, pbase, pstage);
        if (i + AHEAD + 1 < nb) pbase += blk_bytes; // stops at the last block
        stage = stage == STAGES - 1 ? 0 : stage + 1;
        pstage = pstage == STAGES - 1 ? 0 : pstage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no DMA may outlive the workgroup's LDS allocation

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
