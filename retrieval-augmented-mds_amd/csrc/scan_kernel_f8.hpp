// scan_kernel_f8: the query-stationary fused score + top-K scan over an fp8 (OCP e4m3) index
// (BASELINE config 5).  Same structure, contract and epilogue as scan_kernel_v3 (see that file); what
// differs is the arithmetic and the byte counts:
//   * documents AND queries are e4m3; products run on the CDNA4 K = 64 matrix instruction
//     v_mfma_f32_32x32x64_f8f6f4 (fp32 accumulate, unit scales): 2x the bf16 MFMA rate per clock;
//   * a wave's stationary B fragments are 8 VGPRs per 64-k step -> d/8 VGPRs for its 32 queries
//     (96 at d = 768), 8 waves = 256 queries per workgroup;
//   * a 32-document block is 32 x d bytes (24 KiB at d = 768): half the HBM / L2 / LDS-DMA traffic of the
//     bf16 index for the same documents, and the same 16 B/clk/CU fill need at twice the MFMA speed.
// Operand mapping: lane (row|col = lane & 31, half = lane >> 5) supplies the 32 consecutive k-bytes
// [64 s + 32 half, +32) of its row for k-step s, for A (documents, from LDS) and B (queries, registers)
// alike -- the instruction pairs A and B bytes position by position, so any k assignment used on BOTH
// sides yields the full dot product.
// LDS image: 128-byte slabs of [32 rows][128 B], 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7),
// filled by 8-row x 128-B LDS-DMA pieces whose per-lane SOURCE address carries the swizzle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"
#include "scan_kernel_v3.hpp"

namespace mips {

typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));

struct ScanArgsF8 {
    const uint8_t* docs; // [capacity][ld] e4m3 bytes
    const uint8_t* qbuf; // [nq_pad][ld]  e4m3 bytes
    ScanArgs c;          // common fields (docs/qbuf in there are unused)
};

// NT_DOCS: non-temporal document DMA for searches of ONE query tile (every block has a single reader), as in scan_kernel_v3 / v4
template <int KL, int LD, int AD, bool NT_DOCS = false>
__global__ __launch_bounds__(512, 2) void scan_kernel_f8(ScanArgsF8 pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ScanArgs& p = pa.c;
    constexpr int WAVES = 8;
    constexpr int TN = WAVES * 32;
    constexpr int STAGES = 3;
    constexpr int KS = LD / 64;             // MFMA k-steps per block
    constexpr int STAGE_BYTES = V3_DB * LD; // 32 rows x LD bytes
    constexpr int PIECES = STAGE_BYTES / 1024;
    constexpr int PPW = PIECES / WAVES;
    static_assert(LD % 256 == 0, "row length must be a multiple of 256 bytes (equal DMA piece counts per wave)");

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
    if (qt * TN >= nq_run) return;
    if (p.spin_limit < 0 && tid == 0) *p.err = 1u; // test-only: force the scan-error path (include/mips_hip.h, "spin_limit")
    const bool idle_wave = (qt * TN + wave * 32) >= nq_run;

    const int b0 = split * p.tiles_per_split;
    int b1 = b0 + p.tiles_per_split;
    if (b1 > p.ntiles) b1 = p.ntiles;
    const int nb = b1 > b0 ? b1 - b0 : 0;

    // ---- stationary query fragments: 32 bytes per 64-k step
    v8i32 bq[KS];
    {
        const uint8_t* qrow = pa.qbuf + ((int64_t)qt * TN + wave * 32 + l31) * LD + 32 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i32 lo = *reinterpret_cast<const v4i32*>(qrow + 64 * s);
            const v4i32 hi = *reinterpret_cast<const v4i32*>(qrow + 64 * s + 16);
            bq[s] = v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(bq[s]));
#endif
    }

    float ls[KL];
    int li[KL];
    float thr = -INFINITY;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
        ls[i] = -INFINITY;
        li[i] = IDX_NONE;
    }

    // shared per-query thresholds: class maxima, 8 words per query, sparse re-read (scan_kernel_v3.hpp, TMODE 2)
    constexpr int PUB = (KL + 7) / 8; // a list publishes its PUB-th best: 8 classes x PUB >= K' documents
    static_assert(8 * PUB >= KL, "the class words must prove at least K' documents");
    constexpr unsigned THR_AREA = STAGES * STAGE_BYTES;
    constexpr unsigned THR_WAVE = 1024u;
    constexpr unsigned DUMP_AREA = THR_AREA + WAVES * THR_WAVE;
    static_assert(THR_AREA % 1024 == 0, "the wave areas are recovered from thr_addr by masking");
    const unsigned thr_addr = THR_AREA + wave * THR_WAVE + lane * 16;
    *reinterpret_cast<uint4*>(smem + thr_addr) = make_uint4(0u, 0u, 0u, 0u);
    const __amdgpu_buffer_rsrc_t thr_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<unsigned char*>(p.gthr) + (int64_t)qt * (WAVES * THR_WAVE) - (int64_t)THR_AREA), 0,
        (int)(THR_AREA + WAVES * THR_WAVE), 0x00020000);
    auto refresh_thresholds = [&](bool real) { // !real: out-of-range dummy into the dump area (keeps vmcnt uniform)
        lds_void* dst = (lds_void*)(smem + (real ? THR_AREA + wave * THR_WAVE : DUMP_AREA));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(thr_rsrc, dst, 16, real ? thr_addr : (thr_addr | 0x40000000u), 0, 0, 16);
    };

    // ---- LDS-DMA map: piece pc = slab * 4 + rg: rows 8 rg .. 8 rg + 7 of the 128-byte slab `slab`
    const int lrow = lane >> 3;
    const unsigned lane_off0 = (unsigned)(lrow * LD + (((lane & 7) ^ ((lane >> 4) & 7)) << 4));
    constexpr int64_t blk_bytes = (int64_t)V3_DB * LD;
    auto issue_piece = [&](const unsigned char* blk_base, int stage, int i) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)blk_base, 0, (int)blk_bytes, 0x00020000);
        const int pc = wave + WAVES * i;
        const int slab = pc >> 2, rg = pc & 3;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + stage * STAGE_BYTES + pc * 1024), 16,
                                                 (rg & 1) ? (lane_off0 ^ 64u) : lane_off0, rg * 8 * LD + slab * 128, 0, NT_DOCS ? 2 : 0);
    };
    auto issue = [&](const unsigned char* blk_base, int stage) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(blk_base, stage, i);
    };

    const int rd_row = l31 * 128;
    const int rd_swz = (l31 >> 1) & 7;

    // split barrier on an LDS arrival counter (see scan_kernel_v3.hpp)
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
        const unsigned char* sa = smem + stage * STAGE_BYTES + rd_row;
        auto lds_frag = [&](int s) {
            // k-step s = 64 bytes = chunks 4 (s & 1) .. + 3 of slab s >> 1; this lane half takes two of them
            const int c0 = 4 * (s & 1) + 2 * h;
            const v4i32 lo = *reinterpret_cast<const v4i32*>(sa + (s >> 1) * 4096 + ((c0 ^ rd_swz) << 4));
            const v4i32 hi = *reinterpret_cast<const v4i32*>(sa + (s >> 1) * 4096 + (((c0 + 1) ^ rd_swz) << 4));
            return v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        v8i32 ar[AD];
#pragma unroll
        for (int s = 0; s < AD; ++s) ar[s] = lds_frag(s);
        refresh_thresholds(refresh); // first VMEM op of the block
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ar[s % AD], bq[s], acc, 0, 0, 0, 0, 0, 0);
            if (s + AD < KS) ar[s % AD] = lds_frag(s + AD);
            if ((s % (KS / PPW)) == (KS / PPW) / 2) issue_piece(pbase, pstage, s / (KS / PPW));
            __builtin_amdgcn_sched_barrier(0);
        }
        arrive(); // chain done, share of the next block landed; the epilogue below runs un-synchronised
        const int base = blk * V3_DB + (int)((thr_addr >> 7) & 4u);
        if ((int64_t)(blk + 1) * V3_DB > p.ntotal) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((int64_t)(base + (r & 3) + 8 * (r >> 2)) >= p.ntotal) acc[r] = -INFINITY;
        }
        if (refresh) { // minimum of the query's 8 class words (inline asm: see scan_kernel_v3.hpp)
            const unsigned qwords = (unsigned)(size_t)(lds_void*)smem + (thr_addr & ~0x3FFu) + ((thr_addr & 0x1F0u) << 1);
            u32x4 c0 = {0u, 0u, 0u, 0u}, c1 = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(c0), "=&v"(c1)
                         : "v"(qwords)
                         : "memory");
#endif
            const unsigned key = min(min(min(c0[0], c0[1]), min(c0[2], c0[3])), min(min(c1[0], c1[1]), min(c1[2], c1[3])));
            thr = fmaxf(thr, key > 1u ? thr_decode(key - 1u) : -INFINITY);
        }
        float mx = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
        if (__ballot(mx > thr) != 0ull) {
            const float mark = ls[PUB - 1];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = acc[r];
                if (s > thr) {
                    list_insert<KL>(ls, li, s, base + (r & 3) + 8 * (r >> 2));
                    thr = fmaxf(thr, ls[KL - 1]);
                }
            }
            if (ls[PUB - 1] > mark) { // this list vouches for more: raise its class word
                const unsigned cls = (2u * (unsigned)split + ((thr_addr >> 9) & 1u)) & 7u;
                publish_umax(thr_encode(ls[PUB - 1]), (thr_addr & ~0x3FFu) + ((thr_addr & 0x1F0u) << 1) + 4u * cls, thr_rsrc);
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

    const int q = qt * TN + wave * 32 + l31;
    const size_t o = (((size_t)q * p.nsplit + split) * 2 + h) * KL;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
        p.part_s[o + i] = ls[i];
        p.part_i[o + i] = li[i];
    }
}

} // namespace mips
