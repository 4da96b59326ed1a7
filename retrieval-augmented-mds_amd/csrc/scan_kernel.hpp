// Fused score + top-K scan: the hot kernel of the MIPS path.
//
// Replaces the arithmetic of faiss IndexFlat.search behind sotasum/mips.py:383-386 (and of the
// brute-force cross-check mips.py:557-558: scores = x @ y.T; argsort).  The [Q, N] score matrix
// never exists in memory: every 128-doc x 128-query tile of it lives in MFMA accumulators and is
// folded straight into per-lane running top-K lists.
//
// Orientation (gfx950 v_mfma_f32_32x32x16_bf16, C/D map col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)):
//     A operand = document rows (M index), B operand = query rows (N index)
// so the QUERY sits on the lane and the 16 accumulator registers of a lane are 16 different
// documents scored against that one query.  A lane therefore owns one query's running list: the
// threshold test is a per-lane register compare, no cross-lane traffic in the steady state.
// Lanes l and l + 32 hold disjoint document rows of the same query; their two lists, and the lists
// of the other index splits, are merged afterwards (merge_select_kernel + rescore_rank_kernel).
//
// Within one list documents arrive in strictly increasing index order, so "insert only if strictly
// greater than the current K-th" implements the tie rule "lowest index wins" without comparing
// indices.
//
// Work decomposition: grid = (#query tiles, padded to the group count) x (#index splits, a multiple of 8).  A workgroup keeps
// its query tile and walks the document tiles of its split; the flattened (tile, k-step) sequence
// is software pipelined (global loads of step s+1 in flight during the MFMAs of step s, LDS double
// buffered, one barrier per step).  blockIdx is decoded so that all query tiles of one split share
// an XCD (blocks b and b + 8 share one): the split's documents are fetched from HBM once and served
// to the other query tiles from that XCD's L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mips {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128;           // documents per tile
constexpr int TN = 128;           // queries per tile
constexpr int BK = 64;            // k elements per pipeline step (128 B of a bf16 row)
constexpr int SCAN_THREADS = 256; // 4 waves; wave w owns queries [32w, 32w+32) x all 128 docs
constexpr int IDX_NONE = 0x7fffffff;
constexpr int SCAN_LDS_BYTES = 2 * (TM + TN) * BK * 2; // double buffered A|B = 64 KiB

struct ScanArgs {
    const uint16_t* docs; // [capacity][ld] bf16 bits, capacity a multiple of TM
    const uint16_t* qbuf; // [nq_pad][ld]  bf16 bits, nq_pad a multiple of TN, pad rows/cols zero
    int64_t ntotal;       // valid documents
    int ld;               // row length in elements (d padded to a multiple of BK)
    int ksteps;           // ld / BK
    int ntiles;           // ceil(ntotal / TM)
    int tiles_per_split;
    int nsplit;           // multiple of 8
    int nqt;              // query tiles
    int nq;               // real queries (rows nq .. of the last tile are zero padding)
    int qgroups;          // QG in {1,2,4,8}: query-tile groups; XCD x serves group x % QG, split group x / QG
    int qt_per_group;     // ceil(nqt / QG)
    int splits_per_group; // nsplit / (8 / QG)
    float* part_s;        // [nq_pad][nsplit][2][KL]
    int* part_i;
    unsigned* gthr;       // [2 nq_pad] shared per-query thresholds (order-preserving keys, 0 = none), v3 only
    unsigned* err;        // one word, zeroed per launch: set when a bounded spin gave up (never expected).  The exact
                          // re-score reads it and POISONS the call's output (idx -2, NaN) -- aux_kernels.hpp
    int spin_limit;       // polls of the split barrier before a wave gives up (1 << 22; tests shrink it to force the flag)
    const int* nq_dev;    // stream-ordered re-scan ("margin_check" = 3): the query count lives on the device; nq / nqt
                          // above are then upper bounds the launch was sized for, and workgroups past the count leave
    int plane;            // > 0: fp32-exact mode (generic kernel only).  Rows are two bf16 planes [hi | lo] of
                          // `plane` elements each (x = hi + lo up to 2^-17 relative) and the k-loop runs three
                          // segments hi.qhi + hi.qlo + lo.qhi; ld = 2 * plane, ksteps = 3 * plane / BK.
};

// order-preserving map float -> uint32 (larger float <=> larger key); key 0 is below every float
__device__ __forceinline__ unsigned thr_encode(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float thr_decode(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// sorted (descending) K-list kept in registers; `s` is known to be > ls[KL-1]
template <int KL>
__device__ __forceinline__ void list_insert(float (&ls)[KL], int (&li)[KL], float s, int id) {
    ls[KL - 1] = s;
    li[KL - 1] = id;
#pragma unroll
    for (int j = KL - 1; j > 0; --j) {
        const float lo = ls[j], hi = ls[j - 1];
        const int ilo = li[j], ihi = li[j - 1];
        const bool sw = lo > hi; // strict: equal scores keep the earlier (lower index) one above
        ls[j - 1] = sw ? lo : hi;
        ls[j] = sw ? hi : lo;
        li[j - 1] = sw ? ilo : ihi;
        li[j] = sw ? ihi : ilo;
    }
}

template <int KL>
__global__ __launch_bounds__(SCAN_THREADS, 2) void scan_kernel(ScanArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h = lane >> 5;
    const int l31 = lane & 31;

    // XCD-aware decode: blocks with equal (blockIdx & 7) share an XCD and its 4 MiB L2.  Each XCD
    // is given a GROUP of query tiles (its query working set, qt_per_group x 192 KiB, must stay L2
    // resident next to the streaming document tiles) and a group of index splits; consecutive blocks
    // of an XCD are the query tiles of ONE split, so they walk the same document tiles together.
    const int xcd = blockIdx.x & 7;
    const int j = blockIdx.x >> 3;
    const int qt = (xcd % p.qgroups) + p.qgroups * (j % p.qt_per_group);
    const int split = (xcd / p.qgroups) * p.splits_per_group + j / p.qt_per_group;
    if (qt >= p.nqt) return; // padding block of the last query group (whole workgroup leaves)
    if (p.nq_dev != nullptr && qt * TN >= *p.nq_dev) return;

    const int t0 = split * p.tiles_per_split;
    int t1 = t0 + p.tiles_per_split;
    if (t1 > p.ntiles) t1 = p.ntiles;
    const int nt = t1 > t0 ? t1 - t0 : 0;

    float ls[KL];
    int li[KL];
#pragma unroll
    for (int i = 0; i < KL; ++i) {
        ls[i] = -INFINITY;
        li[i] = IDX_NONE;
    }

    // ---- staging map: thread -> (row srow + 32 i, 16-byte chunk schunk) of a 128 x 64 slab
    const int srow = tid >> 3;
    const int schunk = tid & 7;
    // LDS image: row r at r * 128 B, chunk c stored at chunk slot c ^ ((r >> 1) & 7): the 16 rows a
    // ds_read_b128 lane group touches then fall on 16 distinct 16-byte slots of the 256-byte bank row.
    const int st_off = srow * 128 + ((schunk ^ ((srow >> 1) & 7)) << 4);
    const int rd_swz = (l31 >> 1) & 7;

    const uint16_t* qbase = p.qbuf + (int64_t)qt * TN * p.ld + schunk * 8 + (int64_t)srow * p.ld;
    const uint16_t* dbase = p.docs + schunk * 8 + (int64_t)srow * p.ld;
    const int64_t row32 = (int64_t)32 * p.ld;

    u32x4 ra[4], rb[4];
    auto gload = [&](int tile, int ks) {
        int kd = ks * BK, kq = ks * BK;
        if (p.plane > 0) { // segment 0: hi.qhi, 1: hi.qlo, 2: lo.qhi
            const int kpp = p.plane / BK;
            const int seg = ks / kpp;
            const int kk = (ks - seg * kpp) * BK;
            kd = (seg == 2 ? p.plane : 0) + kk;
            kq = (seg == 1 ? p.plane : 0) + kk;
        }
        const uint16_t* a = dbase + (int64_t)tile * TM * p.ld + kd;
        const uint16_t* b = qbase + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const u32x4*>(a + i * row32);
            rb[i] = *reinterpret_cast<const u32x4*>(b + i * row32);
        }
    };
    auto swrite = [&](int buf) {
        unsigned char* sa = smem + buf * ((TM + TN) * BK * 2);
        unsigned char* sb = sa + TM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(sa + st_off + i * 32 * 128) = ra[i];
            *reinterpret_cast<u32x4*>(sb + st_off + i * 32 * 128) = rb[i];
        }
    };

    f32x16 acc[4];
    auto compute = [&](int buf) {
        const unsigned char* sa = smem + buf * ((TM + TN) * BK * 2);
        const unsigned char* sb = sa + TM * BK * 2;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int coff = ((2 * kk + h) ^ rd_swz) << 4;
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(sb + (wave * 32 + l31) * 128 + coff);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sa + (m * 32 + l31) * 128 + coff);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
            }
        }
    };

    auto epilogue = [&](int tile) {
        const int base = tile * TM + 4 * h;
        if ((int64_t)(tile + 1) * TM > p.ntotal) { // ragged last tile: rows past ntotal never rank
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((int64_t)(base + m * 32 + (r & 3) + 8 * (r >> 2)) >= p.ntotal) acc[m][r] = -INFINITY;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float mx = acc[m][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[m][r]);
            if (__ballot(mx > ls[KL - 1]) != 0ull) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { // ascending document index
                    const float s = acc[m][r];
                    if (s > ls[KL - 1]) list_insert<KL>(ls, li, s, base + m * 32 + (r & 3) + 8 * (r >> 2));
                }
            }
        }
    };

    const int total = nt * p.ksteps;
    if (total > 0) gload(t0, 0);
    int tile = t0, ks = 0;
    for (int step = 0; step < total; ++step) {
        const int buf = step & 1;
        swrite(buf);
        __syncthreads();
        int ntile = tile, nks = ks + 1;
        if (nks == p.ksteps) {
            nks = 0;
            ++ntile;
        }
        if (step + 1 < total) gload(ntile, nks);
        if (ks == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }
        compute(buf);
        if (ks == p.ksteps - 1) epilogue(tile);
        tile = ntile;
        ks = nks;
    }

    const int q = qt * TN + wave * 32 + l31;
    const size_t o = (((size_t)q * p.nsplit + split) * 2 + h) * KL;
#pragma unroll
    for (int i = 0; i < KL; ++i) {
        p.part_s[o + i] = ls[i];
        p.part_i[o + i] = li[i];
    }
}

} // namespace mips
