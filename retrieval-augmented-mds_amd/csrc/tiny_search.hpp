// tiny_search_kernel: the reference's OWN call shape in ONE launch.
//
// sotasum's training / generation step searches B <= 16 queries against a ~10^4-row knowledge base
// (retriever_generator.py:145-153 -> mips.py:402-463; BASELINE config 1: N = 10 000, d = 768, B = 8, k = 5).  Through
// the general path that is five dependent launches (query staging, fused scan, select, exact re-score, ignore filter;
// six with the query normalisation) of a few microseconds each: 0.11 ms device-resident, all of it launch latency.
// Here one kernel does it all (prepare + search + ignore mask fused):
//
//   every workgroup   stages the <= 16 queries itself: optional row normalisation exactly as mips_l2_normalize
//                     (faiss.normalize_L2, mips.py:369-370, 521-525), rounding to the index's bf16 (RNE), into LDS;
//   phase 1           each wave scores 16-document tiles straight from global memory on v_mfma_f32_16x16x32_bf16
//                     (A = 16 documents x 32 k per lane-load, B = the 16 staged queries), lane (c, g) keeps the running
//                     top-6 of documents 4 g .. 4 g + 3 of its tiles against query c (strict '>', ascending document
//                     order: lowest index wins ties), and writes its list; then the workgroup takes a ticket;
//   phase 2           the LAST workgroup to finish (agent-scope release / acquire around the ticket) selects the 8 best
//                     candidates per query, re-scores them exactly (sequential fp64, the canonical score), ranks,
//                     applies the k + 1 / ignore filter of mips.py:388-398 and writes the results.
// Same candidate-pool logic, same canonical scores and the same margin check as the general path: results are
// bit-identical to it (tests/test_gpu_parity.py::test_tiny_search_*).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hpp"
#include "scan_kernel.hpp"
#include "scan_kernel_v4.hpp"

namespace mips {

struct TinyArgs {
    const uint16_t* docs;  // [capacity][ld] bf16
    const void* q;         // [nq][d] caller's queries (device), float32 or bf16
    int q_is_f32;
    int normalize;         // row-normalise the (float32) queries first
    int nq;                // <= 16
    int d, ld;
    int64_t ntotal;
    int ntiles;            // ceil(ntotal / 16)
    int nwaves;            // gridDim.x * 4
                           // m.part_s / m.part_i: [16][gridDim.x][8] candidates of the workgroups, m.pre_bnd: [16][gridDim.x]
                           // their bounds, m.ncand = gridDim.x * 8, m.npre = gridDim.x, m.ll = INT_MAX (set by the host)
    unsigned* ticket;      // zeroed per launch
    const int64_t* ignore; // [nq] or nullptr: fetch k + 1, drop hits equal to ignore[q], keep k (mips.py:388-398)
    int k_out;             // results per query written to out_* (k); m.k = k or k + 1
    float* out_s;          // final outputs [nq][k_out] (device)
    int64_t* out_i;
    int64_t* out_packed;   // or the packed all-gather payload [nq][k_out][2]
    MergeArgs m;           // part_s / part_i / ncand / ll / docs / ld / k / metric / phi / idx_offset / margin fields;
                           // qbuf, out_s, out_i are set by the kernel (LDS)
};

constexpr int TINY_KL = 6;   // entries per lane list
constexpr int TINY_POOL = 8; // re-score pool
constexpr int TINY_MAXK = 6; // k (or k + 1) <= 6

template <bool L2>
__global__ __launch_bounds__(256, 1) void tiny_search_kernel(TinyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* qs = reinterpret_cast<uint16_t*>(smem);                            // [16][ld] staged queries
    float* wl_s = reinterpret_cast<float*>(smem + 16 * a.ld * 2);                // [16][96] this workgroup's lane lists
    int* wl_i = reinterpret_cast<int*>(wl_s + 16 * 96);                          // [16][96]
    int* cand = wl_i + 16 * 96;                                                  // [16][TINY_POOL]
    float* cand_s = reinterpret_cast<float*>(cand + 16 * TINY_POOL);             // [16][TINY_POOL]
    float* res_s = cand_s + 16 * TINY_POOL;                                      // [16][8]
    int64_t* res_i = reinterpret_cast<int64_t*>(res_s + 16 * 8);                 // [16][8] (8-byte aligned)
    __shared__ int is_last;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int nwg = gridDim.x;

    // ---- stage the queries: wave w takes rows w, w + 4, ...; one wave per row as mips_l2_normalize does it
    for (int r = wave; r < 16; r += 4) {
        uint16_t* dst = qs + r * a.ld;
        if (r >= a.nq) {
            for (int t = lane; t < a.ld; t += 64) dst[t] = 0;
            continue;
        }
        float inv = 1.0f;
        bool scale = false;
        if (a.normalize) { // (float32 source only: checked on the host)
            const float* src = reinterpret_cast<const float*>(a.q) + (size_t)r * a.d;
            float nr = 0.f;
            for (int t = lane; t < a.d; t += 64) nr += src[t] * src[t];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nr += __shfl_xor(nr, off);
            if (nr > 0.f) { // rows of norm 0 stay as they are (faiss fvec_renorm_L2)
                inv = 1.0f / sqrtf(nr);
                scale = true;
            }
        }
        for (int t = lane; t < a.ld; t += 64) {
            uint16_t v = 0;
            if (t < a.d) {
                if (a.q_is_f32) {
                    float x = reinterpret_cast<const float*>(a.q)[(size_t)r * a.d + t];
                    if (scale) x *= inv;
                    v = f32_to_bf16_rne(x);
                } else {
                    v = reinterpret_cast<const uint16_t*>(a.q)[(size_t)r * a.d + t];
                }
            }
            dst[t] = v;
        }
    }
    __syncthreads();

    // ---- phase 1: MFMA scores of this wave's 16-document tiles, running top-6 per lane
    const int ks32 = a.ld / 32; // 4 .. 32
    float ls[TINY_KL];
    int li[TINY_KL];
#pragma unroll
    for (int i = 0; i < TINY_KL; ++i) {
        ls[i] = -INFINITY;
        li[i] = IDX_NONE;
    }
    const int gw = blockIdx.x * 4 + wave;
    const uint16_t* brow = qs + c * a.ld + 8 * g;
    for (int tile = gw; tile < a.ntiles; tile += a.nwaves) {
        const uint16_t* arow = a.docs + ((size_t)tile * 16 + c) * a.ld + 8 * g;
        bf16x8 av[32]; // the whole K of the tile in flight: one memory round trip per tile
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (u < ks32) av[u] = *reinterpret_cast<const bf16x8*>(arow + 32 * u);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (u < ks32) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(brow + 32 * u);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[u], bv, acc, 0, 0, 0);
            }
        const int base = tile * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sc = (int64_t)(base + r) < a.ntotal ? acc[r] : -INFINITY;
            if (sc > ls[TINY_KL - 1]) list_insert<TINY_KL>(ls, li, sc, base + r);
        }
    }
    // ---- first selection level, inside the workgroup: the 16 lane lists of a query (4 waves x 4 g) -> its 8 best
    {
        const int o = c * 96 + (wave * 4 + g) * TINY_KL;
#pragma unroll
        for (int i = 0; i < TINY_KL; ++i) {
            wl_s[o + i] = ls[i];
            wl_i[o + i] = li[i];
        }
    }
    __syncthreads();
    float* gps = const_cast<float*>(a.m.part_s); // [16 q][nwg][8]
    int* gpi = const_cast<int*>(a.m.part_i);
    float* gbnd = const_cast<float*>(a.m.pre_bnd); // [16 q][nwg]
    {
        MergeArgs mw = a.m;
        mw.part_s = wl_s;
        mw.part_i = wl_i;
        mw.ncand = 96;
        mw.ll = TINY_KL;
        mw.pre_bnd = nullptr;
        mw.npre = 0;
        mw.bnd = nullptr;
        for (int q = wave; q < a.nq; q += 4) {
            merge_select_body<TINY_POOL>(mw, cand, q, lane, cand_s, &gbnd[(size_t)q * nwg + blockIdx.x]);
            if (lane < TINY_POOL) {
                gps[((size_t)q * nwg + blockIdx.x) * TINY_POOL + lane] = cand_s[q * TINY_POOL + lane];
                gpi[((size_t)q * nwg + blockIdx.x) * TINY_POOL + lane] = cand[q * TINY_POOL + lane];
            }
        }
    }
    // ---- ticket: release this workgroup's candidates, find out whether it is the last one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // every wave of the last workgroup reads other workgroups' candidates
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- phase 2 (last workgroup): final selection, exact re-score, rank, ignore filter
    MergeArgs m = a.m;
    m.qbuf = qs;
    m.out_s = res_s;
    m.out_i = res_i;
    m.out_packed = nullptr;
    if (tid == 0) {
        *a.ticket = 0u;                    // ready for the next launch on this stream
        if (m.nflag) *m.nflag = 0u;        // this call's flag counter
    }
    for (int q = wave; q < a.nq; q += 4) merge_select_body<TINY_POOL>(m, cand, q, lane);
    __syncthreads();
    // 64 / 8 = 8 queries per wave: waves 0 and 1 cover 16 queries; the k results go to LDS [q][m.k]
    if (wave < 2) rescore_rank_body<TINY_POOL, ElemBF16, L2>(m, cand, a.nq, wave, lane);
    __syncthreads();
    if (tid < a.nq) {
        const int q = tid;
        const int64_t banned = a.ignore ? a.ignore[q] : INT64_MIN;
        int w = 0;
        for (int t = 0; t < m.k && w < a.k_out; ++t) {
            const int64_t id = res_i[q * m.k + t];
            if (a.ignore && id == banned) continue;
            const float sc = res_s[q * m.k + t];
            const size_t o = (size_t)q * a.k_out + w;
            if (a.out_packed) {
                a.out_packed[2 * o] = (int64_t)__float_as_uint(sc);
                a.out_packed[2 * o + 1] = id;
            } else {
                a.out_s[o] = sc;
                a.out_i[o] = id;
            }
            ++w;
        }
    }
}

} // namespace mips
