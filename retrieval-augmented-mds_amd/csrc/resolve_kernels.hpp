// Exact resolution of flagged queries: what the margin check could not certify is settled by brute
// force on the canonical scores, at the speed the index streams from HBM.
//
// A query is flagged when the k-th exact score of its candidate pool is not provably above everything OUTSIDE the pool.
// The first versions re-scanned flagged queries with the widest lists (K' = 32) -- one more approximate scan plus one
// more margin check, a whole 128-query tile of MFMA work per handful of queries (13 - 24 ms for ONE flagged query on the
// three-segment scan of a 2^20 x 768 fp32 index), and queries could stay "unresolved".  Here instead:
//
//   exact_filter_kernel   streams the stored rows ONCE per batch of 8 flagged queries and computes, for every row, the
//                         canonical score itself -- the sequential fp64 sum of the exact products, the definition of
//                         include/mips_hip.h -- one lane per row, 8 accumulators per lane, the queries as fp64 in LDS;
//                         rows arrive through a per-wave LDS transpose (coalesced 128-byte row segments in, one row per
//                         lane out).  A row whose canonical key reaches the key of the query's current k-th result is
//                         appended to the query's hit list (the current top k are among them by construction).
//   resolve_finalize_kernel  ranks a query's hits by (key desc, id asc) and overwrites its result row.  More than
//                         RESOLVE_CAP hits (floods of exact ties) leave the first result in place, counted unresolved.
//
// Everything is sized on the device (flag list + count from compact_flags_kernel): no synchronisation, graph-capturable;
// with nothing flagged every workgroup reads the count and leaves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aux_kernels.hpp"

namespace mips {

constexpr int RESOLVE_QB = 8;     // flagged queries per pass over the index
constexpr int RESOLVE_WAVES = 8;  // per workgroup
constexpr int RESOLVE_CAP = 64;   // hits kept per query
constexpr int RESOLVE_MAX = 1024; // flagged queries resolved per search = 128 passes over the index (the tile re-scan costs more per
                                  // query even then); a search that flags more is served by the tile re-scan when it may
                                  // synchronise, left unresolved (counted) when it may not

struct ResolveArgs {
    const void* rows;     // stored rows the canonical score is defined on (bf16 / e4m3 index rows, fp32 rows of the fp32-exact index)
    const void* y;        // staged queries of ALL nq queries, same element type, row pitch ld
    int ld;               // elements per row (rows and queries), a multiple of 128 bytes' worth
    int64_t ntotal;
    const int* ids;       // [n] flagged query numbers, ascending
    const int* n_dev;     // n
    int max_n;            // resolve only if n <= max_n (<= RESOLVE_MAX): a search that flags more keeps its first results, counted unresolved
    const float* keyk;    // [nq] canonical key of the query's current k-th result (IP: score; L2: -distance)
    const double* qq;     // [nq] |q|^2 (L2)
    double phi;
    double* hit_d;        // [RESOLVE_MAX][RESOLVE_CAP] dot products of the hits
    int* hit_i;           // [RESOLVE_MAX][RESOLVE_CAP] their rows
    int* hit_n;           // [RESOLVE_MAX] hit counts (zeroed by the host)
    // finalize
    int k;
    int64_t idx_offset;
    float* out_s;         // [nq][k]
    int64_t* out_i;
    int64_t* out_packed;  // or [nq][k][2]
    unsigned* unresolved; // counter (zeroed by the host)
    // the ignore filter of Mips.search (sotasum/mips.py:388-398) when the search that flagged was the fused hook call: the
    // k (= k_out + 1) ranked hits lose the one equal to ignore[q] and are cut to k_out; output rows are k_out wide
    const int64_t* ignore = nullptr;
    int k_out = 0;        // 0: no filter, rows are k wide
};

template <bool L2>
__device__ __forceinline__ float resolve_key(double dot, double qq, double phi) {
    return L2 ? -(float)(qq + phi - 2.0 * dot) : (float)dot;
}

// RESOLVE_WAVES waves; wave w of workgroup b owns rows (b * RESOLVE_WAVES + w) * 64 .. + 63 of each grid stride
// (8 waves = two per SIMD: one wave alone cannot cover the LDS round trips between its fp64 chains)
// ELQ: element type of the staged queries (= EL except for the e4m3-documents / bf16-queries index)
template <typename EL, bool L2, typename ELQ = EL>
__global__ __launch_bounds__(64 * RESOLVE_WAVES) void exact_filter_kernel(ResolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = *a.n_dev;
    if (n == 0 || n > a.max_n) return;
    constexpr int PER = EL::PER16;               // elements per 16-byte chunk
    constexpr int TCH = 8;                       // chunks per row per tile step: 128 bytes
    double* yd = reinterpret_cast<double*>(smem);                                   // [RESOLVE_QB][ld] queries of the batch, fp64
    unsigned char* tiles = smem + (size_t)RESOLVE_QB * a.ld * sizeof(double);       // [waves][64 rows][TCH + 1 chunks]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4* tile = reinterpret_cast<u32x4*>(tiles) + wave * 64 * (TCH + 1);
    const int nchunk = a.ld / PER;               // 16-byte chunks per row, a multiple of TCH
    const typename EL::type* rows = reinterpret_cast<const typename EL::type*>(a.rows);
    const typename ELQ::type* ys = reinterpret_cast<const typename ELQ::type*>(a.y);
    constexpr int PERQ = ELQ::PER16;
    const int nchunkq = a.ld / PERQ;

    for (int j0 = 0; j0 < n; j0 += RESOLVE_QB) {
        __syncthreads(); // (the previous batch's queries are no longer read)
        // the batch's queries -> fp64 in LDS (rows past n: zeros)
        for (int t = tid; t < RESOLVE_QB * nchunkq; t += 64 * RESOLVE_WAVES) {
            const int j = t / nchunkq, c = t % nchunkq;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (j0 + j < n) v = *reinterpret_cast<const u32x4*>(ys + (size_t)a.ids[j0 + j] * a.ld + (size_t)c * PERQ);
#pragma unroll
            for (int e = 0; e < PERQ; ++e) yd[(size_t)j * a.ld + c * PERQ + e] = j0 + j < n ? (double)ELQ::get(v, e) : 0.0;
        }
        __syncthreads();
        float kk[RESOLVE_QB];
        double qn[RESOLVE_QB];
#pragma unroll
        for (int j = 0; j < RESOLVE_QB; ++j) {
            const int qid = j0 + j < n ? a.ids[j0 + j] : -1;
            kk[j] = qid >= 0 ? a.keyk[qid] : INFINITY;
            qn[j] = qid >= 0 && L2 ? a.qq[qid] : 0.0;
        }
        for (int64_t r0 = ((int64_t)blockIdx.x * RESOLVE_WAVES + wave) * 64; r0 < a.ntotal; r0 += (int64_t)gridDim.x * 64 * RESOLVE_WAVES) {
            const int64_t row = r0 + lane;
            double acc[RESOLVE_QB];
#pragma unroll
            for (int j = 0; j < RESOLVE_QB; ++j) acc[j] = 0.0;
            // coalesced: 8 lanes cover the 128-byte segment of one row, 8 rows per load instruction; the NEXT segment's loads
            // are in flight while this one is summed (4 waves per CU: nothing else hides the memory round trip)
            const typename EL::type* src[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int64_t rr = r0 + 8 * i + (lane >> 3);
                if (rr >= a.ntotal) rr = a.ntotal - 1; // (clamped: the lane's own row test drops it below)
                src[i] = rows + (size_t)rr * a.ld + (size_t)(lane & 7) * PER;
            }
            u32x4 in[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) in[i] = *reinterpret_cast<const u32x4*>(src[i]);
            for (int c0 = 0; c0 < nchunk; c0 += TCH) {
#pragma unroll
                for (int i = 0; i < 8; ++i) tile[(8 * i + (lane >> 3)) * (TCH + 1) + (lane & 7)] = in[i];
                if (c0 + TCH < nchunk) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) in[i] = *reinterpret_cast<const u32x4*>(src[i] + (size_t)(c0 + TCH) * PER);
                }
                __builtin_amdgcn_wave_barrier(); // (one wave: LDS operations complete in order)
#pragma unroll
                for (int c = 0; c < TCH; ++c) {
                    const u32x4 v = tile[lane * (TCH + 1) + c];
                    double x[PER];
#pragma unroll
                    for (int e = 0; e < PER; ++e) x[e] = (double)EL::get(v, e);
                    const double* yy = yd + (size_t)(c0 + c) * PER;
#pragma unroll
                    for (int j = 0; j < RESOLVE_QB; ++j)
#pragma unroll
                        for (int e = 0; e < PER; ++e) acc[j] += x[e] * yy[(size_t)j * a.ld + e]; // sequential in the column index
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (row < a.ntotal) {
#pragma unroll
                for (int j = 0; j < RESOLVE_QB; ++j) {
                    if (j0 + j < n && resolve_key<L2>(acc[j], qn[j], a.phi) >= kk[j]) {
                        const int pos = atomicAdd(&a.hit_n[j0 + j], 1);
                        if (pos < RESOLVE_CAP) {
                            a.hit_d[(size_t)(j0 + j) * RESOLVE_CAP + pos] = acc[j];
                            a.hit_i[(size_t)(j0 + j) * RESOLVE_CAP + pos] = (int)row;
                        }
                    }
                }
            }
        }
    }
}

// one wave per flagged query
template <bool L2>
__global__ __launch_bounds__(64) void resolve_finalize_kernel(ResolveArgs a) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const int nall = *a.n_dev;
    if (j >= nall) return;
    if (nall > a.max_n) { // more than this search resolves: nothing was computed (the grid only covers max_n queries)
        if (j == 0 && lane == 0) atomicAdd(a.unresolved, (unsigned)nall);
        return;
    }
    const int c = a.hit_n[j];
    const int q = a.ids[j];
    if (c > RESOLVE_CAP || c < a.k) { // a flood of ties (or nothing to rank: cannot happen, the current top k always hit)
        if (lane == 0) atomicAdd(a.unresolved, 1u);
        return;
    }
    const double dot = lane < c ? a.hit_d[(size_t)j * RESOLVE_CAP + lane] : 0.0;
    const int id = lane < c ? a.hit_i[(size_t)j * RESOLVE_CAP + lane] : IDX_NONE;
    const double qq = L2 ? a.qq[q] : 0.0;
    const float outv = L2 ? (float)(qq + a.phi - 2.0 * dot) : (float)dot;
    const float key = lane < c ? (L2 ? -outv : outv) : -INFINITY;
    int rank = 0;
    for (int t = 0; t < c; ++t) {
        const float ok = __shfl(key, t);
        const int oi = __shfl(id, t);
        rank += ranks_before(ok, oi, key, id) ? 1 : 0;
    }
    int pos = rank, width = a.k;
    bool keep = lane < c && rank < a.k;
    if (a.ignore != nullptr) { // (ids are unique: at most one hit is the banned one)
        const int64_t banned = a.ignore[q];
        const unsigned long long bb = __ballot(keep && (int64_t)id + a.idx_offset == banned);
        const int rb = bb != 0ull ? __shfl(rank, __ffsll((long long)bb) - 1) : 0x7fffffff;
        width = a.k_out;
        keep = keep && rank != rb;
        pos = rank - (rank > rb ? 1 : 0);
        keep = keep && pos < width;
    }
    if (keep) {
        const size_t o = (size_t)q * width + pos;
        if (a.out_packed) {
            a.out_packed[2 * o] = (int64_t)__float_as_uint(outv);
            a.out_packed[2 * o + 1] = (int64_t)id + a.idx_offset;
        } else {
            a.out_s[o] = outv;
            a.out_i[o] = (int64_t)id + a.idx_offset;
        }
    }
}

// The fall-back re-scan behind the exact pass (mips_hip.hip, rescan_on_stream with a gate): if it ran (gated count > 0), the
// queries IT still flags on true K' = 32 lists are what this search leaves unresolved.
__global__ void adopt_rescan_count_kernel(const int* gated_n, const unsigned* rescan_still_flagged, unsigned* unresolved) {
    if (*gated_n > 0) *unresolved = *rescan_still_flagged;
}

} // namespace mips
