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
//   exact_filter_mfma_kernel  (round 3; bf16-stored rows and the fp32-exact index) the same pass with an MFMA pre-filter: 16
//                         flagged queries per pass as the B operand of v_mfma_f32_16x16x32_bf16, the rows (bf16 rows, or the
//                         bf16 image rows_hi of fp32 rows) streamed as A operands straight from HBM; only rows whose APPROXIMATE
//                         score comes within the error bound of the query's k-th key get the canonical fp64 evaluation (a
//                         handful per query).  Same hit lists as the kernel above, at the rate the rows stream instead of the
//                         rate of 8 fp64 FMAs per row element (2.5 - 2.8 TB/s of bf16 rows), and twice the queries per pass.
//   resolve_finalize_kernel  ranks a query's hits by (key desc, id asc) and overwrites its result row.  More than
//                         RESOLVE_CAP hits (floods of exact ties) leave the first result in place, counted unresolved.
//
// Everything is sized on the device (flag list + count from compact_flags_kernel): no synchronisation, graph-capturable;
// with nothing flagged every workgroup reads the count and leaves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "aux_kernels.hpp"
#include "scan_kernel_v4.hpp" // (f32x4)

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
    // exact_filter_mfma_kernel: the bf16 rows the pre-filter multiplies (the index rows themselves, or rows_hi of an fp32-exact
    // index), their pitch in elements, and what bounds |approximate - exact| (the margin check's terms: aux_kernels.hpp)
    const uint16_t* frows = nullptr;
    int fld = 0;
    const double* xmax2 = nullptr; // max_i |x_i|^2
    const double* dres2 = nullptr; // max_i |x_i - bf16 x_i|^2 (fp32-exact index)
    double err_c = 0.0;            // MFMA accumulation: |approximate - exact product sum| <= err_c |q| max|x|
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

// ---------------------------------------------------------------------------------------------------------------------------
// The exact pass behind an MFMA pre-filter.  F32 = false: bf16 index (canonical operands = the stored bf16 rows and the staged
// bf16 queries = the filter's operands).  F32 = true: fp32-exact index (canonical operands = rows_f32 and the fp32 queries; the
// filter multiplies rows_hi = bf16(x) with bf16(q), rounded here).
// Workgroup = 8 waves; wave w of workgroup b owns the 16-row tiles (b * 8 + w) + t * gridDim.x * 8.  Per tile: fld / 32 MFMAs
// whose A fragments come straight from global memory (lane (c, g): 16 bytes of row c at k = 32 s + 8 g, two steps ahead) and
// whose B fragments come from the batch's queries in LDS (row pitch + 16 bytes: conflict-free 16-byte reads).
// A (row, query) pair is a CANDIDATE if  approximate dot >= (dot that would reach the k-th key) - e - rounding slack,
//   e = err_c |q| max|x|  [+ F32: dres |q| + (max|x| + dres) |q - bf16 q|],
// i.e. no pair the canonical test below would accept is ever skipped; candidates are evaluated canonically by the whole wave
// (products, exact in fp64, written to LDS by 64 lanes; summed in column order by one) and appended exactly as exact_filter_kernel
// appends them.
constexpr int RESOLVE_QM = 16;

template <bool L2, bool F32>
__global__ __launch_bounds__(64 * RESOLVE_WAVES) void exact_filter_mfma_kernel(ResolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = *a.n_dev;
    if (n == 0 || n > a.max_n) return;
    typedef typename std::conditional<F32, ElemF32, ElemBF16>::type EL; // canonical element type (rows and staged queries)
    constexpr int PER = EL::PER16;
    const int tid = threadIdx.x, wave = tid >> 6;
    const unsigned qpitch = (unsigned)a.fld * 2u + 16u;               // bytes per query row of the B image
    unsigned char* bq = smem;                                          // [RESOLVE_QM][qpitch]
    double* prod = reinterpret_cast<double*>(smem + ((RESOLVE_QM * qpitch + 15u) & ~15u)) + (size_t)wave * 64 * 8; // [waves][64 lanes][8]
    double* stat = reinterpret_cast<double*>(smem + ((RESOLVE_QM * qpitch + 15u) & ~15u)) + (size_t)RESOLVE_WAVES * 64 * 8; // [2][RESOLVE_QM]: |q|^2, |q - bf16 q|^2
    float* tdot_s = reinterpret_cast<float*>(stat + 2 * RESOLVE_QM);  // [RESOLVE_QM] candidate thresholds on the approximate dot
    float* kk_s = tdot_s + RESOLVE_QM;                                 // [RESOLVE_QM] k-th keys
    int* qid_s = reinterpret_cast<int*>(kk_s + RESOLVE_QM);            // [RESOLVE_QM] query numbers (-1: none)
    const typename EL::type* rows = reinterpret_cast<const typename EL::type*>(a.rows);
    const typename EL::type* ys = reinterpret_cast<const typename EL::type*>(a.y);
    const int nchunk = a.ld / PER;                                     // 16-byte chunks of a canonical row / query
    const int ks = a.fld / 32;                                         // k32-steps of the filter (fld: a multiple of 64)
    const int64_t ntile = (a.ntotal + 15) / 16;

    for (int j0 = 0; j0 < n; j0 += RESOLVE_QM) {
        __syncthreads(); // (the previous batch's image is no longer read)
        if (tid < 2 * RESOLVE_QM) stat[tid] = 0.0;
        if (tid < RESOLVE_QM) qid_s[tid] = j0 + tid < n ? a.ids[j0 + tid] : -1;
        __syncthreads();
        // the batch's queries -> bf16 image in LDS (zeros past the query's columns and for absent queries), |q|^2, |q - bf16 q|^2
        const int fchunk = a.fld / 8;                                  // 16-byte chunks of an image row
        for (int t = tid; t < RESOLVE_QM * fchunk; t += 64 * RESOLVE_WAVES) {
            const int j = t / fchunk, c = t % fchunk;
            const int qid = qid_s[j];
            u32x4 o = {0u, 0u, 0u, 0u};
            double s2 = 0.0, r2 = 0.0;
            if (qid >= 0) {
                if (F32) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) { // two fp32 chunks per bf16 chunk
                        const int cc = 2 * c + h;
                        if (cc < nchunk) {
                            const u32x4 v = *reinterpret_cast<const u32x4*>(ys + (size_t)qid * a.ld + (size_t)cc * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x = __uint_as_float(v[e]);
                                const uint16_t hb = f32_to_bf16_rne(x);
                                const double r = (double)x - (double)bf16_bits_to_f32(hb);
                                s2 += (double)x * (double)x;
                                r2 += r * r;
                                o[2 * h + (e >> 1)] |= (unsigned)hb << ((e & 1) * 16);
                            }
                        }
                    }
                } else if (c < nchunk) {
                    o = *reinterpret_cast<const u32x4*>(ys + (size_t)qid * a.ld + (size_t)c * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const double x = (double)ElemBF16::get(o, e);
                        s2 += x * x;
                    }
                }
            }
            *reinterpret_cast<u32x4*>(bq + (size_t)j * qpitch + (size_t)c * 16) = o;
            if (s2 != 0.0) atomicAdd(&stat[j], s2);
            if (r2 != 0.0) atomicAdd(&stat[RESOLVE_QM + j], r2);
        }
        __syncthreads();
        if (tid < RESOLVE_QM) {
            const int qid = qid_s[tid];
            float td = INFINITY, kk = INFINITY;
            if (qid >= 0) {
                kk = a.keyk[qid];
                if (kk < INFINITY) {
                    const double qn = sqrt(stat[tid]), xm = sqrt(*a.xmax2);
                    double e = a.err_c * qn * xm;
                    if (F32) {
                        const double dr = sqrt(*a.dres2);
                        e += dr * qn + (xm + dr) * sqrt(stat[RESOLVE_QM + tid]);
                    }
                    const double qq = L2 ? a.qq[qid] : 0.0;
                    const double need = L2 ? 0.5 * (qq + a.phi + (double)kk) : (double)kk; // canonical key >= kk  <=>  dot >= need (before rounding)
                    const double slack = 9.5367431640625e-07 * (fabs(need) + (L2 ? fabs(qq + a.phi) : 0.0)) + 1e-30; // 2^-20: the key's float rounding, generously
                    td = __double2float_rd(need - e - slack);
                    if (!(td == td)) td = -INFINITY; // (NaN: evaluate everything rather than skip anything)
                }
            }
            tdot_s[tid] = td;
            kk_s[tid] = kk;
        }
        __syncthreads();

        const unsigned ln = (unsigned)(tid & 63);
        const unsigned c = ln & 15u, g = ln >> 4;
        const float my_td = tdot_s[c];
        const unsigned char* bsrc = bq + (size_t)c * qpitch + g * 16u;
        for (int64_t tile = (int64_t)blockIdx.x * RESOLVE_WAVES + wave; tile < ntile; tile += (int64_t)gridDim.x * RESOLVE_WAVES) {
            int64_t arow = tile * 16 + c;
            if (arow >= a.ntotal) arow = a.ntotal - 1; // (clamped: such rows are dropped below)
            const uint16_t* asrc = a.frows + (size_t)arow * a.fld + g * 8u;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // two k-steps per iteration, the next two fragments in flight (32 waves per CU hide the rest)
            u32x4 a0 = *reinterpret_cast<const u32x4*>(asrc), a1 = *reinterpret_cast<const u32x4*>(asrc + 32);
            for (int s0 = 0; s0 < ks; s0 += 2) {
                u32x4 n0 = a0, n1 = a1;
                if (s0 + 2 < ks) {
                    n0 = *reinterpret_cast<const u32x4*>(asrc + (size_t)(s0 + 2) * 32);
                    n1 = *reinterpret_cast<const u32x4*>(asrc + (size_t)(s0 + 3) * 32);
                }
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(bsrc + (size_t)s0 * 64);
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(bsrc + (size_t)(s0 + 1) * 64);
                bf16x8 fa0, fa1, fb0, fb1;
                __builtin_memcpy(&fa0, &a0, 16);
                __builtin_memcpy(&fa1, &a1, 16);
                __builtin_memcpy(&fb0, &b0, 16);
                __builtin_memcpy(&fb1, &b1, 16);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, fb0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, fb1, acc, 0, 0, 0);
                a0 = n0;
                a1 = n1;
            }
            // acc[r] = approximate dot of row tile * 16 + 4 g + r with query c of the batch
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = tile * 16 + 4 * (int64_t)g + r;
                unsigned long long m = __ballot(row < a.ntotal && acc[r] >= my_td);
                while (m != 0ull) { // rare: one candidate at a time, the whole wave on it
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    const int64_t crow = tile * 16 + 4 * (int64_t)(l >> 4) + r;
                    const int cj = l & 15;
                    const int qid = qid_s[cj];
                    double dot = 0.0;
                    for (int c0 = 0; c0 < nchunk; c0 += 64) { // 64 chunks per round: products to LDS, lane 0 adds them in column order
                        const int cc = c0 + (int)ln;
                        if (cc < nchunk) {
                            const u32x4 xv = *reinterpret_cast<const u32x4*>(rows + (size_t)crow * a.ld + (size_t)cc * PER);
                            const u32x4 yv = *reinterpret_cast<const u32x4*>(ys + (size_t)qid * a.ld + (size_t)cc * PER);
#pragma unroll
                            for (int e = 0; e < PER; ++e) prod[ln * 8 + e] = (double)EL::get(xv, e) * (double)EL::get(yv, e); // exact in fp64
                        }
                        __builtin_amdgcn_wave_barrier();
                        if (ln == 0) {
                            const int lim = nchunk - c0 < 64 ? nchunk - c0 : 64;
                            for (int t = 0; t < lim; ++t)
#pragma unroll
                                for (int e = 0; e < PER; ++e) dot += prod[t * 8 + e];
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (ln == 0 && resolve_key<L2>(dot, L2 ? a.qq[qid] : 0.0, a.phi) >= kk_s[cj]) {
                        const int pos = atomicAdd(&a.hit_n[j0 + cj], 1);
                        if (pos < RESOLVE_CAP) {
                            a.hit_d[(size_t)(j0 + cj) * RESOLVE_CAP + pos] = dot;
                            a.hit_i[(size_t)(j0 + cj) * RESOLVE_CAP + pos] = (int)crow;
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
