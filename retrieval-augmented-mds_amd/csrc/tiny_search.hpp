// tiny_search_kernel: the reference's OWN call shape in ONE launch.
//
// sotasum's training / generation step searches B <= 16 queries against a ~10^4-row knowledge base
// (retriever_generator.py:145-153 -> mips.py:402-463; BASELINE config 1: N = 10 000, d = 768, B = 8, k = 5).  Through
// the general path that is five dependent launches (query staging, fused scan, select, exact re-score, ignore filter;
// six with the query normalisation) of a few microseconds each: 0.11 ms device-resident, all of it launch latency.
// Here one kernel does it all (prepare + search + ignore mask fused):
//
//   every workgroup   issues the loads of its waves' first 16-document tiles, THEN stages the <= 16 queries (optional
//                     row normalisation exactly as mips_l2_normalize: faiss.normalize_L2, mips.py:369-370, 521-525;
//                     rounding to the index's bf16, RNE) into LDS while those loads are in flight;
//   phase 1           each wave scores its tiles on v_mfma_f32_16x16x32_bf16 (A = 16 documents x 32 k per lane-load
//                     straight from global memory, B = the 16 staged queries), lane (c, g) keeps the running top-6 of
//                     documents 4 g .. 4 g + 3 of its tiles against query c (strict '>', ascending document order:
//                     lowest index wins ties); one wave per query then selects the workgroup's 8 best of the 32 lane
//                     lists (tiny_select8) and the workgroup takes a ticket;
//   phase 2           the LAST workgroup to finish (agent-scope release / acquire around the ticket) selects the 8 best
//                     candidates per query among the workgroups' pools, re-scores them exactly (the canonical
//                     sequential-fp64 score, computed in parallel where that is PROVABLY the same number:
//                     tiny_dot_chunk / tiny_cert_ok), ranks, applies the k + 1 / ignore filter of mips.py:388-398 and
//                     writes the results.
// Same candidate-pool logic, same canonical scores and the same margin check as the general path: results are
// bit-identical to it (tests/test_gpu_parity.py::test_tiny_search_*).
//
// Round-2 timing of the first version (79 workgroups x 4 waves, 10^4 x 768, 8 queries; time stamps in the kernel): query
// staging 4 us, scan 10 us (two dependent tile round trips), selection inside the workgroup 10 us, ticket 1.5 us, final
// selection 19.5 us, re-score 14.5 us = 60 us.  The selections popped 8 wave-wide arg-maxima out of per-lane sorted
// lists and the re-score ran one lane per candidate over 768 dependent fp64 additions; both are replaced here.
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
    int nwaves;            // gridDim.x * TINY_WAVES
                           // m.part_s / m.part_i: [16][gridDim.x][8] candidates of the workgroups, m.pre_bnd: [16][gridDim.x]
                           // their bounds, m.ncand = gridDim.x * 8, m.npre = gridDim.x, m.ll = INT_MAX (set by the host)
    unsigned* ticket;      // zeroed per launch
    const int64_t* ignore; // [nq] or nullptr: fetch k + 1, drop hits equal to ignore[q], keep k (mips.py:388-398)
    int k_out;             // results per query written to out_* (k); m.k = k or k + 1
    float* out_s;          // final outputs [nq][k_out] (device)
    int64_t* out_i;
    int64_t* out_packed;   // or the packed all-gather payload [nq][k_out][2]
    int force_slow;        // test knob ("tiny" = 2): take the fall-back paths (pop selection, sequential re-score) everywhere
    // fp32-exact index (template F32): docs = bf16(x) alone at the bf16 kernels' row pitch ld (mips_index::rows_hi, stage 1 of the
    // two-stage search), rows_f32 = the fp32 rows [capacity][plane] the exact re-score runs on; the margin check is widened by
    // the representation error |x - bf16 x| |q| + |bf16 x| |q - bf16 q| (m.dres2; |q - bf16 q|^2 is computed here)
    const float* rows_f32;
    int plane;
    // stream-ordered certification (resolve_kernels.hpp): the last workgroup writes the flagged queries' numbers + count, clears
    // the hit counters and -- only when something was flagged -- copies the staged queries to global memory for the exact pass
    int* res_ids;          // [nq] or nullptr (no hand-off)
    int* res_cnt;
    int* res_hit_n;        // [>= 16]
    unsigned* res_unres;
    void* q_out;           // [nq][ld] bf16 (pitch ld) or, F32, [nq][plane] float32
#ifdef MIPS_EXPERIMENTAL
    unsigned long long* dbg; // [gridDim.x][16] phase time stamps (100 MHz), or nullptr
#endif
    MergeArgs m;           // part_s / part_i / ncand / ll / docs / ld / k / metric / phi / idx_offset / margin fields;
                           // qbuf, out_s, out_i are set by the kernel (LDS)
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TINY_KL = 6;     // entries per lane list
constexpr int TINY_POOL = 8;   // re-score pool
constexpr int TINY_MAXK = 6;   // k (or k + 1) <= 6
constexpr int TINY_WAVES = 8;  // per workgroup, two per SIMD
constexpr int TINY_THREADS = 64 * TINY_WAVES;
constexpr int TINY_LISTS = 4 * TINY_WAVES;       // lane lists per (query, workgroup)
constexpr int TINY_WL = TINY_LISTS * TINY_KL;    // 192 candidates per (query, workgroup)
constexpr int TINY_MAX_WG = 256;                 // one per CU; the final selection holds 8 x 256 / 64 = 32 candidates per lane

// dynamic LDS of the kernel for row pitch ld
constexpr int tiny_lds_bytes(int ld, int plane = 0) {
    return 16 * ld * 2                 // staged queries
           + 16 * 4 + 16 + 16 * 8 + 64 // margin flags, their count, |q - bf16 q|^2, the final level's bound
           + 16 * plane * 4            // fp32-exact index: the staged queries as float32
           + 2 * 16 * TINY_WL * 4      // lane lists (scores, ids)
           + 2 * TINY_WAVES * 64 * 4   // survivors of the threshold test, per wave
           + 2 * 16 * TINY_POOL * 4    // cand, cand_s
           + 16 * 8 * 4 + 16 * 8 * 8   // res_s, res_i
           + 16 * TINY_POOL * 8        // canonical dot products
           + 16 * 8                    // |q|^2
           + 16 * 4 + 16;              // margin bounds of the queries, max |x|^2
}

#ifdef MIPS_EXPERIMENTAL
#define TINY_STAMP(i)                                                                  \
    do {                                                                               \
        if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 16 + (i)] = wall_clock64();  \
    } while (0)
#else
#define TINY_STAMP(i) \
    do {              \
    } while (0)
#endif

// Cross-lane reductions on DPP (data-parallel primitives: a lane permutation folded into the VALU operand fetch, a few
// cycles) instead of __shfl_xor (ds_bpermute_b32 through the LDS crossbar, ~100 cycles of latency each: six dependent
// ones per wave reduction made the first version's selections latency chains of several microseconds).
//   quad_perm(1,0,3,2) = lane ^ 1, quad_perm(2,3,0,1) = lane ^ 2, row_half_mirror = 7 - lane within 8, row_mirror =
//   15 - lane within 16: after the four steps every lane of a 16-lane row holds the row's reduction (commutative ops).
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
// row_shr:1 (lane l reads lane l - 1 of its 16-lane row) and row_ror:9 (lane l reads lane (l - 9) mod 16): together they rotate
// a value through the 8 lanes of a group -- sub -> sub + 1 for sub < 7, 7 -> 0 (tiny_seq_dot_f32)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_ROR9 = 0x129;
template <int CTRL>
__device__ __forceinline__ unsigned tiny_dpp(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float tiny_dpp(float v) {
    return __uint_as_float(tiny_dpp<CTRL>(__float_as_uint(v)));
}
template <int CTRL>
__device__ __forceinline__ double tiny_dpp(double v) {
    const unsigned lo = tiny_dpp<CTRL>((unsigned)__double2loint(v)), hi = tiny_dpp<CTRL>((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ float tiny_wave_max(float v) { // every lane gets the maximum of the 64
    v = fmaxf(v, tiny_dpp<DPP_XOR1>(v));
    v = fmaxf(v, tiny_dpp<DPP_XOR2>(v));
    v = fmaxf(v, tiny_dpp<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, tiny_dpp<DPP_ROW_MIRROR>(v));
    const float r0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
    const float r1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
    const float r2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    const float r3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// The 8 best (score desc, id asc) of a candidate array made of nlist SORTED lists of LL entries (entry e of list l at
// [l * LL + e], best first; padding = (-inf, IDX_NONE)), one wave per call.
//   T = the 8th largest list head.  Eight candidates (those heads) score >= T, so nothing below T is among the 8 best:
//   the candidates >= T ("survivors", typically 8 .. 20) are compacted into LDS with ballots and ranked by counting.
// Returns false (nothing written) when more than 64 candidates survive -- many equal scores; the caller then takes the
// general pop selection (merge_select_body).  lb = this lane's share of the bound on what EARLIER levels excluded.
// Writes out_i[0..8) (+ out_s), pads (-inf, IDX_NONE), and *out_bnd exactly as merge_select_body does.
// All loads are issued before the first use (clamped addresses instead of predicated loads: no branch per load).
// out_bnd2 != nullptr: the two parts of the bound go to two words -- *out_bnd = what EARLIER levels excluded (lb), *out_bnd2 =
// the last pool member's score (what THIS level excluded) -- for the fp32-exact index, where the two are scores of different
// accuracy (MFMA scores of bf16 operands / refined fp32 scores) and the margin check prices them differently.
template <int LL, int NH, int NC>
__device__ __forceinline__ bool tiny_select8(const float* cs, const int* ci, int nlist, float lb, float* surv_s, int* surv_i, int lane,
                                             float* out_s, int* out_i, float* out_bnd, float* out_bnd2 = nullptr) {
    const int ncand = nlist * LL;
    float sv[NC];
    int iv[NC];
#pragma unroll
    for (int u0 = 0; u0 < NC; u0 += 8) {
        if (u0 * 64 < ncand) { // (uniform)
#pragma unroll
            for (int u = u0; u < u0 + 8; ++u)
                if (u < NC) {
                    const int c = lane + 64 * u;
                    const int cc = c < ncand ? c : ncand - 1;
                    sv[u] = cs[cc];
                    iv[u] = ci[cc];
                }
        } else {
#pragma unroll
            for (int u = u0; u < u0 + 8; ++u)
                if (u < NC) {
                    sv[u] = -INFINITY;
                    iv[u] = IDX_NONE;
                }
        }
    }
    float h[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int l = lane + 64 * j;
        h[j] = cs[(size_t)(l < nlist ? l : nlist - 1) * LL];
    }
#pragma unroll
    for (int j = 0; j < NH; ++j) h[j] = lane + 64 * j < nlist ? h[j] : -INFINITY;
#pragma unroll
    for (int u = 0; u < NC; ++u)
        if (lane + 64 * u >= ncand) {
            sv[u] = -INFINITY;
            iv[u] = IDX_NONE;
        }
    float T = -INFINITY;
    for (int r = 0; r < 8; ++r) {
        float m = h[0];
#pragma unroll
        for (int j = 1; j < NH; ++j) m = fmaxf(m, h[j]);
        const float M = tiny_wave_max(m);
        T = M;
        if (!(M > -INFINITY)) break; // fewer than 8 heads: everything valid survives
        const unsigned long long owners = __ballot(m == M);
        if (lane == __ffsll((long long)owners) - 1) { // one owner retires one instance
            bool done = false;
#pragma unroll
            for (int j = 0; j < NH; ++j) {
                const bool hit = !done && h[j] == M;
                h[j] = hit ? -INFINITY : h[j];
                done = done || hit;
            }
        }
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    int count = 0;
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const bool pr = iv[u] != IDX_NONE && sv[u] >= T;
        const unsigned long long b = __ballot(pr);
        const int pos = count + __popcll(b & below);
        if (pr && pos < 64) {
            surv_s[pos] = sv[u];
            surv_i[pos] = iv[u];
        }
        count += __popcll(b);
    }
    if (count > 64) return false;
    if (lane >= count) { // pad to whole groups of 4 for the ranking loop
        surv_s[lane] = -INFINITY;
        surv_i[lane] = IDX_NONE;
    }
    __builtin_amdgcn_wave_barrier(); // (LDS operations of one wave complete in order; this keeps the compiler from moving them)
    const float ms = surv_s[lane];
    const int mi = surv_i[lane];
    int rank = 0;
    for (int j = 0; j < count; j += 4) {
        const f32x4 os = *reinterpret_cast<const f32x4*>(surv_s + j);
        const i32x4 oi = *reinterpret_cast<const i32x4*>(surv_i + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) rank += ranks_before(os[e], oi[e], ms, mi) ? 1 : 0;
    }
    if (lane < count && rank < TINY_POOL) {
        out_i[rank] = mi;
        if (out_s) out_s[rank] = ms;
    }
    const int nsel = count < TINY_POOL ? count : TINY_POOL;
    if (lane >= nsel && lane < TINY_POOL) {
        out_i[lane] = IDX_NONE;
        if (out_s) out_s[lane] = -INFINITY;
    }
    if (out_bnd != nullptr) {
        lb = tiny_wave_max(lb);
        float s7 = -INFINITY;
        if (count >= TINY_POOL) {
            const unsigned long long b7 = __ballot(lane < count && rank == TINY_POOL - 1);
            s7 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ms), __ffsll((long long)b7) - 1));
        }
        if (lane == 0) {
            if (out_bnd2 != nullptr) {
                *out_bnd = lb;
                *out_bnd2 = s7;
            } else {
                *out_bnd = fmaxf(s7, lb);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// Canonical score = the SEQUENTIAL fp64 sum of the products x[t] * y[t] (include/mips_hip.h).  Eight lanes per
// (query, candidate) pair accumulate strided 16-byte chunks instead and the partial sums are added in a tree -- a
// different order, but the SAME number whenever no addition rounds.  Certificate (per pair, from the products' fp32 bit
// patterns): every product of two bf16 values has <= 16 significant bits, i.e. is an integer multiple of
// 2^(e - 15), e = its exponent; with e_min / e_max the extreme exponents of the non-zero products, every partial sum of
// at most 1024 products is an integer multiple of 2^(e_min - 15) below 2^(e_max + 11) in magnitude -- exactly
// representable in fp64's 53 bits when e_max - e_min <= 27 (all products normal and finite in fp32, where the product
// itself is then exact too).  Pairs that fail the certificate are summed sequentially by one lane.
struct TinyCert {
    unsigned mx, mn; // max of |p| bits, min of (|p| bits - 1) (a zero product wraps to 0xffffffff)
};
__device__ __forceinline__ void tiny_dot_chunk(const u32x4& xv, const u32x4& yv, double& acc, TinyCert& ce) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xe = ElemBF16::get(xv, e);
        const float ye = ElemBF16::get(yv, e);
        const float pr = xe * ye;
        const unsigned bits = __float_as_uint(pr) & 0x7fffffffu;
        ce.mx = ce.mx > bits ? ce.mx : bits;
        const unsigned b1 = bits - 1u;
        ce.mn = ce.mn < b1 ? ce.mn : b1;
        acc += (double)pr;
    }
}
template <int CTRL>
__device__ __forceinline__ void tiny_fold(double& acc, TinyCert& ce) {
    acc += tiny_dpp<CTRL>(acc);
    const unsigned omx = tiny_dpp<CTRL>(ce.mx), omn = tiny_dpp<CTRL>(ce.mn);
    ce.mx = ce.mx > omx ? ce.mx : omx;
    ce.mn = ce.mn < omn ? ce.mn : omn;
}
__device__ __forceinline__ bool tiny_cert_ok(const TinyCert& ce, bool denorm_ok) {
    if (ce.mn == 0xffffffffu) return true; // every product is zero
    const unsigned emax = ce.mx >> 23, emin = (ce.mn + 1u) >> 23;
    return denorm_ok && emin >= 1u && emax <= 254u && emax - emin <= 27u;
}
__device__ __forceinline__ double tiny_dot_sequential(const uint16_t* x, const uint16_t* y, int ld) {
    double dot = 0.0;
    for (int c = 0; c < ld / 8; ++c) {
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + c * 8);
        const u32x4 yv = *reinterpret_cast<const u32x4*>(y + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dot += (double)ElemBF16::get(xv, e) * (double)ElemBF16::get(yv, e);
    }
    return dot;
}

// this lane's share of max_w pre_bnd[w] (all loads first: a load-and-use loop is one memory round trip per iteration)
template <int NH>
__device__ __forceinline__ float tiny_lb(const float* pb, int npre, int lane) {
    float v[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
        const int w = lane + 64 * j;
        v[j] = pb[w < npre ? w : npre - 1];
    }
    float lb = -INFINITY;
#pragma unroll
    for (int j = 0; j < NH; ++j) lb = fmaxf(lb, lane + 64 * j < npre ? v[j] : -INFINITY);
    return lb;
}

// The canonical score on fp32 operands: the SEQUENTIAL fp64 sum of the (exact) products, which rounds at nearly every step -- no
// re-association is the same number, so the parallel-sum certificate of the bf16 path (tiny_cert_ok) has nothing to offer here.
// Eight lanes per pair still load the row coalesced (lane `sub` owns 16-byte chunks sub, sub + 8, ...: one memory round trip
// for the whole row), and the ONE live accumulator travels through them in column order: at step (t, s) lane s adds its
// chunk t (four FMAs, in order) and hands the sum to lane s + 1 by DPP (7 -> 0 with row_ror:9).  Every lane executes every step
// on whatever it holds -- only the live lane's value is ever read.  768 columns = 192 steps of 4 dependent v_fma_f64 + 2 DPP
// moves, ~4 us, two waves per SIMD filling each other's latency.  x: global or LDS (generic), y: LDS.  Result on sub == 0.
__device__ __forceinline__ double tiny_seq_dot_f32(const float* x, const float* y, int nchunk, int sub) {
    double acc = 0.0;
    // the whole row (<= 1024 columns = 4 groups of 64 chunks) is requested before the first use: one memory round trip
    u32x4 xr[8], x1[8], x2[8], x3[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (8 * t < nchunk) xr[t] = *reinterpret_cast<const u32x4*>(x + (sub + 8 * t) * 4);
        if (64 + 8 * t < nchunk) x1[t] = *reinterpret_cast<const u32x4*>(x + (64 + sub + 8 * t) * 4);
        if (128 + 8 * t < nchunk) x2[t] = *reinterpret_cast<const u32x4*>(x + (128 + sub + 8 * t) * 4);
        if (192 + 8 * t < nchunk) x3[t] = *reinterpret_cast<const u32x4*>(x + (192 + sub + 8 * t) * 4);
    }
#pragma unroll 1
    for (int h0 = 0; h0 < nchunk; h0 += 64) { // groups of 64 chunks = 256 columns: a LOOP (code size, see the staging comment)
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (h0 + 8 * t < nchunk) { // (uniform)
                const u32x4 yv = *reinterpret_cast<const u32x4*>(y + (h0 + sub + 8 * t) * 4);
                double xd[4], yd[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xd[e] = (double)__uint_as_float(xr[t][e]);
                    yd[e] = (double)__uint_as_float(yv[e]);
                }
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    double tmp = acc;
#pragma unroll
                    for (int e = 0; e < 4; ++e) tmp = __builtin_fma(xd[e], yd[e], tmp); // (the product is exact in fp64: one rounding, as "acc += x * y")
                    acc = st == 7 ? tiny_dpp<DPP_ROW_ROR9>(tmp) : tiny_dpp<DPP_ROW_SHR1>(tmp);
                }
            }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            xr[t] = x1[t];
            x1[t] = x2[t];
            x2[t] = x3[t];
        }
    }
    return acc;
}

template <bool L2, bool F32>
__global__ __launch_bounds__(TINY_THREADS, 1) void tiny_search_kernel(TinyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* qs = reinterpret_cast<uint16_t*>(smem);                            // [16][ld] staged queries
    float* wl_s = reinterpret_cast<float*>(smem + 16 * a.ld * 2);                // [16][TINY_WL] this workgroup's lane lists
    int* wl_i = reinterpret_cast<int*>(wl_s + 16 * TINY_WL);                     // [16][TINY_WL]
    float* surv_s = reinterpret_cast<float*>(wl_i + 16 * TINY_WL);               // [TINY_WAVES][64]
    int* surv_i = reinterpret_cast<int*>(surv_s + TINY_WAVES * 64);              // [TINY_WAVES][64]
    int* cand = surv_i + TINY_WAVES * 64;                                        // [16][TINY_POOL]
    float* cand_s = reinterpret_cast<float*>(cand + 16 * TINY_POOL);             // [16][TINY_POOL]
    float* res_s = cand_s + 16 * TINY_POOL;                                      // [16][8]
    int64_t* res_i = reinterpret_cast<int64_t*>(res_s + 16 * 8);                 // [16][8] (8-byte aligned)
    double* dots = reinterpret_cast<double*>(res_i + 16 * 8);                    // [16][TINY_POOL]
    double* qq_s = dots + 16 * TINY_POOL;                                        // [16]
    float* bnd_s = reinterpret_cast<float*>(qq_s + 16);                          // [16] what the final pool may have excluded
    double* xmax2_s = reinterpret_cast<double*>(bnd_s + 16);                     // [2] copies of *m.xmax2, *m.dres2 (loaded up front)
    unsigned char* flag_s = reinterpret_cast<unsigned char*>(xmax2_s + 2);       // [16] margin flags of the queries (64 bytes reserved)
    int* nfl_s = reinterpret_cast<int*>(flag_s + 64);                            // [1] their count (16 bytes reserved)
    double* qerr2_s = reinterpret_cast<double*>(flag_s + 80);                    // [16] |q - bf16 q|^2 (F32)
    float* bnd2_s = reinterpret_cast<float*>(qerr2_s + 16);                      // [16] the final level's own bound (F32; 64 bytes)
    float* qf = bnd2_s + 16;                                                     // [16][plane] staged queries as float32 (F32)
    __shared__ int is_last;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int nwg = gridDim.x;
    TINY_STAMP(0);
#ifdef MIPS_EXPERIMENTAL
    if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 16 + 12] = clock64();
#endif

    // ---- the first tile's documents: the whole K of the tile in flight, one memory round trip, issued before anything else
    const int ks128 = a.ld / 128; // 1 .. 8 groups of four 32-k steps
    const int gw = blockIdx.x * TINY_WAVES + wave;
    bf16x8 av[32];
    {
        const int t0 = gw < a.ntiles ? gw : a.ntiles - 1; // (idle waves load a valid tile and drop it)
        const uint16_t* arow = a.docs + ((size_t)t0 * 16 + c) * a.ld + 8 * g;
#pragma unroll
        for (int u4 = 0; u4 < 8; ++u4)
            if (u4 < ks128) {
#pragma unroll
                for (int u = 4 * u4; u < 4 * u4 + 4; ++u) av[u] = *reinterpret_cast<const bf16x8*>(arow + 32 * u);
            }
    }

    TINY_STAMP(1);
    double xm = 0.0, xres = 0.0;
    if (tid == 0 && a.m.nflag != nullptr) {
        xm = *a.m.xmax2; // (margin check: the load travels with the documents)
        if (F32) xres = *a.m.dres2;
    }
    // ---- stage the queries: wave w takes rows w and w + 8, one wave per row and the same per-lane order of the sum of
    // squares as mips_l2_normalize; the row's loads are in flight together with the documents (one memory round trip).
    // Written for FEW INSTRUCTIONS, like everything below: this kernel runs every line of its code once, from a cold
    // instruction cache -- measured ~5 ns per static instruction on the path, whatever the instruction does.
    {
        unsigned* z = reinterpret_cast<unsigned*>(qs + a.nq * a.ld); // rows nq .. 15: zeros
        const int nz = (16 - a.nq) * a.ld / 2;
        for (int t = tid; t < nz; t += TINY_THREADS) z[t] = 0u;
    }
    TINY_STAMP(2);
#pragma unroll 1
    for (int r = wave; r < a.nq; r += TINY_WAVES) {
        const int nfull = a.d >> 6; // whole groups of 64 columns; the (d % 64) others are the "tail" group
        const int tt = lane + 64 * nfull;
        float xv[16];
        float xt = 0.f;
        if (a.q_is_f32) {
            const float* src = reinterpret_cast<const float*>(a.q) + (size_t)r * a.d + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < nfull) xv[j] = src[64 * j];
            if (tt < a.d) xt = src[64 * nfull];
        } else {
            const uint16_t* src = reinterpret_cast<const uint16_t*>(a.q) + (size_t)r * a.d + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < nfull) xv[j] = bf16_bits_to_f32(src[64 * j]); // (rounding a bf16 value to bf16 below returns it)
            if (tt < a.d) xt = bf16_bits_to_f32(src[64 * nfull]);
        }
        float inv = 1.0f;
        if (a.normalize) { // (uniform; float32 source only: checked on the host)
            float nr = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < nfull) nr += xv[j] * xv[j];
            nr += xt * xt; // (+0 where the lane has no tail column)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nr += __shfl_xor(nr, off);
            if (nr > 0.f) inv = 1.0f / sqrtf(nr); // rows of norm 0 stay as they are (faiss fvec_renorm_L2)
        }
        uint16_t* dst = qs + r * a.ld + lane;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < nfull) dst[64 * j] = f32_to_bf16_rne(xv[j] * inv);
        for (int t = tt; t < a.ld; t += 64) qs[r * a.ld + t] = t < a.d ? f32_to_bf16_rne(xt * inv) : (uint16_t)0; // tail, padding
        if (F32) { // the same rows as float32 (what mips_l2_normalize leaves / the caller's values): operands of the exact re-score
            float* dstf = qf + r * a.plane + lane;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < nfull) dstf[64 * j] = xv[j] * inv;
            for (int t = tt; t < a.plane; t += 64) qf[r * a.plane + t] = t < a.d ? xt * inv : 0.f;
        }
    }
    TINY_STAMP(3);
    if (tid == 0) {
        xmax2_s[0] = xm;
        xmax2_s[1] = xres;
    }
    __syncthreads();
    TINY_STAMP(4);

    // ---- phase 1: MFMA scores of this wave's 16-document tiles, running top-6 per lane
    float ls[TINY_KL];
    int li[TINY_KL];
#pragma unroll
    for (int i = 0; i < TINY_KL; ++i) {
        ls[i] = -INFINITY;
        li[i] = IDX_NONE;
    }
    const uint16_t* brow = qs + c * a.ld + 8 * g;
    for (int tile = gw; tile < a.ntiles; tile += a.nwaves) {
        if (tile != gw) {
            const uint16_t* arow = a.docs + ((size_t)tile * 16 + c) * a.ld + 8 * g;
#pragma unroll
            for (int u4 = 0; u4 < 8; ++u4)
                if (u4 < ks128) {
#pragma unroll
                    for (int u = 4 * u4; u < 4 * u4 + 4; ++u) av[u] = *reinterpret_cast<const bf16x8*>(arow + 32 * u);
                }
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u4 = 0; u4 < 8; ++u4)
            if (u4 < ks128) {
                bf16x8 bv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = *reinterpret_cast<const bf16x8*>(brow + 32 * (4 * u4 + e));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[4 * u4 + e], bv[e], acc, 0, 0, 0);
            }
        const int base = tile * 16 + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sc = (int64_t)(base + r) < a.ntotal ? acc[r] : -INFINITY;
            if (sc > ls[TINY_KL - 1]) list_insert<TINY_KL>(ls, li, sc, base + r);
        }
    }
    TINY_STAMP(5);
    // ---- first selection level, inside the workgroup: the 32 lane lists of a query (8 waves x 4 g) -> its 8 best
    {
        const int o = c * TINY_WL + (wave * 4 + g) * TINY_KL;
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
    for (int q = wave; q < a.nq; q += TINY_WAVES) {
        const float* cs = wl_s + q * TINY_WL;
        const int* ci = wl_i + q * TINY_WL;
        const size_t o = ((size_t)q * nwg + blockIdx.x) * TINY_POOL;
        // a document DROPPED from a full lane list scored at most that list's last entry
        float lb = -INFINITY;
        if (lane < TINY_LISTS && ci[lane * TINY_KL + TINY_KL - 1] != IDX_NONE) lb = cs[lane * TINY_KL + TINY_KL - 1];
        bool ok = false;
        if (!a.force_slow)
            ok = tiny_select8<TINY_KL, 1, TINY_WL / 64>(cs, ci, TINY_LISTS, lb, surv_s + wave * 64, surv_i + wave * 64, lane,
                                                         F32 ? cand_s + q * TINY_POOL : gps + o, F32 ? cand + q * TINY_POOL : gpi + o,
                                                         &gbnd[(size_t)q * nwg + blockIdx.x]);
        if (!ok) {
            MergeArgs mw = a.m;
            mw.part_s = wl_s;
            mw.part_i = wl_i;
            mw.ncand = TINY_WL;
            mw.ll = TINY_KL;
            mw.pre_bnd = nullptr;
            mw.npre = 0;
            mw.bnd = nullptr;
            merge_select_body<TINY_POOL>(mw, cand, q, lane, cand_s, &gbnd[(size_t)q * nwg + blockIdx.x]);
            if (!F32 && lane < TINY_POOL) {
                gps[o + lane] = cand_s[q * TINY_POOL + lane];
                gpi[o + lane] = cand[q * TINY_POOL + lane];
            }
        }
        if (F32) {
            // fp32-exact index: the workgroup's 8 candidates leave with REFINED scores -- fp32 rows x fp32 query, summed in fp64
            // (any order: an approximation good to 2^-24 relative once stored as float32) -- instead of the MFMA scores of the
            // bf16-rounded operands (2^-9).  The final level then ranks the workgroups' candidates by scores that differ from the
            // canonical ones by rounding only, so what IT excludes needs a margin of 2^-23 |q| max|x|, not the representation error;
            // what the workgroups excluded by MFMA score (gbnd) lies hundreds of ranks below the top and can afford the wide margin.
            // 8 lanes per candidate, the whole row in flight in one memory round trip.
            __builtin_amdgcn_wave_barrier();
            const int pr = lane >> 3, sub = lane & 7;
            const int id = cand[q * TINY_POOL + pr];
            const float* x = a.rows_f32 + (size_t)(id != IDX_NONE ? id : 0) * a.plane;
            const float* y = qf + (size_t)q * a.plane;
            double acc = 0.0;
            {   // chunks sub, sub + 8, ...: two groups of 16 per lane (<= 1024 columns), BOTH requested before the first use
                const int nch = a.plane / 4; // a multiple of 16
                u32x4 xa[16], xb[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    if (8 * t < nch) xa[t] = *reinterpret_cast<const u32x4*>(x + (sub + 8 * t) * 4);
                    if (128 + 8 * t < nch) xb[t] = *reinterpret_cast<const u32x4*>(x + (128 + sub + 8 * t) * 4);
                }
#pragma unroll 1
                for (int h0 = 0; h0 < nch; h0 += 128) {
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (h0 + 8 * t < nch) {
                            const u32x4 yv = *reinterpret_cast<const u32x4*>(y + (h0 + sub + 8 * t) * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc += (double)__uint_as_float(xa[t][e]) * (double)__uint_as_float(yv[e]);
                        }
#pragma unroll
                    for (int t = 0; t < 16; ++t) xa[t] = xb[t];
                }
            }
            acc += tiny_dpp<DPP_XOR1>(acc);
            acc += tiny_dpp<DPP_XOR2>(acc);
            acc += tiny_dpp<DPP_HALF_MIRROR>(acc);
            const float rs = id != IDX_NONE ? (float)acc : -INFINITY;
            float* sv = surv_s + wave * 64;
            int* si = surv_i + wave * 64;
            if (sub == 0) {
                sv[pr] = rs;
                si[pr] = id;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < TINY_POOL) { // re-sort by the refined score: the final level expects every list best first
                int rank = 0;
#pragma unroll
                for (int t = 0; t < TINY_POOL; ++t) // (the empty slots -- all (-inf, IDX_NONE) -- are ordered by position: every rank is taken once)
                    rank += (ranks_before(sv[t], si[t], sv[lane], si[lane]) || (si[t] == si[lane] && t < lane)) ? 1 : 0;
                gps[o + rank] = sv[lane];
                gpi[o + rank] = si[lane];
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    TINY_STAMP(6);
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
    TINY_STAMP(7);
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // every wave of the last workgroup reads other workgroups' candidates
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- phase 2 (last workgroup): final selection, exact re-score, rank, ignore filter
    MergeArgs m = a.m;
    m.qbuf = qs;
    m.out_s = res_s;
    m.out_i = res_i;
    m.out_packed = nullptr;
    if (m.bnd != nullptr) m.bnd = bnd_s; // the bound travels from the selection to the margin check inside this workgroup
    m.xmax2 = xmax2_s;
    if (m.flag != nullptr) m.flag = flag_s; // (the flags stay in this workgroup too: the hand-off below reads them)
    if (F32) {
        m.dres2 = xmax2_s + 1;
        m.qerr2 = qerr2_s;
        if (m.bnd != nullptr) {
            m.bnd2 = bnd2_s;
            m.err_c2 = 1.1920928955078125e-07; // refined (fp64-summed, float32-stored) against canonical scores: rounding only
        }
        if (tid < 16) bnd2_s[tid] = -INFINITY; // (the fall-back selection reports one combined bound through m.bnd)
        __syncthreads();
    }
    if (tid == 0) {
        *a.ticket = 0u;                    // ready for the next launch on this stream
        if (m.nflag) *m.nflag = 0u;        // this call's flag counter
    }
    // f32 products are exact only while they stay normal, and a FLUSHED product would pass for a zero: the certificate needs
    // f32 denormals kept (MODE.FP_DENORM bits 4..5 = 3: sources and results); otherwise every pair is summed sequentially
    const bool denorm_ok = !a.force_slow && ((__builtin_amdgcn_s_getreg(1 | (4 << 6) | ((2 - 1) << 11)) & 3u) == 3u);
    for (int q = wave; q < a.nq; q += TINY_WAVES) {
        const float* cs = m.part_s + (size_t)q * m.ncand;
        const int* ci = m.part_i + (size_t)q * m.ncand;
        bool ok = false;
        if (!a.force_slow) { // (instances by workgroup count: the loads and ballots of absent candidates are instructions too)
            float* sv_s = surv_s + wave * 64;
            int* sv_i = surv_i + wave * 64;
            float* bo = m.bnd != nullptr ? &m.bnd[q] : nullptr;
            float* bo2 = F32 && m.bnd != nullptr ? &bnd2_s[q] : nullptr;
            const float* pb = m.pre_bnd + (size_t)q * m.npre;
            if (nwg <= 64) ok = tiny_select8<TINY_POOL, 1, 8>(cs, ci, nwg, tiny_lb<1>(pb, m.npre, lane), sv_s, sv_i, lane, nullptr, cand + q * TINY_POOL, bo, bo2);
            else if (nwg <= 128) ok = tiny_select8<TINY_POOL, 2, 16>(cs, ci, nwg, tiny_lb<2>(pb, m.npre, lane), sv_s, sv_i, lane, nullptr, cand + q * TINY_POOL, bo, bo2);
            else ok = tiny_select8<TINY_POOL, 4, 32>(cs, ci, nwg, tiny_lb<4>(pb, m.npre, lane), sv_s, sv_i, lane, nullptr, cand + q * TINY_POOL, bo, bo2);
        }
        if (!ok) merge_select_body<TINY_POOL>(m, cand, q, lane);
        if (F32) {
            // |q - bf16 q|^2 (margin) and, for the inner-product metric, |q|^2 -- there it only enters the margin's error bound, so
            // a parallel fp64 sum with a little slack serves (L2 needs the canonical |q|^2: the re-score loop below computes it)
            const float* y = qf + (size_t)q * a.plane;
            double aq = 0.0, ar = 0.0;
            for (int ch = lane; ch < a.plane / 4; ch += 64) {
                const u32x4 yv = *reinterpret_cast<const u32x4*>(y + ch * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ye = __uint_as_float(yv[e]);
                    const double r = (double)ye - (double)bf16_bits_to_f32(f32_to_bf16_rne(ye));
                    aq += (double)ye * (double)ye;
                    ar += r * r;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                aq += __shfl_xor(aq, off);
                ar += __shfl_xor(ar, off);
            }
            if (lane == 0) {
                qerr2_s[q] = ar * (1.0 + 1e-12);
                if (!L2) qq_s[q] = aq * (1.0 + 1e-12);
            }
        } else {
            const uint16_t* y = qs + (size_t)q * a.ld;
            double acc = 0.0;
            TinyCert ce = {0u, 0xffffffffu};
            for (int ch = lane; ch < a.ld / 8; ch += 64) {
                const u32x4 yv = *reinterpret_cast<const u32x4*>(y + ch * 8);
                tiny_dot_chunk(yv, yv, acc, ce);
            }
            tiny_fold<DPP_XOR1>(acc, ce);
            tiny_fold<DPP_XOR2>(acc, ce);
            tiny_fold<DPP_HALF_MIRROR>(acc, ce);
            tiny_fold<DPP_ROW_MIRROR>(acc, ce);
            double tot = 0.0;
            TinyCert ca = {0u, 0xffffffffu};
#pragma unroll
            for (int rw = 0; rw < 4; ++rw) { // the four 16-lane rows
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)__double2loint(acc), 16 * rw);
                const unsigned hi = __builtin_amdgcn_readlane((unsigned)__double2hiint(acc), 16 * rw);
                tot += __hiloint2double((int)hi, (int)lo);
                const unsigned omx = __builtin_amdgcn_readlane(ce.mx, 16 * rw), omn = __builtin_amdgcn_readlane(ce.mn, 16 * rw);
                ca.mx = ca.mx > omx ? ca.mx : omx;
                ca.mn = ca.mn < omn ? ca.mn : omn;
            }
            if (!tiny_cert_ok(ca, denorm_ok)) tot = tiny_dot_sequential(y, y, a.ld); // (every lane: the same value)
            if (lane == 0) qq_s[q] = tot;
        }
    }
    __syncthreads();
    TINY_STAMP(8);
    // exact re-score: 8 lanes per (query, candidate) pair, 64 pairs per round
    if (F32) { // on the fp32 rows (tiny_seq_dot_f32); L2 also needs the canonical |q|^2: nq more "pairs" (q, q)
        const int npair = a.nq * TINY_POOL + (L2 ? a.nq : 0);
#pragma unroll 1
        for (int base = 0; base < npair; base += TINY_THREADS / 8) {
            const int pair = base + (tid >> 3);
            const int sub = tid & 7;
            const bool inp = pair < npair;
            const bool isqq = inp && pair >= a.nq * TINY_POOL;
            const int qi = !inp ? 0 : isqq ? pair - a.nq * TINY_POOL : pair / TINY_POOL;
            const int ci = !inp ? IDX_NONE : isqq ? 0 : cand[pair];
            const float* y = qf + (size_t)qi * a.plane;
            const float* x = isqq ? y : a.rows_f32 + (size_t)(ci != IDX_NONE ? ci : 0) * a.plane;
            const double acc = tiny_seq_dot_f32(x, y, a.plane / 4, sub);
            if (inp && sub == 0) {
                if (isqq) qq_s[qi] = acc;
                else if (ci != IDX_NONE) dots[pair] = acc;
            }
        }
    } else
    for (int base = 0; base < a.nq * TINY_POOL; base += TINY_THREADS / 8) {
        const int pair = base + (tid >> 3);
        const int sub = tid & 7;
        const bool inp = pair < a.nq * TINY_POOL;
        const int ci = inp ? cand[pair] : IDX_NONE;
        double acc = 0.0;
        TinyCert ce = {0u, 0xffffffffu};
        const uint16_t* x = reinterpret_cast<const uint16_t*>(m.docs) + (size_t)(ci != IDX_NONE ? ci : 0) * a.ld;
        const uint16_t* y = qs + (size_t)(inp ? pair / TINY_POOL : 0) * a.ld;
        {
            // 16-byte chunks sub, sub + 8, ...: groups of 8 per lane (64 chunks = 512 columns per group), the next group's
            // loads in flight while this one is summed; a LOOP over the groups (code size, see the staging comment)
            const int nchunk = a.ld / 8; // 16 .. 128, a multiple of 16
            u32x4 xr[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (8 * t < nchunk) xr[t] = *reinterpret_cast<const u32x4*>(x + (sub + 8 * t) * 8);
#pragma unroll 1
            for (int h0 = 0; h0 < nchunk; h0 += 64) {
                u32x4 xn[8];
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (h0 + 64 + 8 * t < nchunk) xn[t] = *reinterpret_cast<const u32x4*>(x + (h0 + 64 + sub + 8 * t) * 8);
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (h0 + 8 * t < nchunk) {
                        const u32x4 yv = *reinterpret_cast<const u32x4*>(y + (h0 + sub + 8 * t) * 8);
                        tiny_dot_chunk(xr[t], yv, acc, ce);
                    }
#pragma unroll
                for (int t = 0; t < 8; ++t) xr[t] = xn[t];
            }
        }
        tiny_fold<DPP_XOR1>(acc, ce);
        tiny_fold<DPP_XOR2>(acc, ce);
        tiny_fold<DPP_HALF_MIRROR>(acc, ce);
        if (ci != IDX_NONE && sub == 0) {
            if (!tiny_cert_ok(ce, denorm_ok)) acc = tiny_dot_sequential(x, y, a.ld);
            dots[pair] = acc;
        }
    }
    __syncthreads();
    TINY_STAMP(9);
    // rank, margin check, write: one lane per pair, 8 queries per wave; the k results go to LDS [q][m.k]
    if (wave < 2) {
        const int64_t q = wave * 8 + lane / TINY_POOL;
        const int slot = lane % TINY_POOL;
        const bool inq = q < a.nq;
        const int ci = inq ? cand[q * TINY_POOL + slot] : IDX_NONE;
        const bool valid = ci != IDX_NONE;
        const double dot = valid ? dots[q * TINY_POOL + slot] : 0.0;
        const double qq = valid ? qq_s[q] : 0.0;
        rank_flag_write<TINY_POOL, L2>(m, q, slot, inq, ci, valid, dot, qq, lane);
    }
    __syncthreads();
    TINY_STAMP(10);
    // hand-off to the stream-ordered exact pass (resolve_kernels.hpp): what compact_flags_kernel does for the general path, and
    // -- only when a query was flagged -- the staged queries in global memory, where exact_filter_kernel reads them
    if (a.res_cnt != nullptr) {
        if (tid == 0) {
            int n = 0;
            for (int q = 0; q < a.nq; ++q)
                if (flag_s[q]) a.res_ids[n++] = q;
            *a.res_cnt = n;
            *a.res_unres = 0u;
            *nfl_s = n;
        }
        if (tid < 16) a.res_hit_n[tid] = 0;
        __syncthreads();
        if (*nfl_s > 0) { // (uniform)
            if (F32) {
                u32x4* dst = reinterpret_cast<u32x4*>(a.q_out);
                const u32x4* src = reinterpret_cast<const u32x4*>(qf);
                for (int t = tid; t < a.nq * a.plane / 4; t += TINY_THREADS) dst[t] = src[t];
            } else {
                u32x4* dst = reinterpret_cast<u32x4*>(a.q_out);
                const u32x4* src = reinterpret_cast<const u32x4*>(qs);
                for (int t = tid; t < a.nq * a.ld / 8; t += TINY_THREADS) dst[t] = src[t];
            }
        }
    }
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
    TINY_STAMP(11);
#ifdef MIPS_EXPERIMENTAL
    if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 16 + 13] = clock64();
#endif
}

} // namespace mips
