// Kernels either side of the fused scan: dtype conversion into the index layout, the synthetic
// generators of SURVEY.md 8d, the split merge + exact re-score, and the cross-shard merge.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.hpp"

namespace mips {

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0; // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// ---- OCP fp8 e4m3 ("e4m3fn": bias 7, 3 mantissa bits, max 448, 0x7f/0xff = NaN, no infinities).
// Written with integer arithmetic so that oracle/synth.py::round_to_e4m3 is the same algorithm bit for
// bit: round to nearest even, saturate to +-448, NaN stays NaN.
__host__ __device__ __forceinline__ uint8_t f32_to_e4m3(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80u);
    const uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint8_t)(sign | 0x7fu);  // NaN
    if (a >= 0x43e00000u) return (uint8_t)(sign | 0x7eu); // |x| >= 448 (and inf): saturate
    const int e = (int)(a >> 23) - 127;
    if (e < -6) { // subnormal range of e4m3: multiples of 2^-9; 8 * 2^-9 is the smallest normal (code 0x08)
        float ax;
        memcpy(&ax, &a, 4);
        const int qv = (int)rintf(ax * 512.0f); // round-to-nearest-even in the default rounding mode
        return (uint8_t)(sign | (uint8_t)qv);
    }
    uint32_t m3 = (a >> 20) & 7u;
    const uint32_t rem = a & 0xfffffu;
    int ee = e + 7;
    if (rem > 0x80000u || (rem == 0x80000u && (m3 & 1u))) {
        if (++m3 == 8u) {
            m3 = 0;
            ++ee;
        }
    }
    return (uint8_t)(sign | (uint8_t)((ee << 3) | (int)m3)); // < 0x7f: values that would round to 480+ were saturated above
}
__host__ __device__ __forceinline__ float e4m3_to_f32(uint32_t b) {
    const uint32_t e = (b >> 3) & 15u, m = b & 7u;
    float v;
    if (e == 15u && m == 7u) {
        const uint32_t nanbits = 0x7fc00000u;
        memcpy(&v, &nanbits, 4);
    } else if (e == 0u) {
        v = (float)m * (1.0f / 512.0f);
    } else {
        const uint32_t bits = ((e + 120u) << 23) | (m << 20);
        memcpy(&v, &bits, 4);
    }
    return (b & 0x80u) ? -v : v;
}

// element traits of the two index storage types: ELEMS per 16-byte chunk and the exact up-cast
struct ElemBF16 {
    typedef uint16_t type;
    static constexpr int PER16 = 8;
    __device__ static __forceinline__ float get(const u32x4& v, int e) {
        return bf16_bits_to_f32((v[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    }
};
struct ElemF32 {
    typedef float type;
    static constexpr int PER16 = 4;
    __device__ static __forceinline__ float get(const u32x4& v, int e) { return __uint_as_float(v[e]); }
};
struct ElemF8 {
    typedef uint8_t type;
    static constexpr int PER16 = 16;
    // v_cvt_f32_fp8 decodes OCP e4m3 on gfx950 (exact: every e4m3 value is an fp32 value); the byte selector is an
    // immediate, hence the switch (it folds once the callers' loops are unrolled).  One instruction per element
    // instead of ~10: the exact re-score of an fp8 index went from 138 to ~25 us per 4096 queries.
    __device__ static __forceinline__ float get(const u32x4& v, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int w = (int)v[e >> 2];
        switch (e & 3) {
        case 0: return __builtin_amdgcn_cvt_f32_fp8(w, 0);
        case 1: return __builtin_amdgcn_cvt_f32_fp8(w, 1);
        case 2: return __builtin_amdgcn_cvt_f32_fp8(w, 2);
        default: return __builtin_amdgcn_cvt_f32_fp8(w, 3);
        }
#else
        return e4m3_to_f32((v[e >> 2] >> ((e & 3) * 8)) & 0xffu);
#endif
    }
};

// ------------------------------------------------------------------ conversion into [rows][ld] bf16
// one thread per 8 output elements; columns >= d are written as zero.
// Query staging folds two memsets into this launch: rows n .. n_out - 1 are written as zero (the padding up to the
// query-tile multiple) and `zero_words` 32-bit words at `zero` are cleared (the shared insert bounds of the scan).
__device__ __forceinline__ void zero_words_grid_stride(uint32_t* zero, int64_t zero_words) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < zero_words; t += (int64_t)gridDim.x * blockDim.x) zero[t] = 0u;
}

template <typename SRC>
__global__ void convert_rows_kernel(const SRC* __restrict__ src, int64_t n, int d, uint16_t* dst, int ld, int64_t n_out = 0,
                                    uint32_t* zero = nullptr, int64_t zero_words = 0) {
    const int chunks = ld / 8;
    const int64_t total = (n_out > n ? n_out : n) * chunks;
    zero_words_grid_stride(zero, zero_words);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / chunks;
        const int c0 = (int)(t % chunks) * 8;
        uint16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            if (c < d && row < n) {
                if constexpr (sizeof(SRC) == 4)
                    v[e] = f32_to_bf16_rne(((const float*)src)[row * d + c]);
                else
                    v[e] = ((const uint16_t*)src)[row * d + c];
            } else {
                v[e] = 0;
            }
        }
        u32x4 o;
        o[0] = v[0] | ((uint32_t)v[1] << 16);
        o[1] = v[2] | ((uint32_t)v[3] << 16);
        o[2] = v[4] | ((uint32_t)v[5] << 16);
        o[3] = v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<u32x4*>(dst + row * ld + c0) = o;
    }
}

// conversion into [rows][ld] e4m3 bytes; SRC = float, uint16_t (bf16 bits) or uint8_t (raw e4m3)
template <typename SRC>
__global__ void convert_rows_f8_kernel(const SRC* __restrict__ src, int64_t n, int d, uint8_t* dst, int ld, int64_t n_out = 0,
                                       uint32_t* zero = nullptr, int64_t zero_words = 0) {
    const int chunks = ld / 16;
    const int64_t total = (n_out > n ? n_out : n) * chunks;
    zero_words_grid_stride(zero, zero_words);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / chunks;
        const int c0 = (int)(t % chunks) * 16;
        u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = c0 + e;
            uint32_t b = 0;
            if (c < d && row < n) {
                if constexpr (sizeof(SRC) == 4) b = f32_to_e4m3(((const float*)src)[row * d + c]);
                else if constexpr (sizeof(SRC) == 2) b = f32_to_e4m3(bf16_bits_to_f32(((const uint16_t*)src)[row * d + c]));
                else b = ((const uint8_t*)src)[row * d + c];
            }
            o[e >> 2] |= b << ((e & 3) * 8);
        }
        *reinterpret_cast<u32x4*>(dst + row * ld + c0) = o;
    }
}

// fp32-exact mode: split fp32 (or bf16) rows [n][d] (pitch src_ld) into bf16 planes dst [n][2 * plane] =
// [hi | lo], hi = bf16(x), lo = bf16(x - hi) (x - hi is exact in fp32), columns >= d zero; optionally keep
// the fp32 originals in keep [n][plane] for the exact re-score.
template <typename SRC>
__global__ void split_rows_kernel(const SRC* __restrict__ src, int64_t n, int d, int src_ld, uint16_t* dst, int plane,
                                  float* keep) {
    const int chunks = plane / 8;
    const int64_t total = n * chunks;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / chunks;
        const int c0 = (int)(t % chunks) * 8;
        uint16_t hi[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float x = 0.f;
            if (c < d) {
                if constexpr (sizeof(SRC) == 4) x = ((const float*)src)[row * src_ld + c];
                else x = bf16_bits_to_f32(((const uint16_t*)src)[row * src_ld + c]);
            }
            hi[e] = f32_to_bf16_rne(x);
            const float r = x - bf16_bits_to_f32(hi[e]);
            lo[e] = (x == x && fabsf(x) != INFINITY) ? f32_to_bf16_rne(r) : (uint16_t)0; // inf - inf / NaN: keep lo = 0
            if (keep) keep[row * plane + c] = x;
        }
        u32x4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            oh[e] = hi[2 * e] | ((uint32_t)hi[2 * e + 1] << 16);
            ol[e] = lo[2 * e] | ((uint32_t)lo[2 * e + 1] << 16);
        }
        *reinterpret_cast<u32x4*>(dst + row * 2 * plane + c0) = oh;
        *reinterpret_cast<u32x4*>(dst + row * 2 * plane + plane + c0) = ol;
    }
}

__global__ void zero_rows_kernel(uint16_t* dst, int64_t nrows, int ld) {
    const int64_t total = nrows * (ld / 8);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        u32x4 z = {0u, 0u, 0u, 0u};
        reinterpret_cast<u32x4*>(dst)[t] = z;
    }
}

// ------------------------------------------------------------------ synthetic generators
// Bit-identical to oracle/synth.py (a GPU test checks it).
__host__ __device__ __forceinline__ uint64_t synth_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t synth_row_key(uint64_t seed, uint64_t row) {
    return synth_mix(seed ^ (row * 0xD6E8FEB86659FD93ull));
}
__device__ __forceinline__ float synth_value(uint64_t key, uint64_t col, int kind) {
    if (kind == 1) { // GAUSS: centred sum of eight 16-bit uniforms, scaled, rounded to bf16
        const uint64_t h1 = synth_mix(key + 2 * col), h2 = synth_mix(key + 2 * col + 1);
        int s = 0;
#pragma unroll
        for (int sh = 0; sh < 64; sh += 16) s += (int)((h1 >> sh) & 0xffff) + (int)((h2 >> sh) & 0xffff);
        s -= 4 * 65535;
        const float x = (float)s * (float)(1.0 / (65536.0 * 0.816496580927726));
        return bf16_bits_to_f32(f32_to_bf16_rne(x));
    }
    const uint64_t hsh = synth_mix(key + col) >> 40;
    if (kind == 2) return (float)((int)(hsh % 17) - 8) / 8.0f; // LATTICE_FP8
    return (float)((int)(hsh % 255) - 127) / 64.0f;            // LATTICE
}

// out: bf16 [n][ld] (as_f32 == 0), f32 [n][ld] (1) or e4m3 bytes [n][ld] (2); columns >= d are zero
__global__ void synth_fill_kernel(void* out, int64_t n, int d, int ld, int64_t row0, uint64_t seed, int kind,
                                  int as_f32) {
    const int chunks = ld / 8;
    const int64_t total = n * chunks;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / chunks;
        const int c0 = (int)(t % chunks) * 8;
        const uint64_t key = synth_row_key(seed, (uint64_t)(row0 + row));
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (c0 + e < d) ? synth_value(key, (uint64_t)(c0 + e), kind) : 0.f;
        if (as_f32 == 2) { // e4m3 bytes
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo |= (uint32_t)f32_to_e4m3(v[e]) << (8 * e);
                hi |= (uint32_t)f32_to_e4m3(v[4 + e]) << (8 * e);
            }
            uint32_t* o = reinterpret_cast<uint32_t*>((uint8_t*)out + row * ld + c0);
            o[0] = lo;
            o[1] = hi;
        } else if (as_f32) {
            float* o = (float*)out + row * ld + c0;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[e];
        } else {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = f32_to_bf16_rne(v[2 * e]) | ((uint32_t)f32_to_bf16_rne(v[2 * e + 1]) << 16);
            *reinterpret_cast<u32x4*>((uint16_t*)out + row * ld + c0) = o;
        }
    }
}

// ------------------------------------------------------------------ phi = max row |x|^2 (fp64, sequential)
template <typename EL>
__global__ void row_sumsq_max_kernel(const typename EL::type* rows, int64_t n, int ld, unsigned long long* out_bits) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double acc = 0.0; // rows past the end contribute 0 (the maximum of non-negative values is unaffected)
    if (r < n) {
        const typename EL::type* x = rows + r * ld;
        for (int c = 0; c < ld; c += EL::PER16) { // sequential in the column index: the oracle's order
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + c);
#pragma unroll
            for (int e = 0; e < EL::PER16; ++e) {
                const double a = (double)EL::get(v, e);
                acc += a * a;
            }
        }
    }
    // one atomic per wave, not per row (a million atomics on one word take ~12 ms): the maximum is exact in any
    // order; non-negative doubles order like their bit patterns
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = fmax(acc, __shfl_xor(acc, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(acc));
}

// max_i |x_i - bf16(x_i)|^2 over fp32 rows (two-stage fp32-exact search: how far the bf16 scan's operand is from the row)
__global__ void row_resid_sumsq_max_kernel(const float* rows, int64_t n, int ld, unsigned long long* out_bits) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (r < n) {
        const float* x = rows + r * ld;
        for (int c = 0; c < ld; c += 4) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xe = __uint_as_float(v[e]);
                const double a = (double)xe - (double)bf16_bits_to_f32(f32_to_bf16_rne(xe));
                acc += a * a;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc = fmax(acc, __shfl_xor(acc, off));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(acc));
}

// |q - bf16(q)|^2 per staged fp32 query row (one wave per row)
__global__ __launch_bounds__(256) void query_resid_kernel(const float* q, int64_t nq, int ld, double* out) {
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= nq) return;
    double acc = 0.0;
    for (int c = lane; c < ld; c += 64) {
        const float xe = q[row * ld + c];
        const double a = (double)xe - (double)bf16_bits_to_f32(f32_to_bf16_rne(xe));
        acc += a * a;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[row] = acc * (1.0 + 1e-12); // (a bound: the summation order is free)
}

// ------------------------------------------------------------------ row L2 normalisation (fp32, in place)
// Replaces faiss.normalize_L2 as used by Mips.l2_normalization (sotasum/mips.py:521-525; faiss
// fvec_renorm_L2: nr = sum x^2 in fp32, x *= 1/sqrtf(nr) when nr > 0).  One wave per row.
__global__ __launch_bounds__(256) void l2_normalize_kernel(float* x, int64_t n, int d) {
    const int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n) return;
    float* r = x + row * d;
    float nr = 0.f;
    for (int c = lane; c < d; c += 64) nr += r[c] * r[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nr += __shfl_xor(nr, off);
    if (nr > 0.f) {
        const float inv = 1.0f / sqrtf(nr);
        for (int c = lane; c < d; c += 64) r[c] *= inv;
    }
}

// max over rows of |x|^2 for fp32 rows (fp64 accumulation); max_norm of mips.py:298-304 is its sqrt
__global__ __launch_bounds__(256) void f32_rows_max_sumsq_kernel(const float* x, int64_t n, int d,
                                                                 unsigned long long* out_bits) {
    // a wave takes rows blockIdx.x * 4 + wave, + 4 gridDim.x, ... and keeps a running maximum: one atomic per wave
    const int lane = threadIdx.x & 63;
    double best = 0.0;
    for (int64_t row = blockIdx.x * 4ll + (threadIdx.x >> 6); row < n; row += 4ll * gridDim.x) {
        const float* r = x + row * d;
        double acc = 0.0;
        for (int c = lane; c < d; c += 64) acc += (double)r[c] * (double)r[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        best = fmax(best, acc);
    }
    if (lane == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(best));
}

// ------------------------------------------------------------------ split merge + exact re-score
__device__ __forceinline__ bool ranks_before(float s1, int i1, float s2, int i2) {
    return s1 > s2 || (s1 == s2 && i1 < i2);
}

template <int KL>
__device__ __forceinline__ void list_insert_full(float (&ls)[KL], int (&li)[KL], float s, int id) {
    ls[KL - 1] = s;
    li[KL - 1] = id;
#pragma unroll
    for (int j = KL - 1; j > 0; --j) {
        const float lo = ls[j], hi = ls[j - 1];
        const int ilo = li[j], ihi = li[j - 1];
        const bool sw = ranks_before(lo, ilo, hi, ihi);
        ls[j - 1] = sw ? lo : hi;
        ls[j] = sw ? hi : lo;
        li[j - 1] = sw ? ilo : ihi;
        li[j] = sw ? ihi : ilo;
    }
}

struct MergeArgs {
    const float* part_s; // [nq_pad][nsplit][2][KL]
    const int* part_i;
    int ncand;            // nsplit * 2 * KL
    const void* docs; // [cap][ld] index storage (bf16 bits or e4m3 bytes)
    const void* qbuf; // [nq_pad][ld] staged queries, same element type
    int ld;
    int k;
    int metric;
    double phi;
    int64_t idx_offset;
    float* out_s; // [nq][k]
    int64_t* out_i;
    int64_t* out_packed; // optional [nq][k][2] = {float bits (zero-extended), global id}: the all-gather payload
    const unsigned* err; // the scan kernel's error word of this call (nullptr: the generic kernel has no bounded spin)
    unsigned* sticky;    // host-visible (pinned, mapped) word of the index: set to 1 when `err` was set
    // ---- margin check: is the candidate pool provably wide enough?
    int ll;               // entries per running list of the scan kernel (6 for the 16x16 kernels, K' otherwise)
    const float* pre_bnd; // optional [nq][npre]: bounds on what an EARLIER selection level excluded (tiny_search.hpp)
    int npre;
    float* bnd;           // [nq] out of merge_select: no document OUTSIDE the pool has an MFMA score above this
    unsigned char* flag;  // [nq] out of the re-score: 1 = the k-th exact score is within the MFMA error of bnd
    unsigned* nflag;      // device counter of flagged queries of this call (nullptr: margin check off)
    const double* xmax2;  // device scalar: max_i |x_i|^2 over the stored rows
    double err_c;         // MFMA score error <= err_c * |q| * |x|   (d * 2^-23: fp32 accumulation of exact products)
    // two-stage fp32-exact search (mips_hip.hip, "f32_fast"): the scan saw bf16(x) . bf16(q) only; the exact score differs
    // from that by at most |x - bf16 x| |q| + |bf16 x| |q - bf16 q|  (Cauchy-Schwarz), added to the margin
    const double* dres2 = nullptr; // device scalar: max_i |x_i - bf16(x_i)|^2 over the stored rows
    const double* qerr2 = nullptr; // [nq]: |q - bf16(q)|^2
    const int* nq_dev = nullptr;   // stream-ordered re-scan: the query count lives on the device (launches sized for the maximum)
    float* keyk = nullptr;         // [nq] out: canonical key of the k-th result (+inf: fewer than k results) -- resolve_kernels.hpp
    double* qq_out = nullptr;      // [nq] out: |q|^2 as the re-score summed it
    // a second bound with an error constant of its own (tiny_search.hpp, fp32-exact index): bnd2[q] = best REFINED score outside
    // the pool (fp32 operands, fp64 sum, float32 storage: within err_c2 |q| max|x| of the canonical score)
    const float* bnd2 = nullptr;
    double err_c2 = 0.0;
};

// What a search whose scan kernel gave up (split-barrier spin bound, scan_kernel_v3.hpp) returns instead of
// results: every slot idx = IDX_POISON, score = NaN.  The cross-shard merges propagate it, so a timed-out shard
// can never silently drop out of a global top-k (include/mips_hip.h, MIPS_IDX_POISON).
constexpr int64_t IDX_POISON = -2;
__device__ __forceinline__ float poison_score() { return __uint_as_float(0x7fc00000u); }

// Split merge + exact re-score, two launches:
//
// merge_select_kernel  -- one wave per query:
//  1. every lane folds its strided share of the query's candidate lists into a private sorted K-list
//     (full comparator: candidates do not arrive in index order here);
//  2. KL rounds of wave-wide arg-best pop the K best candidates by MFMA score; lane r keeps the r-th and
//     writes its row id to cand[q][r].
// rescore_rank_kernel  -- one THREAD per (query, candidate), 64 / KL queries per wave (every lane busy):
//  3. exact re-score: sequential fp64 sum over k of q[k]*x[k] on the stored values (each product is exact
//     in fp64, so the result does not depend on FMA contraction), cast to float -- the canonical score of
//     include/mips_hip.h; only the loads run ahead (a ring of 16-byte chunks), the sum order is the oracle's;
//  4. rank the KL candidates of a query inside their lane group by (canonical score desc, idx asc)
//     [L2: distance asc] and write the top k (or the packed all-gather payload).
template <int KL>
__device__ __forceinline__ void merge_select_body(const MergeArgs& p, int* cand, int q, int lane, float* cand_s = nullptr,
                                                  float* bnd_out = nullptr) {
    const float* ps = p.part_s + (size_t)q * p.ncand;
    const int* pi = p.part_i + (size_t)q * p.ncand;

    float ls[KL];
    int li[KL];
#pragma unroll
    for (int i = 0; i < KL; ++i) {
        ls[i] = -INFINITY;
        li[i] = IDX_NONE;
    }
    // lb: the best score a document DROPPED from a full running list can have had = that list's last entry
    float lb = -INFINITY;
    // 8 candidates per lane are loaded before any of them is looked at: the insert test is data dependent, and left
    // alone the loop pays one memory round trip per candidate (lists freshly written by other workgroups: ~0.7 us each)
    for (int c0 = lane; c0 < p.ncand; c0 += 8 * 64) {
        float sv[8];
        int iv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + 64 * u;
            sv[u] = c < p.ncand ? ps[c] : -INFINITY;
            iv[u] = c < p.ncand ? pi[c] : IDX_NONE;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + 64 * u;
            const float s = sv[u];
            const int id = iv[u];
            if (ranks_before(s, id, ls[KL - 1], li[KL - 1])) list_insert_full<KL>(ls, li, s, id);
            if (id != IDX_NONE && (c % p.ll) == p.ll - 1) lb = fmaxf(lb, s);
        }
    }
    int ci = IDX_NONE;
    float last_pop = -INFINITY;
    int pops = 0;
    for (int r = 0; r < KL; ++r) {
        const float hs = ls[0];
        const int hi = li[0];
        float bs = hs;
        int bi = hi;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            if (ranks_before(os, oi, bs, bi)) {
                bs = os;
                bi = oi;
            }
        }
        if (hs == bs && hi == bi) { // the owner pops its head (document ids are unique across lists)
#pragma unroll
            for (int i = 0; i + 1 < KL; ++i) {
                ls[i] = ls[i + 1];
                li[i] = li[i + 1];
            }
            ls[KL - 1] = -INFINITY;
            li[KL - 1] = IDX_NONE;
        }
        if (lane == r) {
            ci = bi;
            if (cand_s) cand_s[(size_t)q * KL + r] = bs;
        }
        if (bi != IDX_NONE) {
            last_pop = bs;
            ++pops;
        }
    }
    if (lane < KL) cand[(size_t)q * KL + lane] = ci;
    if (p.pre_bnd != nullptr)
        for (int c = lane; c < p.npre; c += 64) lb = fmaxf(lb, p.pre_bnd[(size_t)q * p.npre + c]);
    if (bnd_out != nullptr) { // an intermediate selection level: hand its bound to the next one
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lb = fmaxf(lb, __shfl_xor(lb, off));
        if (lane == 0) *bnd_out = fmaxf(pops == KL ? last_pop : -INFINITY, lb);
    } else if (p.bnd != nullptr) {
        // Every document outside the pool has an MFMA score <= bnd: list entries that were not popped rank behind
        // the pool's last member; documents rejected by a shared insert bound g scored below g <= the K'-th best
        // list entry (g is vouched for by K' list entries); documents dropped from a full list scored <= its last
        // entry.  A pool that is not full (fewer than K' entries anywhere) excluded nothing by rank.
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lb = fmaxf(lb, __shfl_xor(lb, off));
        if (lane == 0) p.bnd[q] = fmaxf(pops == KL ? last_pop : -INFINITY, lb);
    }
}

template <int KL>
__global__ __launch_bounds__(64) void merge_select_kernel(MergeArgs p, int* cand) {
    if (p.nq_dev != nullptr && (int)blockIdx.x >= *p.nq_dev) return;
    merge_select_body<KL>(p, cand, (int)blockIdx.x, (int)threadIdx.x);
}

// Steps 4 + margin check of the re-score, shared by rescore_rank_body and tiny_search.hpp: lane (q, slot) holds candidate ci
// of query q with its canonical dot product and |q|^2 (fp64); rank inside the query's KL-lane group, flag, write.
template <int KL, bool L2>
__device__ __forceinline__ void rank_flag_write(const MergeArgs& p, int64_t q, int slot, bool inq, int ci, bool valid, double dot, double qq,
                                                int lane) {
    float outv, key;
    if (L2) { // L2 on phi-augmented vectors: |q|^2 + phi - 2 q.x, smaller is better
        outv = (float)(qq + p.phi - 2.0 * dot);
        key = -outv;
    } else {
        outv = (float)dot;
        key = outv;
    }
    if (!valid) key = -INFINITY;

    // rank inside the query's KL-lane group
    const int gbase = lane - slot;
    int rank = 0, nvalid = 0;
#pragma unroll
    for (int jj = 0; jj < KL; ++jj) {
        const float ok = __shfl(key, gbase + jj);
        const int oi = __shfl(ci, gbase + jj);
        if (oi != IDX_NONE) {
            ++nvalid;
            if (ranks_before(ok, oi, key, ci)) ++rank;
        }
    }
    if (p.nflag != nullptr) {
        // Margin check.  tk = exact inner product of the k-th result; a document outside the pool has an exact
        // inner product <= bnd + e, e = err_c |q| max|x|.  If that can reach tk the pool was not provably wide
        // enough: flag the query (the host re-scans flagged queries with the widest lists, mips_hip.hip).
        double tk = 0.0, qn = 0.0;
        bool have = false;
#pragma unroll
        for (int jj = 0; jj < KL; ++jj) {
            const int orank = __shfl(rank, gbase + jj);
            const int oi = __shfl(ci, gbase + jj);
            const double od = __shfl(dot, gbase + jj);
            const double oq = __shfl(qq, gbase + jj);
            if (oi != IDX_NONE) {
                qn = oq;
                if (orank == p.k - 1) {
                    tk = od;
                    have = true;
                }
            }
        }
        if (inq && slot == 0 && p.keyk != nullptr) {
            p.keyk[q] = have ? (L2 ? -(float)(qn + p.phi - 2.0 * tk) : (float)tk) : INFINITY;
            p.qq_out[q] = qn;
        }
        if (inq && slot == 0) {
            const float b = p.bnd[q];
            bool fl = false;
            if (have && b > -INFINITY) {
                double e = p.err_c * sqrt(qn) * sqrt(*p.xmax2);
                if (p.qerr2 != nullptr) {
                    const double dr = sqrt(*p.dres2);
                    e += dr * sqrt(qn) + (sqrt(*p.xmax2) + dr) * sqrt(p.qerr2[q]);
                }
                fl = !((double)b + e < tk); // also true for NaN: never certify what cannot be compared
            }
            if (p.bnd2 != nullptr && have) {
                const float b2 = p.bnd2[q];
                if (b2 > -INFINITY) fl = fl || !((double)b2 + p.err_c2 * sqrt(qn) * sqrt(*p.xmax2) < tk);
            }
            p.flag[q] = fl ? 1 : 0;
            if (fl) atomicAdd(p.nflag, 1u);
        }
    }
    if (!inq) return;
    if (valid && rank < p.k) {
        const size_t o = (size_t)q * p.k + rank;
        if (p.out_packed) {
            p.out_packed[2 * o] = (int64_t)__float_as_uint(outv);
            p.out_packed[2 * o + 1] = (int64_t)ci + p.idx_offset;
        } else {
            p.out_s[o] = outv;
            p.out_i[o] = (int64_t)ci + p.idx_offset;
        }
    }
    if (slot < p.k && slot >= nvalid) {
        const size_t o = (size_t)q * p.k + slot;
        const float pad = L2 ? INFINITY : -INFINITY;
        if (p.out_packed) {
            p.out_packed[2 * o] = (int64_t)__float_as_uint(pad);
            p.out_packed[2 * o + 1] = -1;
        } else {
            p.out_s[o] = pad;
            p.out_i[o] = -1;
        }
    }
}

// ELQ: element type of the staged queries (= EL except for the e4m3-documents / bf16-queries index)
template <int KL, typename EL, bool L2, typename ELQ = EL>
__device__ __forceinline__ void rescore_rank_body(const MergeArgs& p, const int* cand, int64_t nq, int64_t wave_index, int lane) {
    constexpr int QPW = 64 / KL; // queries per wave
    const int64_t q = wave_index * QPW + lane / KL;
    const int slot = lane % KL;
    const bool inq = q < nq && lane < QPW * KL; // K' = 10: lanes 60..63 belong to no query
    if (p.err != nullptr && *p.err != 0u) { // the scan gave up on its barrier: nothing below can be trusted
        if (wave_index == 0 && lane == 0) __hip_atomic_store(p.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (inq && slot == 0 && p.flag != nullptr) p.flag[q] = 0; // (nothing to re-scan: a stream-ordered re-scan must not touch the poison)
        if (inq && slot < p.k) {
            const size_t o = (size_t)q * p.k + slot;
            if (p.out_packed) {
                p.out_packed[2 * o] = (int64_t)__float_as_uint(poison_score());
                p.out_packed[2 * o + 1] = IDX_POISON;
            } else {
                p.out_s[o] = poison_score();
                p.out_i[o] = IDX_POISON;
            }
        }
        return;
    }
    const int ci = inq ? cand[(size_t)q * KL + slot] : IDX_NONE;
    const bool valid = ci != IDX_NONE;
    double dot = 0.0, qq = 0.0;
    if (valid && EL::PER16 != ELQ::PER16) {
        // e4m3 rows x bf16 queries: one 16-byte chunk of a row (16 elements) meets two of the query (8 each); same sequential
        // order of the sum as below
        static_assert(EL::PER16 == ELQ::PER16 || EL::PER16 == 2 * ELQ::PER16, "queries at most twice as wide as the rows");
        const typename EL::type* x = reinterpret_cast<const typename EL::type*>(p.docs) + (size_t)ci * p.ld;
        const typename ELQ::type* y = reinterpret_cast<const typename ELQ::type*>(p.qbuf) + (size_t)q * p.ld;
        constexpr int PF = 12;
        const int nchunk = p.ld / EL::PER16;
        u32x4 xr[PF], ya[PF], yb[PF];
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            const int c = t < nchunk ? t : nchunk - 1;
            xr[t] = *reinterpret_cast<const u32x4*>(x + c * EL::PER16);
            ya[t] = *reinterpret_cast<const u32x4*>(y + c * EL::PER16);
            yb[t] = *reinterpret_cast<const u32x4*>(y + c * EL::PER16 + ELQ::PER16);
        }
        for (int c0 = 0; c0 < nchunk; c0 += PF) {
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const u32x4 xv = xr[t], yav = ya[t], ybv = yb[t];
                const int nx = c0 + PF + t < nchunk ? c0 + PF + t : nchunk - 1;
                xr[t] = *reinterpret_cast<const u32x4*>(x + nx * EL::PER16);
                ya[t] = *reinterpret_cast<const u32x4*>(y + nx * EL::PER16);
                yb[t] = *reinterpret_cast<const u32x4*>(y + nx * EL::PER16 + ELQ::PER16);
                if (c0 + t < nchunk) {
#pragma unroll
                    for (int e = 0; e < EL::PER16; ++e) {
                        const double xe = (double)EL::get(xv, e);
                        const double ye = (double)(e < ELQ::PER16 ? ELQ::get(yav, e % ELQ::PER16) : ELQ::get(ybv, e % ELQ::PER16));
                        dot += xe * ye;
                        qq += ye * ye;
                    }
                }
            }
        }
    } else if (valid) {
        const typename EL::type* x = reinterpret_cast<const typename EL::type*>(p.docs) + (size_t)ci * p.ld;
        const typename EL::type* y = reinterpret_cast<const typename EL::type*>(p.qbuf) + (size_t)q * p.ld;
        constexpr int PF = 24; // 16-byte chunks in flight per row: the kernel is one latency-bound thread per candidate
        const int nchunk = p.ld / EL::PER16;
        u32x4 xr[PF], yr[PF];
#pragma unroll
        for (int t = 0; t < PF; ++t) {
            const int c = t < nchunk ? t : nchunk - 1;
            xr[t] = *reinterpret_cast<const u32x4*>(x + c * EL::PER16);
            yr[t] = *reinterpret_cast<const u32x4*>(y + c * EL::PER16);
        }
        for (int c0 = 0; c0 < nchunk; c0 += PF) {
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                const u32x4 xv = xr[t], yv = yr[t];
                const int nx = c0 + PF + t < nchunk ? c0 + PF + t : nchunk - 1;
                xr[t] = *reinterpret_cast<const u32x4*>(x + nx * EL::PER16);
                yr[t] = *reinterpret_cast<const u32x4*>(y + nx * EL::PER16);
                if (c0 + t < nchunk) {
#pragma unroll
                    for (int e = 0; e < EL::PER16; ++e) {
                        const double xe = (double)EL::get(xv, e);
                        const double ye = (double)EL::get(yv, e);
                        dot += xe * ye;
                        qq += ye * ye; // |q|^2: L2 distances and the margin check
                    }
                }
            }
        }
    }
    rank_flag_write<KL, L2>(p, q, slot, inq, ci, valid, dot, qq, lane);
}

template <int KL, typename EL, bool L2, typename ELQ = EL>
__global__ __launch_bounds__(64) void rescore_rank_kernel(MergeArgs p, const int* cand, int64_t nq) {
    if (p.nq_dev != nullptr) nq = *p.nq_dev;
    rescore_rank_body<KL, EL, L2, ELQ>(p, cand, nq, (int64_t)blockIdx.x, (int)threadIdx.x);
}

// ------------------------------------------------------------------ cross-shard merge (after the all-gather)
// One wave per query; rank by counting.  cand: [nq][c] with c = parts * k <= a few hundred.
__global__ __launch_bounds__(64) void merge_topk_kernel(const float* cand_s, const int64_t* cand_i, int c, int k,
                                                        int metric, float* out_s, int64_t* out_i) {
    const int q = blockIdx.x;
    const float* s = cand_s + (size_t)q * c;
    const int64_t* id = cand_i + (size_t)q * c;
    bool bad = false; // a shard handed over poisoned results: the merged row is poisoned too
    for (int a = threadIdx.x; a < c; a += 64) bad |= id[a] == IDX_POISON;
    if (__ballot(bad) != 0ull) {
        for (int t = threadIdx.x; t < k; t += 64) {
            out_s[(size_t)q * k + t] = poison_score();
            out_i[(size_t)q * k + t] = IDX_POISON;
        }
        return;
    }
    for (int a = threadIdx.x; a < c; a += 64) {
        const float sa = metric == 1 ? -s[a] : s[a];
        const int64_t ia = id[a] < 0 ? INT64_MAX : id[a];
        int rank = 0;
        for (int b = 0; b < c; ++b) {
            const float sb = metric == 1 ? -s[b] : s[b];
            const int64_t ib = id[b] < 0 ? INT64_MAX : id[b];
            const bool before = sb > sa || (sb == sa && (ib < ia || (ib == ia && b < a)));
            rank += before ? 1 : 0;
        }
        if (rank < k) {
            out_s[(size_t)q * k + rank] = s[a];
            out_i[(size_t)q * k + rank] = id[a];
        }
    }
}

// Same merge, reading the all-gathered packed payload directly: gathered [parts][nq][k][2] int64.
__global__ __launch_bounds__(64) void merge_topk_packed_kernel(const int64_t* gathered, int64_t nq, int parts, int k,
                                                               int metric, float* out_s, int64_t* out_i) {
    const int64_t q = blockIdx.x;
    const int c = parts * k;
    auto score = [&](int a) {
        const float f = __uint_as_float((unsigned)gathered[(((size_t)(a / k) * nq + q) * k + a % k) * 2]);
        return metric == 1 ? -f : f;
    };
    auto ident = [&](int a) { return gathered[(((size_t)(a / k) * nq + q) * k + a % k) * 2 + 1]; };
    bool bad = false;
    for (int a = threadIdx.x; a < c; a += 64) bad |= ident(a) == IDX_POISON;
    if (__ballot(bad) != 0ull) {
        for (int t = threadIdx.x; t < k; t += 64) {
            out_s[(size_t)q * k + t] = poison_score();
            out_i[(size_t)q * k + t] = IDX_POISON;
        }
        return;
    }
    for (int a = threadIdx.x; a < c; a += 64) {
        const float sa = score(a);
        const int64_t raw = ident(a);
        const int64_t ia = raw < 0 ? INT64_MAX : raw;
        int rank = 0;
        for (int b = 0; b < c; ++b) {
            const float sb = score(b);
            const int64_t rb = ident(b);
            const int64_t ib = rb < 0 ? INT64_MAX : rb;
            rank += (sb > sa || (sb == sa && (ib < ia || (ib == ia && b < a)))) ? 1 : 0;
        }
        if (rank < k) {
            out_s[(size_t)q * k + rank] = metric == 1 ? -sa : sa;
            out_i[(size_t)q * k + rank] = raw;
        }
    }
}

// ------------------------------------------------------------------ ignore filter of Mips.search on the device
// sotasum/mips.py:388-398: k + 1 hits were fetched; per query drop every hit whose id equals ignore[q] and
// keep the first k of the rest.  One thread per query (k1 <= 30).
__global__ void filter_ignore_kernel(const float* s, const int64_t* id, const int64_t* ignore, int64_t nq, int k1, int k,
                                     float* out_s, int64_t* out_i) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const int64_t banned = ignore[q];
    int w = 0;
    for (int t = 0; t < k1 && w < k; ++t) {
        const int64_t v = id[q * k1 + t];
        if (v != banned) {
            out_s[q * k + w] = s[q * k1 + t];
            out_i[q * k + w] = v;
            ++w;
        }
    }
}

// ------------------------------------------------------------------ cosine re-score of the scoring hook
// sotasum/retriever_generator.py:158-172: scores[b][j] = q_b . c_bj / (|q_b| |c_bj|).  One wave per (b, j);
// fp32 accumulation.  T = float or bf16 bits (uint16_t).
template <typename T>
__device__ __forceinline__ float load_as_f32(const T* p, int64_t i);
template <>
__device__ __forceinline__ float load_as_f32<float>(const float* p, int64_t i) { return p[i]; }
template <>
__device__ __forceinline__ float load_as_f32<uint16_t>(const uint16_t* p, int64_t i) { return bf16_bits_to_f32(p[i]); }

template <typename T>
__global__ __launch_bounds__(256) void cosine_rescore_kernel(const T* query, const T* cls, int64_t pairs, int k, int d,
                                                             float* out, int64_t mem_len, float* bias) {
    const int64_t pr = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pr >= pairs) return;
    const T* q = query + (pr / k) * d;
    const T* c = cls + pr * d;
    float qc = 0.f, qq = 0.f, cc = 0.f;
    for (int t = lane; t < d; t += 64) {
        const float a = load_as_f32<T>(q, t), b = load_as_f32<T>(c, t);
        qc += a * b;
        qq += a * a;
        cc += b * b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        qc += __shfl_xor(qc, off);
        qq += __shfl_xor(qq, off);
        cc += __shfl_xor(cc, off);
    }
    const float cosv = qc / (sqrtf(qq) * sqrtf(cc));
    if (lane == 0) out[pr] = cosv;
    // memory_bias of the hook (retriever_generator.py:188-192): the score of hit (b, j) repeated over
    // the mem_len tokens of that hit, laid out [b, k * mem_len] -> pair pr owns one contiguous run
    if (bias != nullptr)
        for (int64_t t = lane; t < mem_len; t += 64) bias[pr * mem_len + t] = cosv;
}

// Backward of the hook's re-score.  In the reference only the NORMS are under torch.no_grad()
// (retriever_generator.py:160-171); `query @ mips_cls.transpose(1, 2)` stays in the autograd graph and is the path
// through which the retriever's encoders get their gradient.  With w[b][j] = g[b][j] / (|q_b| |c_bj|), norms constant:
//     dL/dq_b = sum_j w[b][j] c_bj          dL/dc_bj = w[b][j] q_b
// g = grad of the scores + the sum over the memory tokens of the grad of memory_bias (an expand, :188-192).
// One 256-thread workgroup per batch row b; k <= 64.
template <typename T>
__global__ __launch_bounds__(256) void cosine_rescore_bwd_kernel(const T* query, const T* cls, int k, int d, const float* g_scores,
                                                                 const float* g_bias, int64_t mem_len, float* g_query, float* g_cls) {
    __shared__ float w[64];
    __shared__ float qq_s;
    const int64_t b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const T* q = query + b * d;
    if (wave == 0) {
        float qq = 0.f;
        for (int t = lane; t < d; t += 64) {
            const float a = load_as_f32<T>(q, t);
            qq += a * a;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) qq += __shfl_xor(qq, off);
        if (lane == 0) qq_s = qq;
    }
    __syncthreads();
    for (int j = wave; j < k; j += 4) {
        const T* c = cls + (b * k + j) * d;
        float cc = 0.f, g = 0.f;
        for (int t = lane; t < d; t += 64) {
            const float v = load_as_f32<T>(c, t);
            cc += v * v;
        }
        if (g_bias != nullptr)
            for (int64_t t = lane; t < mem_len; t += 64) g += g_bias[(b * k + j) * mem_len + t];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            cc += __shfl_xor(cc, off);
            g += __shfl_xor(g, off);
        }
        if (lane == 0) w[j] = ((g_scores ? g_scores[b * k + j] : 0.f) + g) / (sqrtf(qq_s) * sqrtf(cc));
    }
    __syncthreads();
    for (int t = threadIdx.x; t < d; t += 256) {
        const float qv = load_as_f32<T>(q, t);
        float gq = 0.f;
        for (int j = 0; j < k; ++j) {
            gq += w[j] * load_as_f32<T>(cls + (b * k + j) * d, t);
            g_cls[(b * k + j) * d + t] = w[j] * qv;
        }
        g_query[b * d + t] = gq;
    }
}

// ------------------------------------------------------------------ re-scan of flagged queries: gather / scatter
// gather staged query rows (row_bytes each, a multiple of 16) ids[t] -> row t; rows n .. n_out - 1 are zeroed
// n_dev (optional): the row count lives on the device (stream-ordered re-scan); n is then ignored
__global__ void gather_rows_kernel(const unsigned char* src, const int* ids, int64_t n, int64_t n_out, int row_bytes, unsigned char* dst,
                                   const int* n_dev = nullptr) {
    if (n_dev != nullptr) n = *n_dev;
    const int chunks = row_bytes / 16;
    const int64_t total = n_out * chunks;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / chunks;
        const int c = (int)(t % chunks);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < n) v = *reinterpret_cast<const u32x4*>(src + (size_t)ids[row] * row_bytes + (size_t)c * 16);
        *reinterpret_cast<u32x4*>(dst + (size_t)row * row_bytes + (size_t)c * 16) = v;
    }
}
// scatter result rows back: row t of the compact results -> row ids[t]; `words` 8-byte words per row (k for the
// index array or the packed payload's 2 k; the float scores go through scatter_f32)
__global__ void scatter_i64_kernel(const int64_t* src, const int* ids, int64_t n, int words, int64_t* dst, const int* n_dev = nullptr) {
    if (n_dev != nullptr) n = *n_dev;
    const int64_t total = n * words;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        dst[(size_t)ids[t / words] * words + t % words] = src[t];
}
__global__ void scatter_f32_kernel(const float* src, const int* ids, int64_t n, int words, float* dst, const int* n_dev = nullptr) {
    if (n_dev != nullptr) n = *n_dev;
    const int64_t total = n * words;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        dst[(size_t)ids[t / words] * words + t % words] = src[t];
}

// flagged queries -> ascending list of their numbers + count (one workgroup; stream-ordered re-scan)
// zero / nzero: words the same launch clears for what follows (hit counters and the unresolved counter of the exact pass)
// only_above >= 0: the count is written as 0 unless it exceeds only_above (the fall-back re-scan of a search that flagged more
// queries than the exact pass resolves: its launches are sized by this gated count and leave at once otherwise)
__global__ __launch_bounds__(256) void compact_flags_kernel(const unsigned char* flags, int nq, int* ids, int* count, int* zero = nullptr,
                                                            int nzero = 0, unsigned* zero2 = nullptr, int only_above = -1) {
    __shared__ int part[256];
    const int t = threadIdx.x;
    for (int z = t; z < nzero; z += 256) zero[z] = 0;
    if (t == 0 && zero2 != nullptr) *zero2 = 0u;
    const int per = (nq + 255) / 256;
    const int lo = t * per, hi = lo + per < nq ? lo + per : nq;
    int c = 0;
    for (int q = lo; q < hi; ++q) c += flags[q] ? 1 : 0;
    part[t] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) { // inclusive scan
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int w = part[t] - c;
    for (int q = lo; q < hi; ++q)
        if (flags[q]) ids[w++] = q;
    if (t == 255) *count = part[255] > only_above ? part[255] : 0;
}

__global__ void fill_empty_kernel(float* out_s, int64_t* out_i, int64_t* out_packed, int64_t total, int metric) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < total) {
        const float pad = metric == 1 ? INFINITY : -INFINITY;
        if (out_packed) {
            out_packed[2 * t] = (int64_t)__float_as_uint(pad);
            out_packed[2 * t + 1] = -1;
        } else {
            out_s[t] = pad;
            out_i[t] = -1;
        }
    }
}

} // namespace mips
