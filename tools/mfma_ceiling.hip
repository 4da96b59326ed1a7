// What does v_mfma_f32_32x32x16_bf16 sustain on THIS chip, under its power management, when nothing else is
// in the way?  A diagnostic, not part of the product: bare MFMA loops with the scan kernel's occupancy
// (8 waves per CU, two per SIMD, one accumulation chain per wave), run back to back for >= 2 s per case so
// the clock settles.  Cases: operands all zero / random N(0,1) bf16 in registers / the A operand re-read
// from LDS by one ds_read_b128 per MFMA (the scan kernel's inner loop without DMA, barrier or epilogue).
// The scan kernel's TFLOP/s is to be read against the random-data rows, not against 2.5 PFLOP/s alone.
//
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_ceiling tools/mfma_ceiling.hip && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int NFRAG = 16;   // distinct register-resident B fragments per wave (64 VGPRs)
constexpr int MFMA_PER_IT = 48; // as one 32-document block at d = 768

// MODE 0: A and B from registers; MODE 1: A re-read from LDS (conflict-free rows of 128 B) per MFMA
template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_loop(const uint16_t* src, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    const int lane = threadIdx.x & 63;
    bf16x8 b[NFRAG], a[2];
    const uint16_t* base = src + ((size_t)blockIdx.x * 512 + threadIdx.x) * 8;
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) b[i] = *reinterpret_cast<const bf16x8*>(base + (size_t)i * 512 * 8 * gridDim.x);
    a[0] = b[3];
    a[1] = b[7];
    if (MODE == 1) {
        for (int o = threadIdx.x * 16; o < 48 * 1024; o += 512 * 16)
            *reinterpret_cast<bf16x8*>(lds + o) = *reinterpret_cast<const bf16x8*>(src + (o / 2) % 4096 + (size_t)blockIdx.x * 4096);
        __syncthreads();
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int row = (lane & 31) * 128, h = lane >> 5, swz = ((lane & 31) >> 1) & 7;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            bf16x8 ar[2];
            auto frag = [&](int s) { return *reinterpret_cast<const bf16x8*>(lds + row + (s >> 2) * 4096 + (((2 * (s & 3) + h) ^ swz) << 4)); };
            ar[0] = frag(0);
            ar[1] = frag(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < MFMA_PER_IT; ++s) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ar[s & 1], b[s % NFRAG], acc, 0, 0, 0);
                if (s + 2 < MFMA_PER_IT) ar[s & 1] = frag(s + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < MFMA_PER_IT; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 1], b[s % NFRAG], acc, 0, 0, 0);
        }
        // keep the chain bounded (random data would overflow fp32 after ~1e5 steps): fold it back cheaply
        if ((it & 63) == 63) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] *= 1e-6f;
        }
    }
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[r];
    if (t == 12345.678f) sink[0] = t; // never true; keeps the chain alive
}

// The 16x16x32 shape at the same bytes and flops per wave: per step ONE A fragment (16 documents x 32 k, from
// registers or LDS) feeds two MFMAs (two 16-query column blocks), as scan_kernel_v4 does.  MODE as above.
template <int MODE, bool B_IN_AGPR = false>
__global__ __launch_bounds__(512, 2) void mfma_loop16(const uint16_t* src, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    const int lane = threadIdx.x & 63;
    bf16x8 b[NFRAG], a[2];
    const uint16_t* base = src + ((size_t)blockIdx.x * 512 + threadIdx.x) * 8;
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) b[i] = *reinterpret_cast<const bf16x8*>(base + (size_t)i * 512 * 8 * gridDim.x);
    a[0] = b[3];
    a[1] = b[7];
#if defined(__HIP_DEVICE_COMPILE__)
    if (B_IN_AGPR) { // stationary B fragments in the accumulation half of the register file (MFMA reads them there directly)
#pragma unroll
        for (int i = 0; i < NFRAG; ++i) asm volatile("" : "+a"(b[i]));
    }
#endif
    if (MODE == 1) {
        for (int o = threadIdx.x * 16; o < 48 * 1024; o += 512 * 16)
            *reinterpret_cast<bf16x8*>(lds + o) = *reinterpret_cast<const bf16x8*>(src + (o / 2) % 4096 + (size_t)blockIdx.x * 4096);
        __syncthreads();
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int c = lane & 15, g = lane >> 4, swz = (c >> 1) & 7;
    const int rd0 = c * 128 + ((g ^ swz) << 4);
    for (int it = 0; it < iters; ++it) {
        // 48 steps x 2 MFMAs of 16x16x32 = the flops of 48 MFMAs of 32x32x16 ... per HALF block; two halves
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (MODE == 1) {
                bf16x8 ar[2];
                auto frag = [&](int s) { return *reinterpret_cast<const bf16x8*>(lds + half * 2048 + (s >> 1) * 4096 + ((s & 1) ? (rd0 ^ 64) : rd0)); };
                ar[0] = frag(0);
                ar[1] = frag(1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[s & 1], b[(2 * s) % NFRAG], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[s & 1], b[(2 * s + 1) % NFRAG], acc1, 0, 0, 0);
                    if (s + 2 < 24) ar[s & 1] = frag(s + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1], b[(2 * s) % NFRAG], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1], b[(2 * s + 1) % NFRAG], acc1, 0, 0, 0);
                }
            }
        }
        if ((it & 63) == 63) {
            acc0 *= 1e-6f;
            acc1 *= 1e-6f;
        }
    }
    float t = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
    if (t == 12345.678f) sink[0] = t;
}


// ---- Round 2: what would a mapping with FEWER LDS bytes per MFMA sustain?  mfma_loop16q: one A fragment (16
// documents x 32 k) feeds NQB MFMAs (NQB 16-query column blocks per wave), the wave's stationary B fragments are
// the REAL count (NQB x KS x 4 registers, half of them pinned in AGPRs when they exceed 200), the A operand comes
// from a 48-KiB LDS block image as in scan_kernel_v4.  Shapes:
//   NQB 2, KS 24, 8 waves (2 per SIMD)  = scan_kernel_v4's loop (1 read per 2 MFMAs, 192 fragment registers)
//   NQB 4, KS 24, 4 waves (1 per SIMD)  = 64 stationary queries per wave (1 read per 4 MFMAs, 384 registers)
//   NQB 4, KS 12, 8 waves (2 per SIMD)  = a wave PAIR splits K: each wave holds 64 queries x half of K (192
//                                          registers) and reads only its K half of the block (1 read per 4 MFMAs);
//                                          the partial sums would still have to be exchanged (not modelled here)
// The kernel also reports the clock it ran at (s_memtime / s_memrealtime around the loop, block 0).
template <int NQB, int KS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void mfma_loop16q(const uint16_t* src, float* sink, int iters, unsigned long long* stamps) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    bf16x8 b[NQB][KS];
    const uint16_t* base = src + ((size_t)blockIdx.x * 512 + (threadIdx.x & 511)) * 8;
#pragma unroll
    for (int n = 0; n < NQB; ++n)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            b[n][s] = *reinterpret_cast<const bf16x8*>(base + (size_t)((n * KS + s) % NFRAG) * 512 * 8 * gridDim.x);
#if defined(__HIP_DEVICE_COMPILE__)
            if (NQB * KS * 4 > 200 && (n * KS + s) >= NQB * KS / 2) asm volatile("" : "+a"(b[n][s]));
            else asm volatile("" : "+v"(b[n][s]));
#endif
        }
    for (int o = threadIdx.x * 16; o < 48 * 1024; o += WAVES * 64 * 16)
        *reinterpret_cast<bf16x8*>(lds + o) = *reinterpret_cast<const bf16x8*>(src + (o / 2) % 4096 + (size_t)blockIdx.x * 4096);
    __syncthreads();
    f32x4 acc[NQB];
#pragma unroll
    for (int n = 0; n < NQB; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int c = lane & 15, g = lane >> 4, swz = (c >> 1) & 7;
    const int rd0 = c * 128 + ((g ^ swz) << 4);
    const int koff = KS == 12 ? (wave & 1) * 12 : 0; // K-split pair: odd waves read the second half of K
    unsigned long long t0 = 0, r0 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#endif
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 ar[2];
            auto frag = [&](int s) {
                const int ss = s + koff;
                return *reinterpret_cast<const bf16x8*>(lds + half * 2048 + (ss >> 1) * 4096 + ((ss & 1) ? (rd0 ^ 64) : rd0));
            };
            ar[0] = frag(0);
            ar[1] = frag(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
#pragma unroll
                for (int n = 0; n < NQB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[s & 1], b[n][s], acc[n], 0, 0, 0);
                if (s + 2 < KS) ar[s & 1] = frag(s + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if ((it & 63) == 63) {
#pragma unroll
            for (int n = 0; n < NQB; ++n) acc[n] *= 1e-6f;
        }
    }
    unsigned long long t1 = 0, r1 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
#endif
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
    float t = 0.f;
#pragma unroll
    for (int n = 0; n < NQB; ++n) t += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    if (t == 12345.678f) sink[0] = t;
}

typedef void (*kernq_t)(const uint16_t*, float*, int, unsigned long long*);
static void runq(kernq_t kern, int threads, double flops_per_wg_iter, const char* name, const uint16_t* d_src, float* d_sink, int grid,
                 double seconds, unsigned long long* d_stamps) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) kern<<<grid, threads>>>(d_src, d_sink, iters, d_stamps);
    CHECK(hipDeviceSynchronize());
    double best = 0, last = 0, total_ms = 0;
    int launches = 0;
    while (total_ms < seconds * 1e3) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 8; ++r) kern<<<grid, threads>>>(d_src, d_sink, iters, d_stamps);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        launches += 8;
        last = 8.0 * grid * iters * flops_per_wg_iter / (ms * 1e-3) / 1e12;
        if (last > best) best = last;
    }
    static unsigned long long h[2 * 1024];
    CHECK(hipMemcpy(h, d_stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double ghz[1024];
    for (int i = 0; i < grid; ++i) ghz[i] = h[2 * i + 1] ? (double)h[2 * i] / (double)h[2 * i + 1] * 0.1 : 0.0;
    for (int i = 1; i < grid; ++i) // insertion sort: median over workgroups
        for (int j = i; j > 0 && ghz[j] < ghz[j - 1]; --j) { double t = ghz[j]; ghz[j] = ghz[j - 1]; ghz[j - 1] = t; }
    const double cyc_per_mfma = (double)h[0] / ((double)iters * flops_per_wg_iter / (threads / 64) / 16384.0);
    printf("{\"case\": \"%s\", \"tflops_last\": %.1f, \"tflops_best\": %.1f, \"launches\": %d, \"workgroups\": %d, \"clock_ghz_median\": %.3f, \"cycles_per_mfma_wave0\": %.2f}\n",
           name, last, best, launches, grid, ghz[grid / 2], cyc_per_mfma);
    fflush(stdout);
}

// ---- fp8 (e4m3) on the block-scaled f8f6f4 instructions with unit scales, same structure: 32x32x64 (one 32-byte
// A fragment per MFMA, two ds_read_b128) against 16x16x128 (one 32-byte A fragment per TWO MFMAs).
typedef int v8i32 __attribute__((ext_vector_type(8)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
constexpr int NFRAG8 = 8; // 64 VGPRs of B fragments

template <int MODE, int SHAPE> // SHAPE 0: 32x32x64, 1: 16x16x128
__global__ __launch_bounds__(512, 2) void mfma_loop_f8(const uint16_t* src, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[48 * 1024];
    const int lane = threadIdx.x & 63;
    v8i32 b[NFRAG8], a[2];
    const uint16_t* base = src + ((size_t)blockIdx.x * 512 + threadIdx.x) * 8;
#pragma unroll
    for (int i = 0; i < NFRAG8; ++i) {
        const v4i32 lo = *reinterpret_cast<const v4i32*>(base + (size_t)(2 * i) * 512 * 8 * gridDim.x);
        const v4i32 hi = *reinterpret_cast<const v4i32*>(base + (size_t)(2 * i + 1) * 512 * 8 * gridDim.x);
        b[i] = v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    a[0] = b[3];
    a[1] = b[5];
    if (MODE == 1) {
        for (int o = threadIdx.x * 16; o < 48 * 1024; o += 512 * 16)
            *reinterpret_cast<v4i32*>(lds + o) = *reinterpret_cast<const v4i32*>(src + (o / 2) % 4096 + (size_t)blockIdx.x * 4096);
        __syncthreads();
    }
    f32x16 acc32;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc32[r] = 0.f;
    const int off = (lane * 32) % (16 * 1024);
    for (int it = 0; it < iters; ++it) {
        // 24 KiB "block" of 32 rows x 768 bytes = 12 steps of 32x32x64 or 2 halves x 6 steps x 2 MFMAs of 16x16x128
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            v8i32 av = a[s & 1];
            if (MODE == 1) {
                const v4i32 lo = *reinterpret_cast<const v4i32*>(lds + off + s * 2048);
                const v4i32 hi = *reinterpret_cast<const v4i32*>(lds + off + s * 2048 + 16);
                av = v8i32{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (SHAPE == 0) {
                acc32 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, b[s % NFRAG8], acc32, 0, 0, 0, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, b[s % NFRAG8], acc0, 0, 0, 0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, b[(s + 3) % NFRAG8], acc1, 0, 0, 0, 0, 0, 0);
            }
        }
        if ((it & 63) == 63) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[r] *= 1e-6f;
            acc0 *= 1e-6f;
            acc1 *= 1e-6f;
        }
    }
    float t = acc0[0] + acc1[0] + acc0[3] + acc1[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc32[r];
    if (t == 12345.678f) sink[0] = t;
}

static uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef void (*kern_t)(const uint16_t*, float*, int);
static void run(kern_t kern, double flops_per_wave_iter, const char* name, const uint16_t* d_src, float* d_sink, int grid, double seconds) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) kern<<<grid, 512>>>(d_src, d_sink, iters);
    CHECK(hipDeviceSynchronize());
    double best = 0, last = 0, total_ms = 0;
    int launches = 0;
    while (total_ms < seconds * 1e3) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 8; ++r) kern<<<grid, 512>>>(d_src, d_sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        launches += 8;
        const double flops = 8.0 * grid * 8.0 * iters * flops_per_wave_iter;
        last = flops / (ms * 1e-3) / 1e12;
        if (last > best) best = last;
    }
    printf("{\"case\": \"%s\", \"tflops_last\": %.1f, \"tflops_best\": %.1f, \"launches\": %d, \"workgroups\": %d}\n", name, last, best, launches, grid);
    fflush(stdout);
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int grid = prop.multiProcessorCount; // one 8-wave workgroup per CU, as the scan kernel
    const size_t n = (size_t)NFRAG * grid * 512 * 8 + 4096 * (size_t)grid;
    uint16_t* h = (uint16_t*)malloc(n * 2);
    uint16_t *d_rand, *d_zero;
    float* d_sink;
    // N(0,1)-like bf16 (sum of 8 uniforms, as the bench data generator)
    for (size_t i = 0; i < n; ++i) {
        uint64_t a = mix(i * 2), b = mix(i * 2 + 1);
        int64_t s = 0;
        for (int sh = 0; sh < 64; sh += 16) s += (int64_t)((a >> sh) & 0xFFFF) + (int64_t)((b >> sh) & 0xFFFF);
        float f = (float)(s - 4 * 65535) * (1.0f / (65536.0f * 0.8165f));
        uint32_t u;
        memcpy(&u, &f, 4);
        h[i] = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
    CHECK(hipMalloc(&d_rand, n * 2));
    CHECK(hipMalloc(&d_zero, n * 2));
    CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMemcpy(d_rand, h, n * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_zero, 0, n * 2));
    printf("{\"device\": \"%s\", \"compute_units\": %d}\n", prop.name, grid);
    const double f32 = MFMA_PER_IT * 32768.0;      // 48 MFMAs of 32x32x16
    const double f16 = 2 * 24 * 2 * 16384.0;       // two halves x 24 steps x 2 MFMAs of 16x16x32 (same flops)
    if (getenv("MFMA_CEILING_ROUND2")) { // the round-2 mapping study only
        unsigned long long* d_stamps;
        CHECK(hipMalloc(&d_stamps, sizeof(unsigned long long) * 2 * 1024));
        CHECK(hipMemset(d_stamps, 0, sizeof(unsigned long long) * 2 * 1024));
        const double mf = 16384.0;
        for (int rep = 0; rep < 2; ++rep) {
            runq(mfma_loop16q<2, 24, 8>, 512, 8 * 2 * 24 * 2 * mf, "16x16x32 LDS-fed, 2 query blocks/wave (v4 loop), 8 waves, random", d_rand, d_sink, grid, 2.0, d_stamps);
            runq(mfma_loop16q<4, 24, 4>, 256, 4 * 2 * 24 * 4 * mf, "16x16x32 LDS-fed, 4 query blocks/wave, 4 waves (1/SIMD), random", d_rand, d_sink, grid, 2.0, d_stamps);
            runq(mfma_loop16q<4, 12, 8>, 512, 8 * 2 * 12 * 4 * mf, "16x16x32 LDS-fed, 4 query blocks/wave x half K (wave-pair K split), 8 waves, random", d_rand, d_sink, grid, 2.0, d_stamps);
        }
        runq(mfma_loop16q<2, 24, 8>, 512, 8 * 2 * 24 * 2 * mf, "16x16x32 LDS-fed, 2 query blocks/wave (v4 loop), 8 waves, zeros", d_zero, d_sink, grid, 2.0, d_stamps);
        runq(mfma_loop16q<4, 24, 4>, 256, 4 * 2 * 24 * 4 * mf, "16x16x32 LDS-fed, 4 query blocks/wave, 4 waves (1/SIMD), zeros", d_zero, d_sink, grid, 2.0, d_stamps);
        runq(mfma_loop16q<4, 12, 8>, 512, 8 * 2 * 12 * 4 * mf, "16x16x32 LDS-fed, wave-pair K split, 8 waves, zeros", d_zero, d_sink, grid, 2.0, d_stamps);
        run(mfma_loop16<0>, f16, "16x16x32, registers, random N(0,1) bf16", d_rand, d_sink, grid, 2.0);
        return 0;
    }
    run(mfma_loop<0>, f32, "32x32x16, registers, zeros", d_zero, d_sink, grid, 2.0);
    run(mfma_loop<0>, f32, "32x32x16, registers, random N(0,1) bf16", d_rand, d_sink, grid, 2.0);
    run(mfma_loop<1>, f32, "32x32x16, A from LDS (ds_read_b128 per MFMA), zeros", d_zero, d_sink, grid, 2.0);
    run(mfma_loop<1>, f32, "32x32x16, A from LDS (ds_read_b128 per MFMA), random N(0,1) bf16", d_rand, d_sink, grid, 2.0);
    run(mfma_loop16<0>, f16, "16x16x32, registers, random N(0,1) bf16", d_rand, d_sink, grid, 2.0);
    run(mfma_loop16<1>, f16, "16x16x32, A from LDS (ds_read_b128 per 2 MFMAs), zeros", d_zero, d_sink, grid, 2.0);
    run(mfma_loop16<1>, f16, "16x16x32, A from LDS (ds_read_b128 per 2 MFMAs), random N(0,1) bf16", d_rand, d_sink, grid, 2.0);
    run(mfma_loop16<1, true>, f16, "16x16x32, A from LDS, B fragments in AGPRs, random N(0,1) bf16", d_rand, d_sink, grid, 2.0);
    run(mfma_loop16<1>, f16, "16x16x32, A from LDS (repeat), random N(0,1) bf16", d_rand, d_sink, grid, 2.0);
    // fp8: reinterpret the random bf16 bit patterns as e4m3 bytes, NaN codes (0x7f / 0xff) cleared
    {
        uint8_t* hb = (uint8_t*)h;
        for (size_t i = 0; i < n * 2; ++i) {
            hb[i] = (uint8_t)(mix(i) >> 17);
            if ((hb[i] & 0x7f) == 0x7f) hb[i] &= 0x77;
        }
        CHECK(hipMemcpy(d_rand, h, n * 2, hipMemcpyHostToDevice));
    }
    const double f8a = 12 * 2.0 * 32 * 32 * 64;        // 12 MFMAs of 32x32x64
    const double f8b = 12 * 2 * 2.0 * 16 * 16 * 128;   // 12 steps x 2 MFMAs of 16x16x128 (same flops)
    run(mfma_loop_f8<0, 0>, f8a, "fp8 32x32x64, registers, random e4m3 bytes", d_rand, d_sink, grid, 2.0);
    run(mfma_loop_f8<1, 0>, f8a, "fp8 32x32x64, A from LDS (2 ds_read_b128 per MFMA), random e4m3 bytes", d_rand, d_sink, grid, 2.0);
    run(mfma_loop_f8<0, 1>, f8b, "fp8 16x16x128, registers, random e4m3 bytes", d_rand, d_sink, grid, 2.0);
    run(mfma_loop_f8<1, 1>, f8b, "fp8 16x16x128, A from LDS (2 ds_read_b128 per 2 MFMAs), random e4m3 bytes", d_rand, d_sink, grid, 2.0);
    run(mfma_loop_f8<1, 1>, f8b, "fp8 16x16x128, A from LDS, zeros", d_zero, d_sink, grid, 2.0);
    return 0;
}
