/* Plain-C consumer of the C ABI (include/mips_hip.h): no Python, no torch, host buffers only.
 * Built and run by tests/test_gpu_parity.py::test_c_abi_from_plain_c:
 *     gcc tests/c_abi_smoke.c -Iinclude -L<lib dir> -lmips_hip -Wl,-rpath,<lib dir> -lm
 * Index: n x d lattice values from a tiny LCG (exact in bf16, sums exact in fp32), so the expected
 * top-k is computed right here with integer arithmetic and compared bit for bit. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mips_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mips_last_error());   \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static unsigned lcg(unsigned* s) { return *s = *s * 1664525u + 1013904223u; }

int main(void) {
    const int64_t n = 5003, d = 768, nq = 37;
    const int k = 5;
    if (mips_abi_version() != MIPS_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    int* xi = malloc(sizeof(int) * n * d);
    int* qi = malloc(sizeof(int) * nq * d);
    float* x = malloc(sizeof(float) * n * d);
    float* q = malloc(sizeof(float) * nq * d);
    unsigned s = 12345u;
    for (int64_t i = 0; i < n * d; ++i) { xi[i] = (int)((lcg(&s) >> 16) % 255) - 127; x[i] = xi[i] / 64.0f; }
    for (int64_t i = 0; i < nq * d; ++i) { qi[i] = (int)((lcg(&s) >> 16) % 255) - 127; q[i] = qi[i] / 64.0f; }
    memcpy(x + 17 * d, x + 4000 * d, sizeof(float) * d); /* a duplicate row: tie -> lower index first */
    memcpy(xi + 17 * d, xi + 4000 * d, sizeof(int) * d);

    mips_index_t* ix = NULL;
    CHECK(mips_index_create(&ix, 0, d, MIPS_DTYPE_BF16, MIPS_METRIC_IP));
    CHECK(mips_index_add(ix, x, 3000, MIPS_DTYPE_F32, 0, NULL));
    CHECK(mips_index_add(ix, x + 3000 * d, n - 3000, MIPS_DTYPE_F32, 0, NULL));
    if (mips_index_ntotal(ix) != n || mips_index_dim(ix) != d) { fprintf(stderr, "ntotal/dim wrong\n"); return 1; }

    float* D = malloc(sizeof(float) * nq * k);
    int64_t* I = malloc(sizeof(int64_t) * nq * k);
    CHECK(mips_search(ix, q, MIPS_DTYPE_F32, nq, k, D, I, 0, 0, NULL));

    int bad = 0;
    for (int64_t a = 0; a < nq; ++a) {
        long best_s[5];
        int64_t best_i[5];
        int filled = 0;
        for (int64_t r = 0; r < n; ++r) {
            long dot = 0;
            for (int64_t c = 0; c < d; ++c) dot += (long)qi[a * d + c] * xi[r * d + c];
            int pos = filled < k ? filled : k - 1;
            if (filled == k && dot <= best_s[k - 1]) continue; /* strict: equal scores keep the lower index */
            while (pos > 0 && dot > best_s[pos - 1]) { best_s[pos] = best_s[pos - 1]; best_i[pos] = best_i[pos - 1]; --pos; }
            best_s[pos] = dot;
            best_i[pos] = r;
            if (filled < k) ++filled;
        }
        for (int t = 0; t < k; ++t)
            if (I[a * k + t] != best_i[t] || D[a * k + t] != (float)best_s[t] / 4096.0f) ++bad;
    }
    int rc = mips_search(ix, q, MIPS_DTYPE_F32, nq, MIPS_MAX_K + 1, D, I, 0, 0, NULL);
    if (rc != MIPS_E_UNSUPPORTED || strlen(mips_last_error()) == 0) { fprintf(stderr, "k limit not reported\n"); return 1; }
    CHECK(mips_index_destroy(ix));
    printf("c_abi_smoke: %lld queries x %lld docs, top-%d, mismatches: %d\n", (long long)nq, (long long)n, k, bad);
    return bad != 0;
}
