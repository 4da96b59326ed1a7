/* TEST INFRASTRUCTURE -- C part of the CPU oracle.  NOT product code.
 *
 * Restates, in plain C, the arithmetic the reference's brute-force cross-check
 * performs (sotasum/mips.py:552-560: scores = x @ y.T, descending order, gather
 * top-k) under the build's canonical score definition:
 *
 *     score(q, x) = (float) sum_{k = 0 .. d-1, sequential, double} q[k] * x[k]
 *
 * Inputs are float32 values (already rounded to the index dtype, so each product
 * is exact in double and fused/unfused multiply-add give identical bits); the
 * only freedom of an fp32 implementation -- summation order -- is fixed here.
 * Order: (score desc, index asc), i.e. ties go to the lowest index (the
 * reference leaves ties undefined).
 *
 * Built by oracle/mips_oracle.py:build_c() with
 *     gcc -O2 -ffp-contract=off -fPIC -shared
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static double canon_dot(const float* q, const float* x, int64_t d) {
    double acc = 0.0;
    for (int64_t k = 0; k < d; ++k) acc += (double)q[k] * (double)x[k];
    return acc;
}

/* out[i*c + j] = canon_dot(q[i], x[cand[i*c + j]]); negative candidate -> -inf */
void oracle_canon_pairs(const float* q, const float* x, const int64_t* cand, int64_t nq,
                        int64_t c, int64_t d, double* out) {
    for (int64_t i = 0; i < nq; ++i)
        for (int64_t j = 0; j < c; ++j) {
            int64_t r = cand[i * c + j];
            out[i * c + j] = r < 0 ? -INFINITY : canon_dot(q + i * d, x + r * d, d);
        }
}

/* out[i] = sequential double sum of squares of row i */
void oracle_sumsq(const float* x, int64_t n, int64_t d, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = canon_dot(x + i * d, x + i * d, d);
}

/* does (s1, i1) rank before (s2, i2)?  score desc, then index asc */
static int before(float s1, int64_t i1, float s2, int64_t i2) {
    if (s1 > s2) return 1;
    if (s1 < s2) return 0;
    return i1 < i2;
}

/* Exhaustive canonical top-k.  out_s/out_i are [nq, k]; when k > n the tail is
 * padded with -inf / -1 (FAISS IndexFlat padding convention). */
void oracle_search_exact(const float* q, int64_t nq, const float* x, int64_t n, int64_t d,
                         int k, float* out_s, int64_t* out_i) {
    for (int64_t i = 0; i < nq; ++i) {
        float* bs = out_s + i * k;
        int64_t* bi = out_i + i * k;
        int filled = 0;
        for (int j = 0; j < k; ++j) { bs[j] = -INFINITY; bi[j] = -1; }
        for (int64_t r = 0; r < n; ++r) {
            float s = (float)canon_dot(q + i * d, x + r * d, d);
            if (s != s) continue; /* NaN never ranks */
            if (filled == k && !before(s, r, bs[k - 1], bi[k - 1])) continue;
            int pos = filled < k ? filled : k - 1;
            while (pos > 0 && before(s, r, bs[pos - 1], bi[pos - 1])) {
                bs[pos] = bs[pos - 1];
                bi[pos] = bi[pos - 1];
                --pos;
            }
            bs[pos] = s;
            bi[pos] = r;
            if (filled < k) ++filled;
        }
    }
}
