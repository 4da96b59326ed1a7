/* mips_hip.h -- C ABI of the MI355X-native MIPS (maximum inner product search) backend.
 *
 * Drop-in boundary for ONE path of florianbaud/retrieval-augmented-mds ("sotasum"):
 * the exact top-k inner-product search that sotasum/mips.py performs through a CPU
 * FAISS IndexFlat.  The reference is pure Python and has no FFI of its own; the seam it
 * exposes is the Python duck type `faiss_index.search(x, k) -> (D, I)`
 * (sotasum/mips.py:383-386).  The entry points below are what a ctypes binding for that
 * seam binds (INTEGRATION.md shows the stub); each one cites the reference interface it
 * stands in for.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this boundary.
 *   - every function returns 0 on success or a negative MIPS_E_* code; the message for
 *     the calling thread is available from mips_last_error().  No C++ exception escapes.
 *   - the caller owns every input/output buffer; the library owns the index storage in
 *     HBM and its scratch.  Work is enqueued on the HIP stream passed in (NULL = the
 *     default stream); there is no hidden device-wide synchronisation.  Calls that take
 *     HOST output buffers synchronise that stream before returning.
 *   - one index lives on one GPU.  Concurrent calls on one index (from several host
 *     threads) must be serialised by the caller (the Python wrapper holds a lock).  Calls
 *     issued on different streams are ordered by the library: a call arriving on another
 *     stream than the previous one records an event on that previous stream and waits for
 *     it, so the shared scratch is never used by two in-flight searches (a stream handed
 *     to the library must stay valid until the next call on the index).
 *   - ties: the lowest document index wins (the reference leaves ties undefined).
 *   - padding (FAISS IndexFlat convention): when k > ntotal the tail of every row is
 *     idx = -1, score = -inf (inner product) or +inf (L2).
 *
 * Score definition (see DESIGN.md "Canonical score"): candidates are found with bf16 MFMA
 * products accumulated in fp32; the final k results are re-scored exactly,
 *     score = (float) sum_{j=0..d-1, sequential, fp64} q[j] * x[j]
 * on the values as stored in the index (bf16-rounded), and ordered by (score desc, idx asc).
 */
#ifndef MIPS_HIP_H
#define MIPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPS_ABI_VERSION 1

/* element types of caller buffers / of the index storage */
#define MIPS_DTYPE_F32 0
#define MIPS_DTYPE_BF16 1
#define MIPS_DTYPE_FP8_E4M3 2 /* OCP e4m3 bytes: index storage AND queries quantised to e4m3, fp8 MFMA (d <= 1024, k <= 13) */
#define MIPS_DTYPE_FP8_E4M3_DOCS 3 /* index storage only (mips_index_create): OCP e4m3 rows, bf16 QUERIES -- BASELINE config 5 as it is
                                      worded ("fp8 e4m3 doc embeddings"); d <= 1024 */

/* faiss.METRIC_INNER_PRODUCT / faiss.METRIC_L2 as used by mips.py:306,316,369,371 */
#define MIPS_METRIC_IP 0
#define MIPS_METRIC_L2 1

/* mips_search flags */
#define MIPS_Q_DEVICE 1   /* queries pointer is device memory */
#define MIPS_OUT_DEVICE 2 /* out_scores / out_idx are device memory */
#define MIPS_FORCE_IP 8   /* rank by inner product for this call even on an L2 index (Mips.np_search, mips.py:527-529) */
#define MIPS_OUT_PACKED 4 /* with MIPS_OUT_DEVICE: out_idx receives [nq, k, 2] int64 = {float32 score bits
                             (zero-extended), index}, the all-gather payload; out_scores is ignored */

/* synthetic data kinds (same functions as oracle/synth.py) */
#define MIPS_SYNTH_LATTICE 0
#define MIPS_SYNTH_GAUSS 1
#define MIPS_SYNTH_LATTICE_FP8 2

#define MIPS_OK 0
#define MIPS_E_INVALID -1     /* bad argument */
#define MIPS_E_HIP -2         /* a HIP runtime call failed */
#define MIPS_E_UNSUPPORTED -3 /* valid request this build does not implement */
#define MIPS_E_NOMEM -4       /* device allocation failed */
#define MIPS_E_SCAN_TIMEOUT -5 /* a scan kernel gave up on its bounded block barrier: that search's output is poisoned */

/* What a search whose scan kernel timed out writes into EVERY output slot instead of results: idx =
 * MIPS_IDX_POISON (distinct from the -1 padding of k > ntotal), score = NaN.  mips_merge_topk(_packed) propagate
 * it (a poisoned shard poisons the merged row), so the failure is visible on device-resident paths without any
 * synchronisation; the host learns of it from the next call on the index or from mips_index_check_error. */
#define MIPS_IDX_POISON (-2)

#define MIPS_MAX_K 29 /* largest k one mips_search call accepts */

typedef struct mips_index mips_index_t;

/* ABI version of the loaded library (compare with MIPS_ABI_VERSION). */
int mips_abi_version(void);

/* Message of the last failing call on this thread ("" if none). */
const char* mips_last_error(void);

/* Create an empty index of dimension d on GPU `device`.
 * Replaces: datasets.Dataset.add_faiss_index(string_factory="Flat", metric_type=...) ->
 * faiss.IndexFlat(d, metric), sotasum/mips.py:333-340 and retriever_lightning.py:395-404.
 * doc_dtype is the storage type in HBM:
 *   MIPS_DTYPE_BF16      2 B/element, inputs rounded to bf16 (RNE); the fast path
 *   MIPS_DTYPE_FP8_E4M3  1 B/element, inputs AND queries rounded to OCP e4m3 (d <= 1024, k <= 13): the fp8 MFMA at twice the
 *                        bf16 rate -- for many queries per pass over the index
 *   MIPS_DTYPE_FP8_E4M3_DOCS  1 B/element, inputs rounded to OCP e4m3, queries to bf16: half the HBM bytes of a bf16 index with
 *                        bf16 query precision; the rows are up-converted on their way into the bf16 MFMA
 *                        (csrc/scan_kernel_e8.hpp), which costs vector instructions -- for FEW queries per pass (<= 64 per tile),
 *                        where bytes bind.  Canonical score: exact products of (e4m3 row element, bf16 query element), fp64 sum
 *   MIPS_DTYPE_F32       fp32-exact: results are those of an fp32 brute force on the caller's values -- what the Python
 *                        facade (Mips / KnowledgeBase.add_faiss_index / inner_product) creates by default, because the
 *                        reference's embeddings are fp32 and "drop-in" means ITS neighbours
 *                        (bf16 hi|lo planes for the three-segment scan + the fp32 rows for the exact re-score, 8 B/element;
 *                        for d <= 1024 also bf16(x) alone at the fast kernels' row pitch, + 2 B/element: searches that
 *                        certify (host buffers; device outputs certify on the stream, see mips_index_margin_stats) scan
 *                        THAT like a bf16 index, re-score on the fp32 rows, and settle the queries whose margin -- widened by the
 *                        representation error |x - bf16 x| |q| + |bf16 x| |q - bf16 q| -- is not certified by the exact pass
 *                        over the fp32 rows: same results, ~5x the rate on well-separated data; "f32_fast" below.  The
 *                        one-launch kernel of mips_search_fused serves this storage as well). */
int mips_index_create(mips_index_t** out, int device, int64_t d, int doc_dtype, int metric);

/* Free the index and its scratch.  Replaces Dataset.drop_index (mips.py:537). */
int mips_index_destroy(mips_index_t* index);

/* Pre-size the HBM storage for n rows (optional; add grows geometrically otherwise). */
int mips_index_reserve(mips_index_t* index, int64_t n);

/* Append n rows of dimension d.  rows: [n, d] row-major of src_dtype (F32 or BF16), in
 * host (src_is_device = 0) or device memory.  F32 input is rounded to bf16 (RNE) on the
 * device.  Replaces faiss Index.add as driven by HF datasets in batches of 1000 rows
 * (mips.py:333-340). */
int mips_index_add(mips_index_t* index, const void* rows, int64_t n, int src_dtype,
                   int src_is_device, void* hip_stream);

/* Drop all rows, keep the allocation.  Replaces faiss Index.reset. */
int mips_index_reset(mips_index_t* index);

/* faiss Index.ntotal / Index.d / metric_type */
int64_t mips_index_ntotal(const mips_index_t* index);
int64_t mips_index_dim(const mips_index_t* index);
int mips_index_metric(const mips_index_t* index);

/* phi = max_i |x_i|^2 over the stored rows (fp64), the constant of the MIPS->L2 reduction
 * (mips.py:55-56, 316-324).  Synchronises the stream. */
int mips_index_phi(mips_index_t* index, double* out_phi, void* hip_stream);

/* Override phi (>= 0).  A row-sharded index needs ONE phi for all shards -- the maximum of the shards'
 * local values -- or L2 distances from different shards are not comparable.  Stays in force until
 * mips_index_reset or until it is dropped by passing a NEGATIVE phi (the next mips_index_phi then recomputes this
 * shard's own maximum): rows added under an override do not change it, so after adding rows the caller drops it,
 * re-reads the local values, reduces them over the shards and sets the result again. */
int mips_index_set_phi(mips_index_t* index, double phi);

/* Copy stored rows [row0, row0+n) in the index dtype ([n, d]: uint16 bf16 bits, uint8 e4m3 codes or float32) to HOST memory.
 * Used by save() (replaces Dataset.save_faiss_index, mips.py:536) and by tests. */
int mips_index_read_rows(mips_index_t* index, int64_t row0, int64_t n, void* out_host_u16,
                         void* hip_stream);

/* Append n rows generated on the device by the counter-based generator
 * value(seed, row, col) with row = row0 .. row0+n-1 (global row numbers, so row shards of
 * one logical index are generated independently).  kind: MIPS_SYNTH_*. */
int mips_index_add_synthetic(mips_index_t* index, int64_t n, int64_t row0, uint64_t seed,
                             int kind, void* hip_stream);

/* Fill a DEVICE buffer [n, d] of dtype (F32 or BF16) with generator values. */
int mips_synth_fill(void* out_device, int64_t n, int64_t d, int64_t row0, uint64_t seed,
                    int kind, int dtype, int device, void* hip_stream);

/* Exact top-k search.  Replaces faiss_index.search(queries, k) at sotasum/mips.py:383-386
 * (and get_nearest_examples_batch at retriever_lightning.py:317-321).
 *   q          [nq, d] row-major, q_dtype F32 or BF16, host or device (MIPS_Q_DEVICE)
 *   out_scores [nq, k] float32, out_idx [nq, k] int64, host or device (MIPS_OUT_DEVICE)
 *   idx_offset added to every returned index (row offset of this shard)
 * Inner product: scores descending.  L2: squared distances on the phi-augmented vectors
 * (|q|^2 + phi - 2 q.x, mips.py:59-70,316-331), ascending. */
int mips_search(mips_index_t* index, const void* q, int q_dtype, int64_t nq, int k,
                float* out_scores, int64_t* out_idx, int64_t idx_offset, int flags,
                void* hip_stream);

/* mips_search with its two halves on two streams: query staging and the fused scan are enqueued on scan_stream, the
 * candidate selection and exact re-score on tail_stream (behind an event), device outputs only.  Consecutive calls use
 * two alternating scratch sets, so the scan of search t + 1 starts right behind the scan of search t while the tail
 * of t -- and whatever the caller enqueues behind it on tail_stream: the all-gather and merge of a row-sharded search
 * (sharded.py) -- runs beside it.  The results are complete on tail_stream; any other stream must wait for it (an
 * event recorded on tail_stream after the call).  Margin check: as mips_search with device outputs -- the exact pass over the
 * flagged queries is part of the tail (mips_index_margin_stats waits for it when asked to synchronise).  Nothing in the
 * reference corresponds. */
int mips_search_split(mips_index_t* index, const void* q, int q_dtype, int64_t nq, int k, float* out_scores,
                      int64_t* out_idx, int64_t idx_offset, int flags, void* scan_stream, void* tail_stream);

/* The device-resident scoring hook in one call: what retriever_generator.py:143-153 -> mips.py:421-422 does per
 * training / generation step -- `_prepare_query` (row normalisation for the normalised inner-product index,
 * mips.py:369-370), `search` with k or k + 1 hits and the `ignore_indexes` filter of mips.py:388-398 -- on DEVICE
 * buffers, stream-ordered, no host hop:
 *   q_device [nq, d] float32 or bf16; normalize != 0 (float32 only): faiss.normalize_L2 arithmetic, the caller's
 *   buffer is NOT modified; ignore_device [nq] int64 or NULL; out_scores / out_idx DEVICE [nq, k].
 * For the reference's own call shape (nq <= 16, bf16 or fp32-exact index of <= 65536 rows, k + 1 <= 6) all of it is ONE
 * kernel launch (csrc/tiny_search.hpp: staging, MFMA scan, select, exact re-score, filter) followed -- unless the margin check
 * is off or "count only" -- by the stream-ordered exact pass over the queries that kernel flagged (two launches that leave at
 * once when nothing was flagged; the ignore filter is applied to their result too): the hook's results are CERTIFIED without a
 * synchronisation.  Other shapes run the same steps as separate launches.  Results are identical either way ("tiny" = 0 in
 * mips_index_set_param forces the general path, "tiny" = 2 the one-launch kernel with its fall-back selection and sequential
 * re-score -- test knobs; mips_search takes the one-launch kernel for eligible shapes as well). */
int mips_search_fused(mips_index_t* index, const void* q_device, int q_dtype, int64_t nq, int k, int normalize,
                      const int64_t* ignore_device, float* out_scores_device, int64_t* out_idx_device,
                      int64_t idx_offset, void* hip_stream);

/* Merge `parts` per-shard top-k lists into the global top-k (the step after the RCCL
 * all-gather; nothing in the reference corresponds, its index is replicated per rank,
 * lightning_model.py:180).  cand_s / cand_i: DEVICE [nq, parts*k] (shard-major inside a
 * row), out_s / out_i: DEVICE [nq, k].  Order (score desc | asc for L2, idx asc), idx < 0
 * last. */
int mips_merge_topk(const float* cand_s, const int64_t* cand_i, int64_t nq, int parts, int k,
                    int metric, float* out_s, int64_t* out_i, int device, void* hip_stream);

/* The same merge reading the all-gathered MIPS_OUT_PACKED payload as it arrives:
 * gathered = DEVICE [parts, nq, k, 2] int64 (rank-major). */
int mips_merge_topk_packed(const int64_t* gathered, int64_t nq, int parts, int k, int metric,
                           float* out_s, int64_t* out_i, int device, void* hip_stream);

/* The ignore filter of Mips.search (sotasum/mips.py:388-398) on the device: from k_fetched (= k + 1)
 * hits per query drop every hit whose id equals ignore[q] and keep the first k.  All DEVICE buffers:
 * scores/idx [nq, k_fetched], ignore [nq] int64, out_s/out_i [nq, k]. */
int mips_filter_ignore(const float* scores, const int64_t* idx, const int64_t* ignore, int64_t nq,
                       int k_fetched, int k, float* out_s, int64_t* out_i, int device, void* hip_stream);

/* Cosine re-score of the retriever scoring hook (sotasum/retriever_generator.py:158-172):
 * out[b][j] = q_b . c_bj / (|q_b| |c_bj|).  DEVICE buffers: query [b, d], cls [b, k, d] of dtype
 * (MIPS_DTYPE_F32 or MIPS_DTYPE_BF16), out [b, k] float32 (fp32 accumulation). */
int mips_cosine_rescore(const void* query, const void* cls, int dtype, int64_t b, int k, int64_t d,
                        float* out, int device, void* hip_stream);

/* The same re-score fused with the hook's memory_bias broadcast (retriever_generator.py:188-192:
 * mips_scores.unsqueeze(-1).expand(-1, -1, memory_seq_len).reshape(b, -1)): additionally writes
 * memory_bias[b][j * memory_seq_len + t] = out[b][j] for t < memory_seq_len (DEVICE float32
 * [b, k * memory_seq_len]).  memory_seq_len == 0 skips the broadcast. */
int mips_cosine_rescore_bias(const void* query, const void* cls, int dtype, int64_t b, int k, int64_t d,
                             float* out, int64_t memory_seq_len, float* memory_bias, int device,
                             void* hip_stream);

/* Backward of the two functions above.  In the reference only the norms sit under torch.no_grad()
 * (retriever_generator.py:160-171): `query @ mips_cls.transpose(1, 2)` stays in the autograd graph, and memory_bias
 * is the path through which the query and memory encoders receive their retrieval gradient.  With
 * g[b][j] = grad_scores[b][j] + sum_t grad_memory_bias[b][j * memory_seq_len + t] (either pointer may be NULL) and
 * w = g / (|q| |c|), norms treated as constants:  grad_query[b] = sum_j w[b][j] c_bj,  grad_cls[b][j] = w[b][j] q_b.
 * All DEVICE buffers; grads float32 ([b, d] and [b, k, d]); k <= 64. */
int mips_cosine_rescore_backward(const void* query, const void* cls, int dtype, int64_t b, int k, int64_t d,
                                 const float* grad_scores, const float* grad_memory_bias, int64_t memory_seq_len,
                                 float* grad_query, float* grad_cls, int device, void* hip_stream);

/* In-place row L2 normalisation of a DEVICE float32 matrix [n, d].  Replaces
 * faiss.normalize_L2 behind Mips.l2_normalization (sotasum/mips.py:521-525), used for documents
 * at build time (mips.py:306-314, 358-361) and for queries (mips.py:369-370).  Rows of norm 0 are
 * left unchanged (faiss fvec_renorm_L2). */
int mips_l2_normalize(float* x_device, int64_t n, int64_t d, int device, void* hip_stream);

/* max_i |x_i|^2 of a DEVICE float32 matrix, written to *out_host (stream synchronised).  Its
 * square root is the reference's max_norm (mips.py:298-304, 347-349). */
int mips_rows_max_sumsq(const float* x_device, int64_t n, int64_t d, double* out_host, int device,
                        void* hip_stream);

/* The same maximum accumulated into a DEVICE double (acc_device = max(acc_device, max_i |x_i|^2); the caller zeroes it once):
 * nothing is copied to the host, nothing synchronises.  For index builds that receive the rows in encoder batches
 * (Mips.add_embeddings): the running max-norm stays on the device and is read once when the build ends. */
int mips_rows_max_sumsq_device(const float* x_device, int64_t n, int64_t d, double* acc_device, int device, void* hip_stream);

/* Tuning knobs of the scan launch (0 = automatic): "nsplit" = number of index splits (rounded up
 * to a multiple of 8), "qgroups" = query-tile groups per XCD octet (1, 2, 4 or 8), "variant" = scan kernel (1 = 128x128
 * register-staged tiles, 3 = query-stationary on the 32x32x16 MFMA shape, 4 = query-stationary on the
 * 16x16x32 shape (d padding to 384 .. 768, k <= 5), 7 = wave pairs splitting K at row pitch 1024 (k <= 5); 5 and 6 name
 * measured alternatives that exist in the A/B build of tools/ab.py only and are ignored here).  Results never depend on
 * these four; only speed does.
 * "margin_check" (0 .. 4) selects what happens to queries whose candidate pool is not provably wide enough: see
 * mips_index_margin_stats.  "resolve_budget" (default 0 = 1024): flagged queries one search settles at most.
 * "f32_fast" (fp32-exact index): 0 = always the three-segment scan, 1 (default) = two-stage search in every call that certifies
 * (which is every call unless "margin_check" is 0 or 4), skipped for 8 calls after a SYNCHRONISING call (host buffers,
 * "margin_check" = 2) that sent more than an eighth of its queries to the second stage -- stream-ordered calls never change
 * the path of later ones -- 2 = two-stage always (with "margin_check" = 4 the uncertified queries are then only COUNTED).
 * The same switch (alias "optimistic") governs bf16 searches with 8 <= k <= 13 that certify at row pitches 384 .. 768: their
 * pool of 32 candidates is then selected from the 16x16x32 kernel's sub-lists (the fast kernel) instead of from true K' = 16
 * lists; the margin check decides per query whether that pool was wide enough and the others are settled exactly.
 * Two more names exist for tests and experiments and are NOT tuning knobs:
 *   "spin_limit"  polls a wave spends on the scan's block barrier before it gives up (0 = the shipped 2^22).  A tiny
 *                 value makes the kernel give up spuriously and -1 makes every scan launch raise its error word
 *                 unconditionally -- that is their purpose: tests use them to drive the MIPS_E_SCAN_TIMEOUT /
 *                 MIPS_IDX_POISON path.  Never set it in production.
 *   "sub"         A/B selector of experimental kernel instances (profiles/ experiment logs), two of which skip the
 *                 top-k epilogue and return wrong results by design.  The shipped library does not contain them:
 *                 any value but 0 returns MIPS_E_UNSUPPORTED unless the library was built with -DMIPS_EXPERIMENTAL
 *                 (tools/ab.py does that into tools/_build/). */
int mips_index_set_param(mips_index_t* index, const char* name, int64_t value);

/* Scan-error check.  The fused scan kernels synchronise their waves per document block through a bounded poll; a
 * wave that gives up sets an error word, the exact re-score of the same call then writes MIPS_IDX_POISON / NaN into
 * every output slot and raises a sticky host-visible flag on the index.  This call reports (MIPS_E_SCAN_TIMEOUT) and
 * clears that flag; with synchronize != 0 it first waits for `hip_stream`, so that every search enqueued there so
 * far is covered.  mips_search itself performs the same check on entry (for the searches before it) and, when it
 * writes to HOST buffers, after its own synchronisation.  Nothing in the reference corresponds (faiss IndexFlat
 * cannot fail this way). */
int mips_index_check_error(mips_index_t* index, int synchronize, void* hip_stream);

/* Margin check -- is "exact" certified for this query?  The MFMA scores (fp32 accumulation) only select a candidate
 * pool that is then re-scored exactly; a true top-k document can be missing from the pool only if enough others score
 * within the MFMA rounding error of it.  Every search therefore compares, per query, the exact k-th score tk with the
 * best MFMA score B anything OUTSIDE the pool can have had (unpopped list entries, documents rejected by an insert
 * bound, documents dropped from a full running list) and FLAGS the query when B + e >= tk, e = d 2^-23 |q| max|x|
 * (a rigorous bound for fp32 accumulation of the exact bf16 / e4m3 products).  "margin_check" (mips_index_set_param):
 *   0  off (an fp32-exact index then always runs the three-segment scan);
 *   1  (default) CERTIFY: flagged queries are settled EXACTLY -- one pass over the stored rows per 16 flagged queries (8 on the
 *      e4m3 indexes) computes canonical scores by brute force (csrc/resolve_kernels.hpp: of every row whose MFMA score comes
 *      within the error bound of the current k-th key; on the e4m3 indexes, and with "resolve" = 2, of every row) and the rows
 *      reaching the current k-th result are ranked over it.  Searches into HOST buffers synchronise anyway: they read the flag count first and skip the pass
 *      when it is 0.  Searches with DEVICE outputs never synchronise: flag list and count live on the device, the passes
 *      are enqueued behind the first scan and leave at once when nothing is flagged (a few microseconds), graph-capturable;
 *      split-tail searches (mips_search_split) run them on their tail stream.  The one-launch kernel (mips_search_fused,
 *      small searches) hands its flags to the same pass.  A search that flags more than "resolve_budget" queries
 *      (default 1024 = 128 passes): host buffers -> re-scan of the flagged queries with the widest lists (K' = 32; 16 on
 *      fp8) as in the first version; device outputs -> the first results stand, counted `unresolved` -- unless the first
 *      scan was an OPTIMISTIC one (two-stage fp32 search, pools of 32 out of sub-lists: candidates selected by bf16 scores
 *      or short lists), whose results may not stand uncertified: a stream-ordered re-scan with true K' = 32 lists (the
 *      three-segment scan on an fp32-exact index), sized on the device, runs exactly then.
 *      (Rows of more than 1024 columns and "resolve" = 0: the re-scan with the widest lists instead of the exact pass.)
 *   2  the same, and device-output searches synchronise as well to read the counts (mips_index_margin_stats is then free);
 *   3  the stream-ordered form of 1, explicitly (kept for callers of the round-2 library: identical to 1 for device
 *      outputs);
 *   4  count only: device-output searches flag and count on the device and do nothing about it (the round-2 default;
 *      an fp32-exact index then cannot use the scans that rest on the certificate); host-buffer searches still certify.
 * Outputs of the LAST search on the index: flagged = queries flagged by the first pass (-1: only counted on the
 * device and synchronize == 0), rescanned = queries settled exactly (or re-scanned; 0 for a search over its budget whose first
 * results stand), unresolved = queries left with their first result: more than 64 rows tie with the k-th result exactly, or the
 * search flagged more than it resolves.  What an unresolved first result is worth depends on the first scan: true K'-entry
 * lists of bf16 / e4m3 products accumulated in fp32 are exact in practice on tie-free data (the MFMA error observed is
 * ~sqrt(d) 2^-24, two orders of magnitude below the bound); the optimistic scans never leave one behind (mode 1 above).
 * With synchronize != 0 the call also waits for the tail stream of a split-tail search.  Nothing in the reference corresponds (faiss IndexFlat computes its scores in fp32 as well and offers no
 * certificate). */
int mips_index_margin_stats(mips_index_t* index, int64_t* flagged, int64_t* rescanned, int64_t* unresolved,
                            int synchronize, void* hip_stream);

/* Name of the scan-kernel instance the last mips_search on this index dispatched to, in the form rocprofv3 prints
 * it (e.g. "mips::scan_kernel_v4<6, 24, 2, 0, false, 1>"); "" before the first search.  bench.py's roofline.kernel. */
const char* mips_index_last_kernel(const mips_index_t* index);

/* Timing hook used by bench.py.  reset != 0 opens a measurement window: from then on every
 * mips_search records a HIP event pair around its fused scan kernel on the search stream (at most
 * 128 pairs; recording stops when the window is full -- an event pair costs ~11 us of stream time,
 * which is why it is off outside a window).  Returns the summed duration in ms and the number of
 * scan launches recorded in the current window; the stream must have been synchronised. */
int mips_scan_timing(mips_index_t* index, float* out_sum_ms, int* out_count, int reset);

#ifdef __cplusplus
}
#endif
#endif /* MIPS_HIP_H */
