"""TEST INFRASTRUCTURE -- CPU restatement (oracle) of the reference MIPS hot path.

NOT product code.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py may import this module; the product (retrieval-augmented-mds_amd/)
never does and fails loudly when its HIP library is missing.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Parity pinning: tests/golden/*.npz hold outputs of the REAL
reference functions (sotasum.mips.inner_product / augment_xb / augment_xq /
get_phi, sotasum.pretrain.retriever_metrics) produced in the build container by
tests/golden/make_goldens.py; tests/test_oracle.py checks this file against them.
The FAISS boundary itself (faiss-cpu 1.7.4, IndexFlat.search / normalize_L2,
third-party, absent from the container) has no golden vectors in the reference:
at that boundary parity is pinned through the reference's own brute-force
`inner_product` (exact search == IndexFlat for tie-free inputs) and through the
IP == augmented-L2 property of sotasum/mips.py:655-685.

Two score definitions live here:
  * `inner_product`  -- the literal fp32 NumPy restatement (mips.py:552-560).
  * `search_exact`   -- the BUILD's canonical definition that the HIP path must
    match bit for bit:  score = float32( sum_k^{sequential, fp64} q[k]*x[k] ),
    ordered by (score desc, index asc).  Products of bf16 (or fp32) inputs are
    exact in fp64, so fused or unfused multiply-add give the same bits and the
    only ordering freedom -- the summation order -- is fixed to k = 0..d-1.
    On tie-free inputs both definitions return the same indices (tested).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

METRIC_INNER_PRODUCT = 0  # faiss.METRIC_INNER_PRODUCT
METRIC_L2 = 1             # faiss.METRIC_L2

_HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# MIPS -> L2 reduction helpers            sotasum/mips.py:55-70
# (duplicates: retriever_lightning.py:54-68, pretrain.py:41-55)
# --------------------------------------------------------------------------
def get_phi(xb: np.ndarray):
    # mips.py:55-56
    return (xb ** 2).sum(1).max()


def augment_xb(xb: np.ndarray, phi=None) -> np.ndarray:
    # mips.py:59-65
    norms = (xb ** 2).sum(1)
    if phi is None:
        phi = norms.max()
    extracol = np.sqrt(phi - norms)
    return np.hstack((xb, extracol.reshape(-1, 1)))


def augment_xq(xq: np.ndarray) -> np.ndarray:
    # mips.py:68-70
    extracol = np.zeros(len(xq), dtype="float32")
    return np.hstack((xq, extracol.reshape(-1, 1)))


# --------------------------------------------------------------------------
# normalisation / query preparation       sotasum/mips.py:521-525, 368-375
# --------------------------------------------------------------------------
def l2_normalization(x: np.ndarray) -> np.ndarray:
    """mips.py:521-525 -> faiss.normalize_L2 (faiss-cpu 1.7.4, utils/distances.cpp
    fvec_renorm_L2: nr = sum x^2 in fp32; if nr > 0: x *= 1/sqrtf(nr)), IN PLACE."""
    if not x.flags.c_contiguous:
        x = np.asarray(x, order="C")
    nr = np.einsum("ij,ij->i", x, x, dtype=np.float32)
    inv = np.ones_like(nr)
    np.divide(np.float32(1.0), np.sqrt(nr), out=inv, where=nr > 0)
    x *= inv[:, None]
    return x


def prepare_query(query: np.ndarray, normalize: bool, metric_type: int) -> np.ndarray:
    # mips.py:368-375
    if normalize and metric_type == METRIC_INNER_PRODUCT:
        query = l2_normalization(query)
    if metric_type == METRIC_L2:
        query = augment_xq(query)
    if not query.flags.c_contiguous:
        query = np.asarray(query, order="C")
    return query.astype(np.float32)


# --------------------------------------------------------------------------
# brute force                              sotasum/mips.py:552-560 (np_search :527-529)
# --------------------------------------------------------------------------
def inner_product(x: np.ndarray, y: np.ndarray, k: int = 1, normalize: bool = True):
    """mips.py:552-560.  Optional row normalisation of both sides (:554-556),
    dense fp32 score matrix x @ y.T (:557), full descending sort of every row by
    argsort of the negated scores, first k columns (:558), gather (:559)."""
    if x.ndim != 2 or y.ndim != 2:
        raise AssertionError("inner_product expects two matrices")
    if normalize:
        x = x / np.linalg.norm(x, axis=1, keepdims=True)
        y = y / np.linalg.norm(y, axis=1, keepdims=True)
    dense = x @ y.T
    top = np.argsort(-dense, axis=-1)[:, :k]
    return np.take_along_axis(dense, top, axis=1), top


# --------------------------------------------------------------------------
# Mips.search incl. the ignore filter      sotasum/mips.py:382-400
# --------------------------------------------------------------------------
def filter_ignore(scores, indices, ignore_indexes, k: int):
    """mips.py:388-398: k+1 hits were fetched per query; every hit whose id equals
    ignore_indexes[j] is dropped, the remainder is cut to k.  Returns python lists
    of lists (what the reference returns in this branch)."""
    kept_s, kept_i = [], []
    for j in range(len(indices)):
        banned = ignore_indexes[j]
        row_s, row_i = [], []
        for s, i in zip(scores[j], indices[j]):
            if banned != i:
                row_s.append(s)
                row_i.append(i)
        kept_s.append(row_s[:k])
        kept_i.append(row_i[:k])
    return kept_s, kept_i


def mips_search(index_search, queries: np.ndarray, ignore_indexes=None, k: int = 10):
    """mips.py:382-400 with `index_search(q, k) -> (D, I)` standing in for
    Dataset.get_index(name).faiss_index.search."""
    scores, indices = index_search(queries, k + 1 if ignore_indexes is not None else k)
    if ignore_indexes is not None:
        scores, indices = filter_ignore(scores, indices, ignore_indexes, k)
    return scores, indices


# --------------------------------------------------------------------------
# canonical scores (the build's definition) -- C helper for speed
# --------------------------------------------------------------------------
_lib = None


def build_c(force: bool = False) -> str:
    """gcc-compile oracle/mips_oracle.c -> oracle/_build/libmips_oracle.so."""
    src = os.path.join(_HERE, "mips_oracle.c")
    out_dir = os.path.join(_HERE, "_build")
    out = os.path.join(out_dir, "libmips_oracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, src, "-lm"]
        )
    return out


def _c():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build_c())
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.oracle_canon_pairs.argtypes = [f32p, f32p, i64p, ctypes.c_int64, ctypes.c_int64,
                                           ctypes.c_int64, f64p]
        lib.oracle_canon_pairs.restype = None
        lib.oracle_search_exact.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64,
                                            ctypes.c_int64, ctypes.c_int, f32p, i64p]
        lib.oracle_search_exact.restype = None
        lib.oracle_sumsq.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, f64p]
        lib.oracle_sumsq.restype = None
        _lib = lib
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def canonical_pairs(q: np.ndarray, x: np.ndarray, cand: np.ndarray) -> np.ndarray:
    """fp64 sequential dot of q[i] with x[cand[i, j]]  ->  float64 [nq, c]
    (cand < 0 gives -inf)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    cand = np.ascontiguousarray(cand, dtype=np.int64)
    out = np.empty(cand.shape, dtype=np.float64)
    _c().oracle_canon_pairs(_p(q, ctypes.c_float), _p(x, ctypes.c_float), _p(cand, ctypes.c_int64),
                            q.shape[0], cand.shape[1], q.shape[1], _p(out, ctypes.c_double))
    return out


def canonical_pairs_numpy(q, x, cand):
    """Same as canonical_pairs in pure NumPy (cumsum is a sequential fp64 sum)."""
    q64 = np.asarray(q, dtype=np.float64)
    out = np.empty(cand.shape, dtype=np.float64)
    for i in range(cand.shape[0]):
        rows = np.asarray(x[cand[i]], dtype=np.float64)
        out[i] = np.cumsum(rows * q64[i][None, :], axis=1)[:, -1]
    return out


def sumsq_canonical(x: np.ndarray) -> np.ndarray:
    """fp64 sequential sum of squares per row."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.shape[0], dtype=np.float64)
    _c().oracle_sumsq(_p(x, ctypes.c_float), x.shape[0], x.shape[1], _p(out, ctypes.c_double))
    return out


def _order_desc(scores_f32: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """argsort every row by (score desc, idx asc); ids < 0 (padding) go last."""
    big = np.where(idx < 0, np.iinfo(np.int64).max, idx)
    neg = -scores_f32.astype(np.float64)
    return np.stack([np.lexsort((big[r], neg[r])) for r in range(idx.shape[0])])


def search_exact(q: np.ndarray, x: np.ndarray, k: int, metric: int = METRIC_INNER_PRODUCT,
                 idx_offset: int = 0, slack: int = 16, chunk: int = 1 << 18):
    """The build's canonical exact top-k (what IndexFlat.search returns for
    tie-free data, mips.py:383-386), for inputs that are ALREADY rounded to the
    index dtype.  IP: (float32(canonical dot) desc, idx asc).  L2 (metric 1):
    returns float32(|q|^2 + phi - 2*ip) ascending, the squared distance FAISS
    computes on the phi-augmented vectors of mips.py:316-331 -- same ranking.
    Pads with idx -1 / score -inf (+inf for L2) when k > ntotal.

    Candidate generation uses a float64 BLAS matmul (error ~1e-13 relative, far
    below any fp32 score gap) in doc chunks with a running top-(k+slack); the
    candidates are then rescored canonically with the C helper."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    nq, n = q.shape[0], x.shape[0]
    kc = min(n, k + slack)
    out_s = np.full((nq, k), -np.inf if metric == METRIC_INNER_PRODUCT else np.inf, dtype=np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    if nq == 0 or n == 0 or k == 0:
        return out_s, out_i
    q64 = q.astype(np.float64)
    best_s = np.full((nq, 0), 0.0)
    best_i = np.full((nq, 0), 0, dtype=np.int64)
    for c0 in range(0, n, chunk):
        xs = x[c0:c0 + chunk].astype(np.float64)
        s = q64 @ xs.T
        m = min(kc, s.shape[1])
        part = np.argpartition(-s, m - 1, axis=1)[:, :m]
        ps = np.take_along_axis(s, part, axis=1)
        best_s = np.concatenate([best_s, ps], axis=1)
        best_i = np.concatenate([best_i, part + c0], axis=1)
        if best_s.shape[1] > kc:
            sel = np.argpartition(-best_s, kc - 1, axis=1)[:, :kc]
            best_s = np.take_along_axis(best_s, sel, axis=1)
            best_i = np.take_along_axis(best_i, sel, axis=1)
    # ties at the candidate boundary: widen with every doc whose approximate score
    # equals the smallest kept one is unnecessary for tie-free data; lattice data is
    # handled by search_exact_bruteforce (small sizes) instead.
    canon = canonical_pairs(q, x, best_i)                 # fp64
    if metric == METRIC_L2:
        # the L2 result is ordered by ITS float32 value (ascending, index ascending on ties): two documents
        # whose inner products differ can round to the same distance next to |q|^2 + phi
        phi = sumsq_canonical(x).max()
        qn = sumsq_canonical(q)
        val = (qn[:, None] + phi - 2.0 * canon).astype(np.float32)
        order = _order_desc(-val, best_i)
    else:
        val = canon.astype(np.float32)
        order = _order_desc(val, best_i)
    val = np.take_along_axis(val, order, axis=1)[:, :k]
    ids = np.take_along_axis(best_i, order, axis=1)[:, :k]
    kk = val.shape[1]
    out_s[:, :kk] = val
    out_i[:, :kk] = ids + idx_offset
    return out_s, out_i


def search_exact_bruteforce(q, x, k, metric=METRIC_INNER_PRODUCT, idx_offset=0):
    """Canonical exact top-k by full enumeration in C (every pair rescored
    sequentially in fp64).  Tie-safe: use for lattice inputs and small sizes."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    nq, n = q.shape[0], x.shape[0]
    # L2: ranked by the float32 DISTANCE (see search_exact); fetch a few more by inner product, re-rank, cut
    kf = k if metric != METRIC_L2 else min(max(n, k), k + 16)
    out_s = np.empty((nq, kf), dtype=np.float32)
    out_i = np.empty((nq, kf), dtype=np.int64)
    if nq and kf:
        _c().oracle_search_exact(_p(q, ctypes.c_float), nq, _p(x, ctypes.c_float), n, q.shape[1] if q.ndim == 2 else 0,
                                 kf, _p(out_s, ctypes.c_float), _p(out_i, ctypes.c_int64))
    if metric == METRIC_L2:
        phi = sumsq_canonical(x).max() if n else 0.0
        qn = sumsq_canonical(q)
        valid = out_i >= 0
        safe = np.where(valid, out_i, 0)
        ip = canonical_pairs(q, x, safe) if n else np.zeros(out_i.shape)
        d = (qn[:, None] + phi - 2.0 * ip).astype(np.float32)
        d = np.where(valid, d, np.float32(np.inf)).astype(np.float32)
        key_i = np.where(valid, out_i, np.iinfo(np.int64).max)
        order = _order_desc(-d, key_i)
        out_s = np.take_along_axis(d, order, axis=1)[:, :k]
        out_i = np.take_along_axis(out_i, order, axis=1)[:, :k]
    out_i = np.where(out_i >= 0, out_i + idx_offset, -1)
    return out_s, out_i


# --------------------------------------------------------------------------
# row sharding + merge (partition arithmetic follows the contiguous
# chunking idea of mips.py:227-229)
# --------------------------------------------------------------------------
def shard_bounds(n: int, world: int, rank: int):
    per = -(-n // world)  # ceil
    lo = min(n, rank * per)
    hi = min(n, (rank + 1) * per)
    return lo, hi


def merge_topk(parts_s, parts_i, k: int, metric: int = METRIC_INNER_PRODUCT):
    """Global top-k of the union of per-shard top-k lists by (score desc, idx asc)
    (L2: score asc)."""
    s = np.concatenate(parts_s, axis=1)
    i = np.concatenate(parts_i, axis=1)
    key = -s.astype(np.float64) if metric == METRIC_INNER_PRODUCT else s.astype(np.float64)
    big = np.where(i < 0, np.iinfo(np.int64).max, i)
    order = np.stack([np.lexsort((big[r], key[r])) for r in range(s.shape[0])])[:, :k]
    return np.take_along_axis(s, order, axis=1), np.take_along_axis(i, order, axis=1)


# --------------------------------------------------------------------------
# retriever metrics                        sotasum/pretrain.py:69-85
# (duplicate: retriever_lightning.py:71-87).  Keeps the reference's behaviour of
# 1/argmax -> inf -> 0 for a rank-0 hit (latent bug noted).
# --------------------------------------------------------------------------
def retriever_metrics(pred, counts) -> dict:
    """pred: 0/1 hit matrix [B,k] (torch float), counts: [B] number of relevant
    docs.  recall = hits/counts (:70); reciprocal rank = 1/argmax with inf -> 0
    (:72-74); average precision = sum_j (hits up to j / (j+1)) * hit_j / counts
    (:76-77); all averaged over the batch."""
    import torch

    hits = pred.sum(dim=-1)
    first = pred.argmax(dim=-1)
    rr = torch.ones_like(first, dtype=torch.float32) / first
    rr = torch.where(torch.isinf(rr), torch.zeros_like(rr), rr)
    ranks = torch.arange(1, pred.shape[-1] + 1)
    prec_at = pred.cumsum(dim=-1) / ranks
    ap = (prec_at * pred).sum(dim=-1) / counts
    return {
        "recall": (hits / counts).mean().item(),
        "reciprocal_rank": rr.mean().item(),
        "average_precision": ap.mean().item(),
    }


# --------------------------------------------------------------------------
# cosine re-score of the scoring hook      sotasum/retriever_generator.py:158-172
# --------------------------------------------------------------------------
def cosine_rescore(query, mips_cls):
    """query [B,1,d], mips_cls [B,k,d] (torch) -> [B,k]."""
    import torch

    mips_scores = (query @ mips_cls.transpose(1, 2)).squeeze(1)
    query_norms = torch.norm(query, dim=2, keepdim=True)
    mips_norms = torch.norm(mips_cls, dim=2, keepdim=True)
    mips_scores = mips_scores / (query_norms * mips_norms).squeeze(2)
    return mips_scores


def memory_bias(mips_scores, memory_seq_len: int):
    """sotasum/retriever_generator.py:188-192: every hit's score repeated over its memory tokens,
    [B,k] -> [B, k * memory_seq_len]."""
    b = mips_scores.shape[0]
    return mips_scores.unsqueeze(-1).expand(-1, -1, memory_seq_len).reshape(b, -1)


# --------------------------------------------------------------------------
# full-KB eval consumer      sotasum/retriever_lightning.py:312-321, 339-404
# --------------------------------------------------------------------------
def nearest_examples_batch(index_search, columns: dict, queries: np.ndarray, k: int):
    """`Dataset.get_nearest_examples_batch("mips_cls", queries, k)` as the reference calls it
    (retriever_lightning.py:317-321): HF datasets' search.py runs `faiss_index.search(queries, k)`, drops
    ids < 0 per query and returns (scores per query, rows per query as a dict of columns).
    index_search(q, k) -> (scores [nq, k], ids [nq, k])."""
    s, i = index_search(np.ascontiguousarray(queries, dtype=np.float32), k)
    scores, examples = [], []
    for row_s, row_i in zip(s, i):
        keep = [int(t) for t in row_i if t >= 0]
        scores.append(np.asarray(row_s[: len(keep)]))
        examples.append({c: [v[t] for t in keep] for c, v in columns.items()})
    return scores, examples


def full_kb_eval_index(cls: np.ndarray, inner_product: bool):
    """on_validation_start, retriever_lightning.py:372-404: unless `inner_product`, phi = max |x|^2 and the
    `cls` column is replaced by augment_xb(cls, phi); metric = IP or L2.  Returns (column to index, metric)."""
    if inner_product:
        return cls, METRIC_INNER_PRODUCT
    phi = max((cls ** 2).sum(1))
    return augment_xb(cls, phi), METRIC_L2


# --------------------------------------------------------------------------
# in-batch scoring      sotasum/retriever_lightning.py:304-305 (predict), 273-277 (train accuracy)
# --------------------------------------------------------------------------
def in_batch_scores(query_cls, mips_cls, normalize: bool = False):
    """`scores = query_cls @ mips_cls.T; _, i = scores.topk(1)` (:304-305); with normalize the
    `F.normalize(query_cls) @ F.normalize(mips_cls).T` of :273-277.  torch tensors -> (scores [B,B], top1 [B])."""
    import torch
    import torch.nn.functional as F

    if normalize:
        query_cls, mips_cls = F.normalize(query_cls), F.normalize(mips_cls)
    scores = query_cls @ mips_cls.T
    _, i = scores.topk(1)
    return scores, i.view(-1)


# --------------------------------------------------------------------------
# "reference CPU torch path"      the idiom of sotasum/retriever_lightning.py:304-305 scaled to [Q,d] x [N,d]^T
# --------------------------------------------------------------------------
def torch_topk(q, x, k: int, q_chunk: int = 256, x_block: int = 1 << 20):
    """`scores = q @ x.T; scores.topk(k)` in fp32 on the host cores (torch threads as set by the caller).
    [Q, N] fp32 would be 17 GB at BASELINE config 2, so it runs in q_chunk-query chunks over x_block-row document
    blocks with a running top-k merge.  torch CPU tensors -> (scores [Q,k], indices [Q,k]
    int64), descending."""
    import torch

    out_s, out_i = [], []
    for q0 in range(0, q.shape[0], q_chunk):
        qc = q[q0:q0 + q_chunk]
        best_s = best_i = None
        for x0 in range(0, x.shape[0], x_block):
            sc = qc @ x[x0:x0 + x_block].T
            s, i = sc.topk(min(k, sc.shape[1]), dim=1)
            i = i + x0
            if best_s is None:
                best_s, best_i = s, i
            else:
                cs, ci = torch.cat((best_s, s), 1), torch.cat((best_i, i), 1)
                best_s, sel = cs.topk(min(k, cs.shape[1]), dim=1)
                best_i = torch.gather(ci, 1, sel)
        out_s.append(best_s)
        out_i.append(best_i)
    return torch.cat(out_s), torch.cat(out_i)
