"""Recall of the reduced-precision index storages against the fp32-exact index on fp32 data (GPU box).
The synthetic generator of the parity tests emits bf16-exact values; real embeddings are fp32, so a bf16 / e4m3 index
ranks ROUNDED rows.  This reports how often that changes the top-k: recall@k (set overlap with the fp32-exact top-k),
top-1 agreement and the fraction of queries whose whole ordered top-k is identical.
    python tools/recall.py [--rows 1048576 --dim 768 --queries 1024 --k 5]
Two data sets: i.i.d. Gaussian rows (margins ~ 0.1 sigma at rank k) and a clustered set (every row = one of rows/64
centres + 5 % noise: near-duplicates crowd the top of every list)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=1024)
ap.add_argument("--k", type=int, default=5)
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(1234)


def make(kind):
    if kind == "gauss":
        x = torch.randn(a.rows, a.dim, device="cuda", generator=g)
        q = torch.randn(a.queries, a.dim, device="cuda", generator=g)
    else:
        c = torch.randn(a.rows // 64, a.dim, device="cuda", generator=g)
        x = c.repeat_interleave(64, 0) + 0.05 * torch.randn(a.rows, a.dim, device="cuda", generator=g)
        pick = torch.randint(0, a.rows // 64, (a.queries,), device="cuda", generator=g)
        q = c[pick] + 0.05 * torch.randn(a.queries, a.dim, device="cuda", generator=g)
    return x, q


for kind in ("gauss", "clustered"):
    x, q = make(kind)
    for normalize in (False, True):
        if normalize:
            x = x / x.norm(dim=1, keepdim=True)
            q = q / q.norm(dim=1, keepdim=True)
        ref = ram.MipsIndex(a.dim, dtype="f32")
        ref.add(x)
        _, ti = ref.search(q, a.k)
        ref.check()
        ti = ti.cpu().numpy()
        del ref
        for dtype in ("bf16", "fp8_e4m3", "fp8_e4m3_docs"):    # (the last: e4m3 rows, bf16 queries)
            ix = ram.MipsIndex(a.dim, dtype=dtype)
            ix.add(x)
            _, gi = ix.search(q, a.k)
            ix.check()
            gi = gi.cpu().numpy()
            del ix
            overlap = np.mean([len(set(ti[r]) & set(gi[r])) / a.k for r in range(a.queries)])
            print(json.dumps({"data": kind, "normalized": normalize, "rows": a.rows, "dim": a.dim, "queries": a.queries, "k": a.k,
                              "index_dtype": dtype, "recall_at_k": float(overlap), "top1_agree": float(np.mean(ti[:, 0] == gi[:, 0])),
                              "ordered_topk_identical": float(np.mean((ti == gi).all(axis=1)))}), flush=True)
    del x, q
