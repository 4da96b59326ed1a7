"""bf16 index, 8 <= k <= 13: optimistic pools (16x16x32 kernel + margin check) against true K' = 16 lists (GPU box).
Certifying calls only ("margin_check" = 2 on device tensors: the call synchronises to read the flag count).
    python tools/k_rate.py [--rows 1048576 --dim 768 --queries 4096 --k 10]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=4096)
ap.add_argument("--k", type=int, default=10)
a = ap.parse_args()
ix = ram.MipsIndex(a.dim)
ix.add_synthetic(a.rows, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
q = ram.synth_fill(a.queries, a.dim, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS)
ix.set_param("margin_check", 2)
outs = {}
for mode in (0, 1):
    ix.set_param("optimistic", mode)
    ix.search(q, a.k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        outs[mode] = ix.search(q, a.k)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    print(json.dumps({"optimistic": mode, "rows": a.rows, "dim": a.dim, "queries": a.queries, "k": a.k, "ms_per_call": t * 1e3,
                      "queries_per_s": a.queries / t, "kernel": ix.last_kernel, "margin": ix.margin_stats()}), flush=True)
print(json.dumps({"identical_results": bool(torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0]))}))
