"""fp32-exact index: two-stage search ("f32_fast" = 2) against the one-stage three-segment scan ("f32_fast" = 0), device-resident
queries, HIP-event time per call (GPU box).   python tools/f32_rate.py [--rows 1048576 --dim 768 --queries 4096]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, default=4096)
ap.add_argument("--k", type=int, default=5)
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(7)
ix = ram.MipsIndex(a.dim, dtype="f32")
for r0 in range(0, a.rows, 1 << 18):
    ix.add(torch.randn(min(1 << 18, a.rows - r0), a.dim, device="cuda", generator=g))
q = torch.randn(a.queries, a.dim, device="cuda", generator=g)


def ms(n=5):
    ix.search(q, a.k); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = ix.search(q, a.k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


res = {}
for mode in (0, 2):
    ix.set_param("f32_fast", mode)
    t, out = ms()
    res[mode] = out
    st = ix.margin_stats()
    print(json.dumps({"f32_fast": mode, "rows": a.rows, "dim": a.dim, "queries": a.queries, "k": a.k, "ms_per_call": t,
                      "queries_per_s": a.queries / t * 1e3, "kernel": ix.last_kernel, "flagged": st["flagged"]}), flush=True)
same = bool(torch.equal(res[0][1], res[2][1]) and torch.equal(res[0][0], res[2][0]))
print(json.dumps({"identical_results": same}))
# host buffers (NumPy in / out): the path the reference's call takes; certifies (re-scans what stage 1 flags)
import time, numpy as np
qh = q.cpu().numpy()
for mode in (0, 1):
    ix.set_param("f32_fast", mode)
    ix.search(qh, a.k)
    t0 = time.perf_counter()
    for _ in range(3):
        s, i = ix.search(qh, a.k)
    t = (time.perf_counter() - t0) / 3
    print(json.dumps({"host_buffers": True, "f32_fast": mode, "ms_per_call": t * 1e3, "queries_per_s": a.queries / t, "kernel": ix.last_kernel,
                      "margin": ix.margin_stats(), "same_as_one_stage": bool(np.array_equal(i, res[0][1].cpu().numpy()))}), flush=True)
