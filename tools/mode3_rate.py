"""Device-output searches, HIP-event time per call, "margin_check" = 1 (count only) against 3 (stream-ordered certification:
re-scan of flagged queries enqueued without a synchronisation; opens the optimistic / two-stage paths) (GPU box).
    python tools/mode3_rate.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

N, D, Q = 1 << 20, 768, 4096


def ms(ix, q, k, n=10):
    ix.search(q, k); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = ix.search(q, k)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


g = torch.Generator(device="cuda").manual_seed(3)
for dtype, k in (("bf16", 5), ("bf16", 10), ("f32", 5)):
    ix = ram.MipsIndex(D, dtype=dtype)
    if dtype == "f32":
        for r0 in range(0, N, 1 << 18):
            ix.add(torch.randn(1 << 18, D, device="cuda", generator=g))
        q = torch.randn(Q, D, device="cuda", generator=g)
    else:
        ix.add_synthetic(N, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
        q = ram.synth_fill(Q, D, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS)
    outs = {}
    for mode in (1, 3):
        ix.set_param("margin_check", mode)
        t, outs[mode] = ms(ix, q, k)
        print(json.dumps({"index": dtype, "k": k, "rows": N, "queries": Q, "margin_check": mode, "ms_per_call": t, "queries_per_s": Q / t * 1e3,
                          "kernel": ix.last_kernel, "margin": ix.margin_stats()}), flush=True)
    print(json.dumps({"index": dtype, "k": k, "identical": bool(torch.equal(outs[1][1], outs[3][1]) and torch.equal(outs[1][0], outs[3][0]))}), flush=True)
    del ix
