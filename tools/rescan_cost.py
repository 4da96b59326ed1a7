"""What a few flagged queries cost (GPU box): fp32-exact index 2^20 x 768, 4096 queries of which NF are built to be flagged
(40 near-copies of one row each, query = that row: the pool of 32 cannot separate them), two-stage search, device outputs,
stream-ordered re-scan (default margin mode).     python tools/rescan_cost.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

N, D, Q = 1 << 20, 768, 4096
g = torch.Generator(device="cuda").manual_seed(11)
x = torch.randn(N, D, device="cuda", generator=g)
q = torch.randn(Q, D, device="cuda", generator=g)
for nf in (0, 1, 30, 200):
    xx = x.clone()
    qq = q.clone()
    for j in range(nf):
        base = 5000 * (j + 1)
        xx[base + 1: base + 40] = xx[base] * (1.0 + 1e-6 * torch.arange(1, 40, device="cuda").unsqueeze(1))
        qq[j * 7] = xx[base]
    ix = ram.MipsIndex(D, dtype="f32")
    for r0 in range(0, N, 1 << 18):
        ix.add(xx[r0:r0 + (1 << 18)])
    for mode in ("two-stage", "one-stage"):
        ix.set_param("f32_fast", 2 if mode == "two-stage" else 0)
        ix.set_param("margin_check", 3)
        ix.search(qq, 5); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            out = ix.search(qq, 5)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"planted": nf, "mode": mode, "ms_per_call": e0.elapsed_time(e1) / 3, "kernel": ix.last_kernel, "margin": ix.margin_stats()}), flush=True)
    del ix
