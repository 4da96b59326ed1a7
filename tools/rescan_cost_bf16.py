"""What flagged queries cost on the bf16 index (GPU box): 2^20 x 768, 4096 queries of which NF are built to be flagged (40 copies
of one row whose exact scores differ in the last bf16 ulp of one coordinate; query = that row), device outputs, default margin
mode (stream-ordered certificate).     python tools/rescan_cost_bf16.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

N, D, Q = 1 << 20, 768, 4096
g = torch.Generator(device="cuda").manual_seed(11)
x = torch.randn(N, D, device="cuda", generator=g).bfloat16()
q = torch.randn(Q, D, device="cuda", generator=g).bfloat16()
for nf in (0, 1, 8, 30):
    xx = x.clone()
    qq = q.clone()
    for j in range(nf):
        base = 5000 * (j + 1)
        xx[base + 1: base + 40] = xx[base]
        for t in range(1, 40):                      # one coordinate one bf16 ulp apart per copy
            v = xx[base + t, t].view(torch.int16)
            xx[base + t, t] = (v + 1).view(torch.bfloat16)
        qq[j * 7] = xx[base]
    ix = ram.MipsIndex(D)
    ix.add(xx)
    for mode in (1, 3):
        ix.set_param("margin_check", mode)
        ix.search(qq, 5); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = ix.search(qq, 5)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"planted": nf, "margin_check": mode, "ms_per_call": e0.elapsed_time(e1) / 5, "kernel": ix.last_kernel, "margin": ix.margin_stats()}), flush=True)
    del ix
