"""What flagged queries cost (GPU box): N x D index, 4096 queries of which NF are built to be flagged (40 copies of one row whose
exact scores differ in the last bf16 ulp of one coordinate; query = that row), device outputs, default margin mode (stream-ordered
certificate), with the exact pass behind its MFMA pre-filter ("resolve" = 1, default) and in its plain form ("resolve" = 2).
    python tools/rescan_cost_bf16.py [--rows 1048576 --dim 768 --dtype bf16|f32]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--planted", type=int, nargs="+", default=[0, 1, 8, 30, 200])
a = ap.parse_args()
N, D, Q = a.rows, a.dim, 4096
g = torch.Generator(device="cuda").manual_seed(11)
x = torch.randn(N, D, device="cuda", generator=g).bfloat16()
q = torch.randn(Q, D, device="cuda", generator=g).bfloat16()
for nf in a.planted:
    xx = x.clone()
    qq = q.clone()
    for j in range(nf):
        base = 5000 * (j + 1) % (N - 64)
        xx[base + 1: base + 40] = xx[base]
        for t in range(1, 40):                      # one coordinate one bf16 ulp apart per copy
            v = xx[base + t, t].view(torch.int16)
            xx[base + t, t] = (v + 1).view(torch.bfloat16)
        qq[j * 7] = xx[base]
    ix = ram.MipsIndex(D, dtype=a.dtype)
    ix.add(xx if a.dtype == "bf16" else xx.float())
    qs = qq if a.dtype == "bf16" else qq.float()
    ref = None
    for resolve in (2, 1, 2, 1):   # (twice, alternating: the first timed configuration of an index runs a few per cent faster)
        ix.set_param("resolve", resolve)
        out = ix.search(qs, 5); torch.cuda.synchronize()
        if ref is None:
            ref = out
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = ix.search(qs, 5)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"rows": N, "dim": D, "dtype": a.dtype, "planted": nf, "resolve": resolve, "ms_per_call": e0.elapsed_time(e1) / 5,
                          "kernel": ix.last_kernel, "margin": ix.margin_stats(), "same_results_as_plain_pass": same}), flush=True)
    del ix
