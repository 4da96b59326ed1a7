"""A/B harness (GPU box): times scan-kernel launch variants interleaved in ONE process.
usage: python tools/ab.py "qgroups=1" "qgroups=4" "qgroups=4,nsplit=32" ... [--rows N --queries Q --rounds R]"""
import argparse, sys, os
os.environ["MIPS_HIP_EXPERIMENTAL"] = "1"  # the library build that contains the `sub` instances (tools/_build/)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+")
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--queries", type=int, default=4096)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
ix = ram.MipsIndex(a.dim, dtype=a.dtype)
ix.add_synthetic(a.rows, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
q = ram.synth_fill(a.queries, a.dim, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS)
ref = None
res = {v: [] for v in a.variants}
for r in range(a.rounds):
    for v in a.variants:
        params = dict(kv.split("=") for kv in v.split(",") if kv)
        for name in ("nsplit", "qgroups", "variant", "sub", "optimistic"):
            try:
                ix.set_param(name, int(params.get(name, 1 if name == "optimistic" else 0)))
            except RuntimeError:
                pass
        s, i = ix.search(q, a.k); torch.cuda.synchronize()
        if ref is None: ref = (s.clone(), i.clone())
        if not any(f"sub={t}" in v for t in (8, 9, 46, 47, 48, 61, 62, 63)) and not (torch.equal(i, ref[1]) and torch.equal(s, ref[0])): print(f"!!! variant {v} changed results", flush=True)
        ix.scan_timing(reset=True)
        for _ in range(a.iters): ix.search(q, a.k)
        torch.cuda.synchronize()
        ms, n = ix.scan_timing()
        res[v].append(ms / n)
fl = 2.0 * a.queries * a.rows * a.dim
for v, t in res.items():
    t = np.array(t)
    print(f"{v:32s} median {np.median(t):8.3f} ms  min {t.min():8.3f} ms  -> {fl / np.median(t) / 1e9:7.1f} TFLOP/s (median)", flush=True)
