"""Q sweep / size sweep on one GPU (GPU box): scan-kernel time and whole-call time per configuration."""
import argparse, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--queries", type=int, nargs="+", default=[8, 64, 256, 1024, 4096])
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
ix = ram.MipsIndex(a.dim, dtype=a.dtype)
esz = 1 if a.dtype != "bf16" else 2
t0 = time.perf_counter()
ix.reserve(a.rows)
ix.add_synthetic(a.rows, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
torch.cuda.synchronize()
print(f"index {a.rows}x{a.dim}: generated in {time.perf_counter() - t0:.2f} s", flush=True)
for nq in a.queries:
    q = ram.synth_fill(nq, a.dim, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS)
    for _ in range(2): ix.search(q, a.k)
    torch.cuda.synchronize(); ix.scan_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(a.iters): s, i = ix.search(q, a.k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters
    ms, n = ix.scan_timing(); scan = ms / n * 1e-3
    fl = 2.0 * nq * a.rows * a.dim; by = a.rows * a.dim * esz + nq * a.dim * esz + nq * a.k * 12.0
    print(json.dumps({"dtype": a.dtype, "k": a.k, "rows": a.rows, "dim": a.dim, "Q": nq, "scan_ms": scan * 1e3, "call_ms": wall * 1e3, "qps": nq / wall,
                      "tflops": fl / scan / 1e12, "mfma_frac": fl / scan / (2.5e15 if esz == 2 else 5e15), "hbm_gbs": by / scan / 1e9, "hbm_frac": by / scan / 8e12}), flush=True)
