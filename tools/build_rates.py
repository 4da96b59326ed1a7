"""Index build / persistence rates (GPU box), SURVEY.md section 8(f) row 2: how fast documents get INTO the index.
    python tools/build_rates.py [--rows 1048576 --dim 768]
Paths: fp32 rows already on the device (what an encoder produces), fp32 rows in host memory (the reference's
NumPy batches), the on-device build steps of Mips.build_index (max-norm, normalise), save() / load()."""
import argparse, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1 << 20)
ap.add_argument("--dim", type=int, default=768)
a = ap.parse_args()
n, d = a.rows, a.dim
out = {"rows": n, "dim": d}


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best

x_dev = ram.synth_fill(n, d, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS, dtype="f32")
gb32 = n * d * 4 / 1e9
for dtype in ("bf16", "fp8_e4m3", "f32"):
    def build():
        ix = ram.MipsIndex(d, dtype=dtype); ix.reserve(n); ix.add(x_dev)
    t = timed(build)
    out[f"add_device_f32_to_{dtype}"] = {"s": t, "rows_per_s": n / t, "input_GB_per_s": gb32 / t}
x_host = x_dev.cpu().numpy()
def build_host():
    ix = ram.MipsIndex(d); ix.reserve(n); ix.add(x_host)
t = timed(build_host, reps=2)
out["add_host_f32_to_bf16"] = {"s": t, "rows_per_s": n / t, "input_GB_per_s": gb32 / t}
def build_host_batches():  # the reference's add loop hands over 1000-row batches (HF datasets search.py)
    ix = ram.MipsIndex(d)
    for r0 in range(0, n, 1000):
        ix.add(x_host[r0:r0 + 1000])
t = timed(build_host_batches, reps=1)
out["add_host_f32_1000_row_batches"] = {"s": t, "rows_per_s": n / t, "input_GB_per_s": gb32 / t}
t = timed(lambda: ram.rows_max_sumsq(x_dev)); out["max_norm_device"] = {"s": t, "GB_per_s": gb32 / t}
y = x_dev.clone()
t = timed(lambda: ram.l2_normalize_(y)); out["l2_normalize_device"] = {"s": t, "GB_per_s": 2 * gb32 / t}
for dtype in ("bf16", "fp8_e4m3"):
    il2 = ram.MipsIndex(d, metric=ram.METRIC_L2, dtype=dtype); il2.add(x_dev); torch.cuda.synchronize()
    t0 = time.perf_counter(); il2.phi(); t = time.perf_counter() - t0   # first call computes max |x|^2 over the stored rows
    out[f"phi_{dtype}"] = {"s": t, "GB_per_s": n * d * (2 if dtype == "bf16" else 1) / 1e9 / t}
ix = ram.MipsIndex(d); ix.add(x_dev)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    t = timed(lambda: ix.save(tmp), reps=1); out["save_bf16"] = {"s": t, "GB_per_s": n * d * 2 / 1e9 / t}
    t = timed(lambda: ram.MipsIndex.load(tmp), reps=2); out["load_bf16"] = {"s": t, "GB_per_s": n * d * 2 / 1e9 / t}
print(json.dumps(out, indent=1))
