"""Randomised parity sweep (GPU box): random index sizes, query counts, dimensions, k, storage dtypes, metrics and
launch knobs against the oracle, bit for bit.  Not part of the pytest suite (run time is open-ended):
    python tools/fuzz.py --seconds 240 [--seed 1]
Every case prints one line; the first mismatch stops the run with the case's parameters."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import retrieval_augmented_mds_amd as ram
from oracle import mips_oracle as orc
from oracle import synth

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--big", action="store_true", help="multi-tile searches (nq > 256) over larger indexes: the 16x16 kernels' regime")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
case = 0
while time.time() < t_end:
    case += 1
    dtype = str(rng.choice(["bf16", "bf16", "bf16", "fp8_e4m3", "fp8_e4m3_docs", "fp8_e4m3_docs", "f32", "f32"]))
    d = int(rng.choice([64, 100, 128, 200, 256, 300, 384, 500, 512, 640, 700, 768, 768, 768, 1000, 1024, 1100]))
    if dtype.startswith("fp8"):
        d = min(d, 1024)
    lattice = bool(rng.random() < 0.3)
    n = int(rng.integers(1, 3000 if lattice else 50000))
    nq = int(rng.integers(1, 40 if lattice else 700))
    if rng.random() < 0.25:
        nq = int(rng.integers(1, 17))                                  # the one-launch kernel's shape
    if a.big and not lattice:
        n, nq = int(rng.integers(50000, 300000)), int(rng.integers(257, 1500))
        d = int(rng.choice([384, 512, 640, 700, 768, 768, 768, 1024]))
    kmax = 13 if dtype == "fp8_e4m3" else ram.MAX_K
    k = int(rng.choice([1, 2, 3, 5, 5, 5, 6, 7, 8, 10, 13, min(kmax, 16), kmax]))
    metric = int(rng.choice([ram.METRIC_IP, ram.METRIC_IP, ram.METRIC_L2]))
    if dtype == "fp8_e4m3_docs":   # e4m3 rows, bf16 queries: neither side representable as it comes
        kind = synth.KIND_GAUSS
        lattice = False
        x = (synth.generate(7000 + case, 0, n, d, kind) * np.float32(1.2345)).astype(np.float32)
        q = (synth.generate(8000 + case, 0, nq, d, kind) * np.float32(0.789)).astype(np.float32)
        xs, qs = synth.round_to_e4m3(x), synth.round_to_bf16(q)
    elif dtype == "f32":
        kind = synth.KIND_GAUSS
        x = (synth.generate(7000 + case, 0, n, d, kind) * np.float32(1.2345)).astype(np.float32)   # not bf16-exact
        q = (synth.generate(8000 + case, 0, nq, d, kind) * np.float32(0.789)).astype(np.float32)
        xs, qs = x, q
    else:
        kind = (synth.KIND_LATTICE_FP8 if dtype == "fp8_e4m3" else synth.KIND_LATTICE) if lattice else synth.KIND_GAUSS
        x = synth.generate(7000 + case, 0, n, d, kind)
        q = synth.generate(8000 + case, 0, nq, d, kind)
        xs, qs = (synth.round_to_e4m3(x), synth.round_to_e4m3(q)) if dtype == "fp8_e4m3" else (x, q)
    ix = ram.MipsIndex(d, metric=metric, dtype=dtype)
    # add in one to three pieces (growth path)
    cuts = sorted(set(int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 3))))) + [n]
    lo = 0
    for hi in cuts:
        if hi > lo:
            ix.add(x[lo:hi])
            lo = hi
    knobs = {}
    if rng.random() < 0.4:
        knobs["nsplit"] = int(rng.choice([8, 16, 24, 40, 64, 128]))
    if rng.random() < 0.4:
        knobs["variant"] = int(rng.choice([1, 3, 4, 5, 6, 7]))   # 5 / 6 / 7 apply at their row pitches only (ignored elsewhere)
    if rng.random() < 0.15:
        knobs["tiny"] = 0
    if rng.random() < 0.3:
        knobs["margin_check"] = int(rng.choice([0, 2, 3, 1]))
    if rng.random() < 0.2:
        knobs["qgroups"] = int(rng.choice([1, 2, 4, 8]))
    if rng.random() < 0.3:
        knobs["optimistic"] = int(rng.choice([0, 1]))                # two-stage fp32 search / optimistic pools (2 would only count on device outputs)
    for name, v in knobs.items():
        ix.set_param(name, v)
    dev = bool(rng.random() < 0.4)                                     # CUDA tensors in / out instead of NumPy
    if dev:
        import torch
        s, i = ix.search(torch.from_numpy(q).cuda(), k)
        ix.check()
        s, i = s.cpu().numpy(), i.cpu().numpy()
    else:
        s, i = ix.search(q, k)
    if lattice:
        es, ei = orc.search_exact_bruteforce(qs, xs, k, metric=metric)
    else:
        es, ei = orc.search_exact(qs, xs, k, metric=metric)
    ok = np.array_equal(i, ei) and np.array_equal(s, es)
    print(f"case {case}: dtype={dtype} n={n} nq={nq} d={d} k={k} metric={metric} lattice={lattice} knobs={knobs} dev={dev} kernel={ix.last_kernel} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        bad = np.where((i != ei).any(axis=1) | (s != es).any(axis=1))[0]
        print("rows differing:", bad[:10], "\n gpu", i[bad[0]], s[bad[0]], "\n ora", ei[bad[0]], es[bad[0]])
        sys.exit(1)
print(f"{case} cases, all bit-exact")
