"""Cross-kernel agreement at sizes the oracle cannot reach (GPU box): the same search through every path that may serve it
(true K' lists / optimistic pools, 32x32 / 16x16 kernels, one-stage / two-stage fp32) must return identical bits.  Rare
losses (2 of 4096 queries at 2^20 rows in one case this round) only show at these sizes.
    python tools/cross_check.py --seconds 200 [--seed 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import retrieval_augmented_mds_amd as ram

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
case = 0
while time.time() < t_end:
    case += 1
    dtype = str(rng.choice(["bf16", "bf16", "f32"]))
    d = int(rng.choice([384, 512, 640, 768, 768, 704, 304, 1024, 1000]))
    n = int(rng.integers(200000, 1 << 20)) if dtype == "bf16" else int(rng.integers(100000, 400000))
    if dtype == "bf16" and d > 768 and rng.random() < 0.4:
        n = int(rng.integers(1 << 21, (1 << 21) + 300000))           # row pitch 1024, large index: the pool of 16 for k <= 5
    nq = int(rng.choice([300, 1000, 2048, 4096]))
    k = int(rng.choice([5, 6, 8, 10, 13])) if dtype == "bf16" else int(rng.choice([5, 6, 10]))
    ix = ram.MipsIndex(d, dtype=dtype)
    if dtype == "f32":                                              # rows that are NOT bf16-representable
        g = torch.Generator(device="cuda").manual_seed(1000 + case)
        for r0 in range(0, n, 1 << 17):
            ix.add(torch.randn(min(1 << 17, n - r0), d, device="cuda", generator=g))
    else:
        ix.add_synthetic(n, 0, 1000 + case, ram.SYNTH_GAUSS)
    q = ram.synth_fill(nq, d, 0, 2000 + case, ram.SYNTH_GAUSS)
    if dtype == "f32":
        q = (q.float() * 1.2345).contiguous()
    ix.set_param("margin_check", 2)
    outs = []
    if k >= 8 or dtype == "f32":
        paths = [("optimistic", 0), ("optimistic", 1)]
    elif d > 768:                                                   # pitch 1024: 4-wave kernel, K-split pairs, automatic (+ its pool of 16)
        paths = [("variant", 3), ("variant", 7), ("variant", 0), ("optimistic", 0)]
    else:
        paths = [("variant", 3), ("variant", 4), ("variant", 0)]
    paths.append(("resolve", 2))                                    # the plain exact pass instead of the MFMA-filtered one
    names = []
    for name, v in paths:
        ix.set_param(name, v)
        s, i = ix.search(q, k)
        st = ix.margin_stats()
        outs.append((s, i))
        names.append(f"{name}={v}:{ix.last_kernel}:{st['flagged']}/{st['unresolved']}")
        ix.set_param(name, 1 if name in ("optimistic", "resolve") else 0)
    ok = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    print(f"case {case}: dtype={dtype} n={n} nq={nq} d={d} k={k} {' | '.join(names)} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        for o in outs[1:]:
            bad = ((outs[0][1] != o[1]).any(dim=1) | (outs[0][0] != o[0]).any(dim=1)).nonzero().flatten()
            print(" rows differing:", bad[:8].tolist())
        sys.exit(1)
    del ix
print(f"{case} cases, all paths agree")
