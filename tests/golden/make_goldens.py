"""Regenerates tests/golden/*.npz from the REAL reference functions.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  The reference module is imported with MagicMock stand-ins for the
seven third-party packages that are absent offline; only
pure functions are called -- no constructor that would fetch from the hub:

    sotasum.mips.inner_product        (mips.py:552-560)
    sotasum.mips.get_phi / augment_xb / augment_xq   (mips.py:55-70)
    sotasum.pretrain.retriever_metrics  (pretrain.py:69-85, via sotasum.mips)

What is stored is DATA: seeds, small input slices that pin the synthetic
generator, and the reference's outputs.  No reference source text is stored.

    python tests/golden/make_goldens.py
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402

MISSING = ("faiss", "adapters", "pytorch_lightning", "deepspeed", "mlflow", "pymsteams", "evaluate")


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)


def load_reference():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference")
    import faiss  # the stub

    faiss.METRIC_INNER_PRODUCT, faiss.METRIC_L2 = 0, 1
    return importlib.import_module("sotasum.mips")


def main():
    import torch

    ref = load_reference()
    d = 768

    # G1 / G2: inner_product on the cfg-1 shape (10 000 x 768, 8 queries, k = 5), Gaussian-bf16 data
    y = synth.generate(synth.SEED_DOCS, 0, 10000, d, synth.KIND_GAUSS)
    x = synth.generate(synth.SEED_QUERIES, 0, 8, d, synth.KIND_GAUSS)
    s1, i1 = ref.inner_product(x, y, k=5, normalize=False)
    s2, i2 = ref.inner_product(x, y, k=5, normalize=True)
    # gap between 5th and 6th best (fp64) -> documents that the fixture is tie-free
    full = np.sort(x.astype(np.float64) @ y.astype(np.float64).T, axis=1)[:, ::-1]
    gap = np.min(full[:, :6][:, :-1] - full[:, 1:6])
    np.savez(
        os.path.join(HERE, "g1_g2_inner_product.npz"),
        seed_docs=synth.SEED_DOCS, seed_queries=synth.SEED_QUERIES, n=10000, d=d, nq=8, k=5,
        kind=synth.KIND_GAUSS,
        docs_head=y[:64], queries=x,                      # pins the generator
        scores_raw=s1, indices_raw=i1, scores_norm=s2, indices_norm=i2,
        min_top6_gap_fp64=gap,
    )
    print("G1/G2", s1.dtype, i1.dtype, s1.shape, "min top-6 gap", gap)

    # same on a NON-bf16 fp32 input (plain NumPy RNG, stored in full: 2000 x 64)
    rng = np.random.default_rng(1234)
    yf = rng.standard_normal((2000, 64)).astype(np.float32)
    xf = rng.standard_normal((6, 64)).astype(np.float32)
    s3, i3 = ref.inner_product(xf, yf, k=10, normalize=True)
    s4, i4 = ref.inner_product(xf, yf, k=10, normalize=False)
    np.savez(os.path.join(HERE, "g1b_inner_product_f32.npz"), x=xf, y=yf, k=10,
             scores_norm=s3, indices_norm=i3, scores_raw=s4, indices_raw=i4)

    # G3: augmentation on [256, 768]
    xb = synth.generate(7, 0, 256, d, synth.KIND_GAUSS)
    xq = synth.generate(8, 0, 16, d, synth.KIND_GAUSS)
    phi = ref.get_phi(xb)
    aug_b = ref.augment_xb(xb)
    aug_b_phi = ref.augment_xb(xb, phi=np.float32(phi * 1.5))
    aug_q = ref.augment_xq(xq)
    np.savez(os.path.join(HERE, "g3_augment.npz"), seed_b=7, seed_q=8, n=256, nq=16, d=d,
             kind=synth.KIND_GAUSS, phi=phi, extracol_b=aug_b[:, -1], aug_b_dtype=str(aug_b.dtype),
             extracol_b_phi15=aug_b_phi[:, -1], aug_q_lastcol=aug_q[:, -1],
             aug_q_dtype=str(aug_q.dtype), aug_b_shape=aug_b.shape, aug_q_shape=aug_q.shape,
             body_equal=bool((aug_b[:, :-1] == xb).all() and (aug_q[:, :-1] == xq).all()))
    print("G3 phi", phi, aug_b.dtype, aug_b.shape)

    # G4: retriever_metrics incl. the rank-0-hit quirk
    cases = [
        ([[0, 1, 0, 0, 1], [0, 0, 0, 0, 0], [0, 0, 1, 0, 0]], [2, 3, 1]),
        ([[1, 0, 0, 0, 0], [1, 1, 1, 1, 1]], [1, 5]),           # rank-0 hits: 1/0 -> inf -> 0
        ([[0, 0, 0, 0, 1], [0, 1, 1, 0, 0], [0, 0, 0, 1, 0], [0, 1, 0, 1, 0]], [4, 2, 1, 2]),
    ]
    g4 = {}
    for n, (pred, counts) in enumerate(cases):
        out = ref.retriever_metrics(torch.tensor(pred).float(), torch.tensor(counts))
        g4[f"pred{n}"] = np.array(pred, dtype=np.float32)
        g4[f"counts{n}"] = np.array(counts, dtype=np.int64)
        g4[f"out{n}"] = np.array([out["recall"], out["reciprocal_rank"], out["average_precision"]])
        print("G4", n, out)
    np.savez(os.path.join(HERE, "g4_retriever_metrics.npz"), ncases=len(cases), **g4)

    # G5: IP == L2-on-augmented ordering (property of mips.py:655-685) evaluated with the
    # reference's own helpers: brute-force L2 on augment_xb / augment_xq vs inner_product
    yb = synth.generate(21, 0, 20000, d, synth.KIND_GAUSS)
    xq5 = synth.generate(22, 0, 8, d, synth.KIND_GAUSS)
    _, ip_idx = ref.inner_product(xq5, yb, k=5, normalize=False)
    ab, aq = ref.augment_xb(yb).astype(np.float64), ref.augment_xq(xq5).astype(np.float64)
    d2 = (aq ** 2).sum(1)[:, None] + (ab ** 2).sum(1)[None, :] - 2.0 * aq @ ab.T
    l2_idx = np.argsort(d2, axis=1)[:, :5]
    np.savez(os.path.join(HERE, "g5_ip_equals_aug_l2.npz"), seed_b=21, seed_q=22, n=20000, nq=8,
             d=d, k=5, kind=synth.KIND_GAUSS, ip_indices=ip_idx, l2_indices=l2_idx,
             l2_dist=np.take_along_axis(d2, l2_idx, axis=1), phi=ref.get_phi(yb))
    print("G5 IP == L2 ordering:", bool((ip_idx == l2_idx).all()))


if __name__ == "__main__":
    main()
