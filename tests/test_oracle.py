"""CPU tests: the oracle (oracle/) against the golden vectors produced by the REAL
reference functions (tests/golden/make_goldens.py), and the oracle's two score
definitions against each other."""
import os

import numpy as np
import pytest
import torch

from oracle import mips_oracle as orc
from oracle import synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_generator_pinned_by_golden_slices(golden_dir):
    g = _load(golden_dir, "g1_g2_inner_product.npz")
    docs = synth.generate(int(g["seed_docs"]), 0, 64, int(g["d"]), int(g["kind"]))
    qs = synth.generate(int(g["seed_queries"]), 0, int(g["nq"]), int(g["d"]), int(g["kind"]))
    assert np.array_equal(docs, g["docs_head"])
    assert np.array_equal(qs, g["queries"])
    # sub-blocks are position independent
    assert np.array_equal(synth.generate(int(g["seed_docs"]), 10, 20, int(g["d"]), int(g["kind"])),
                          g["docs_head"][10:30])


def test_generator_values_are_bf16_exact_and_distributed():
    for kind in (synth.KIND_LATTICE, synth.KIND_GAUSS, synth.KIND_LATTICE_FP8):
        x = synth.generate(3, 5, 512, 768, kind)
        assert x.dtype == np.float32
        assert np.array_equal(synth.round_to_bf16(x), x)
        assert np.array_equal(synth.bf16_bits_to_f32(synth.bf16_bits(x)), x)
    lat = synth.generate(3, 0, 256, 768, synth.KIND_LATTICE) * 64
    assert lat.min() >= -127 and lat.max() <= 127 and np.array_equal(lat, np.round(lat))
    ga = synth.generate(3, 0, 2048, 768, synth.KIND_GAUSS)
    assert abs(ga.mean()) < 5e-3 and abs(ga.std() - 1.0) < 5e-3


def test_inner_product_matches_reference_goldens(golden_dir):
    g = _load(golden_dir, "g1_g2_inner_product.npz")
    y = synth.generate(int(g["seed_docs"]), 0, int(g["n"]), int(g["d"]), int(g["kind"]))
    x = g["queries"]
    k = int(g["k"])
    s, i = orc.inner_product(x, y, k=k, normalize=False)
    assert s.dtype == np.float32 and i.dtype == np.int64 and s.shape == (8, k)
    assert np.array_equal(i, g["indices_raw"])
    np.testing.assert_allclose(s, g["scores_raw"], rtol=2e-6)
    s, i = orc.inner_product(x, y, k=k, normalize=True)
    assert np.array_equal(i, g["indices_norm"])
    np.testing.assert_allclose(s, g["scores_norm"], rtol=2e-6)


def test_inner_product_matches_reference_goldens_f32(golden_dir):
    g = _load(golden_dir, "g1b_inner_product_f32.npz")
    k = int(g["k"])
    s, i = orc.inner_product(g["x"], g["y"], k=k, normalize=True)
    assert np.array_equal(i, g["indices_norm"])
    np.testing.assert_allclose(s, g["scores_norm"], rtol=2e-6, atol=1e-7)
    s, i = orc.inner_product(g["x"], g["y"], k=k, normalize=False)
    assert np.array_equal(i, g["indices_raw"])
    np.testing.assert_allclose(s, g["scores_raw"], rtol=2e-6, atol=1e-6)


def test_canonical_search_agrees_with_reference_indices(golden_dir):
    """The build's canonical definition (fp64 sequential, lowest index on ties) returns the
    reference's indices on the tie-free golden inputs; scores agree to fp32 rounding."""
    g = _load(golden_dir, "g1_g2_inner_product.npz")
    assert float(g["min_top6_gap_fp64"]) > 1e-3          # fixture is tie-free by a wide margin
    y = synth.generate(int(g["seed_docs"]), 0, int(g["n"]), int(g["d"]), int(g["kind"]))
    x = g["queries"]
    s, i = orc.search_exact(x, y, int(g["k"]))
    assert np.array_equal(i, g["indices_raw"])
    np.testing.assert_allclose(s, g["scores_raw"], rtol=1e-5)
    s2, i2 = orc.search_exact_bruteforce(x[:2], y[:3000], 5)
    s3, i3 = orc.search_exact(x[:2], y[:3000], 5)
    assert np.array_equal(i2, i3) and np.array_equal(s2, s3)


def test_canonical_pairs_c_equals_numpy_cumsum():
    x = synth.generate(11, 0, 300, 768, synth.KIND_GAUSS)
    q = synth.generate(12, 0, 5, 768, synth.KIND_GAUSS)
    cand = np.arange(15).reshape(5, 3) * 7
    assert np.array_equal(orc.canonical_pairs(q, x, cand), orc.canonical_pairs_numpy(q, x, cand))


def test_lattice_scores_are_order_independent():
    """Lattice products/sums are exact in fp32: literal fp32 matmul == canonical fp64."""
    x = synth.generate(5, 0, 4096, 768, synth.KIND_LATTICE)
    q = synth.generate(6, 0, 4, 768, synth.KIND_LATTICE)
    dense32 = q @ x.T
    canon = orc.canonical_pairs(q, x, np.tile(np.arange(4096), (4, 1)))
    assert np.array_equal(dense32.astype(np.float64), canon)


def test_ties_go_to_lowest_index():
    x = np.zeros((10, 8), dtype=np.float32)
    x[[2, 5, 7], 0] = 1.0          # three identical best docs
    x[9, 0] = 2.0
    q = np.zeros((1, 8), dtype=np.float32)
    q[0, 0] = 1.0
    s, i = orc.search_exact_bruteforce(q, x, 4)
    assert i.tolist() == [[9, 2, 5, 7]] and s.tolist() == [[2.0, 1.0, 1.0, 1.0]]
    s, i = orc.search_exact_bruteforce(q, x, 6)
    assert i.tolist() == [[9, 2, 5, 7, 0, 1]]


def test_padding_when_k_exceeds_ntotal():
    x = synth.generate(1, 0, 3, 16, synth.KIND_LATTICE)
    q = synth.generate(2, 0, 2, 16, synth.KIND_LATTICE)
    for fn in (orc.search_exact, orc.search_exact_bruteforce):
        s, i = fn(q, x, 5)
        assert (i[:, 3:] == -1).all() and np.isneginf(s[:, 3:]).all() and (i[:, :3] >= 0).all()
        s, i = fn(q, x, 5, metric=orc.METRIC_L2)
        assert (i[:, 3:] == -1).all() and np.isposinf(s[:, 3:]).all()


def test_augmentation_matches_reference_goldens(golden_dir):
    g = _load(golden_dir, "g3_augment.npz")
    xb = synth.generate(int(g["seed_b"]), 0, int(g["n"]), int(g["d"]), int(g["kind"]))
    xq = synth.generate(int(g["seed_q"]), 0, int(g["nq"]), int(g["d"]), int(g["kind"]))
    assert bool(g["body_equal"])
    phi = orc.get_phi(xb)
    assert phi == g["phi"]
    ab = orc.augment_xb(xb)
    assert str(ab.dtype) == str(g["aug_b_dtype"]) and tuple(ab.shape) == tuple(g["aug_b_shape"])
    assert np.array_equal(ab[:, -1], g["extracol_b"]) and np.array_equal(ab[:, :-1], xb)
    assert np.array_equal(orc.augment_xb(xb, phi=np.float32(phi * 1.5))[:, -1], g["extracol_b_phi15"])
    aq = orc.augment_xq(xq)
    assert str(aq.dtype) == str(g["aug_q_dtype"]) and tuple(aq.shape) == tuple(g["aug_q_shape"])
    assert np.array_equal(aq[:, -1], g["aug_q_lastcol"]) and np.array_equal(aq[:, :-1], xq)


def test_retriever_metrics_match_reference_goldens(golden_dir):
    g = _load(golden_dir, "g4_retriever_metrics.npz")
    for n in range(int(g["ncases"])):
        out = orc.retriever_metrics(torch.tensor(g[f"pred{n}"]), torch.tensor(g[f"counts{n}"]))
        got = np.array([out["recall"], out["reciprocal_rank"], out["average_precision"]])
        np.testing.assert_allclose(got, g[f"out{n}"], rtol=1e-6)
    # the documented quirk: a hit at rank 0 scores reciprocal rank 0
    out = orc.retriever_metrics(torch.tensor([[1.0, 0.0]]), torch.tensor([1]))
    assert out["reciprocal_rank"] == 0.0


def test_ip_equals_augmented_l2_ordering(golden_dir):
    g = _load(golden_dir, "g5_ip_equals_aug_l2.npz")
    assert np.array_equal(g["ip_indices"], g["l2_indices"])
    yb = synth.generate(int(g["seed_b"]), 0, int(g["n"]), int(g["d"]), int(g["kind"]))
    xq = synth.generate(int(g["seed_q"]), 0, int(g["nq"]), int(g["d"]), int(g["kind"]))
    s_ip, i_ip = orc.search_exact(xq, yb, int(g["k"]), metric=orc.METRIC_INNER_PRODUCT)
    s_l2, i_l2 = orc.search_exact(xq, yb, int(g["k"]), metric=orc.METRIC_L2)
    assert np.array_equal(i_ip, g["ip_indices"]) and np.array_equal(i_l2, g["l2_indices"])
    np.testing.assert_allclose(s_l2, g["l2_dist"], rtol=1e-5)
    assert (np.diff(s_l2, axis=1) >= 0).all() and (np.diff(s_ip, axis=1) <= 0).all()


def test_l2_normalization_in_place_and_zero_rows():
    x = synth.generate(4, 0, 8, 32, synth.KIND_GAUSS).copy()
    x[3] = 0
    y = orc.l2_normalization(x)
    assert y is x
    n = np.linalg.norm(x, axis=1)
    assert np.allclose(np.delete(n, 3), 1.0, atol=1e-6) and n[3] == 0


def test_prepare_query_modes():
    q = synth.generate(4, 0, 4, 16, synth.KIND_GAUSS)
    out = orc.prepare_query(q.copy(), normalize=True, metric_type=orc.METRIC_INNER_PRODUCT)
    assert out.shape == (4, 16) and np.allclose(np.linalg.norm(out, axis=1), 1, atol=1e-6)
    out = orc.prepare_query(q.copy(), normalize=True, metric_type=orc.METRIC_L2)
    assert out.shape == (4, 17) and np.array_equal(out[:, :16], q) and (out[:, 16] == 0).all()
    out = orc.prepare_query(np.asfortranarray(q), normalize=False, metric_type=orc.METRIC_INNER_PRODUCT)
    assert out.flags.c_contiguous and out.dtype == np.float32


def test_ignore_filter_semantics():
    """G6 (hand-derived from mips.py:388-398): k+1 fetched, equal id dropped, cut to k."""
    idx = np.array([[4, 9, 2, 7], [1, 3, 5, 8]])
    sc = np.array([[.9, .8, .7, .6], [.5, .4, .3, .2]], dtype=np.float32)
    s, i = orc.filter_ignore(sc, idx, [9, 6], k=3)
    assert i == [[4, 2, 7], [1, 3, 5]]
    assert np.allclose(s[0], [.9, .7, .6]) and np.allclose(s[1], [.5, .4, .3])
    calls = []

    def fake(q, k):
        calls.append(k)
        return sc[:, :k], idx[:, :k]

    orc.mips_search(fake, None, ignore_indexes=[9, 6], k=3)
    orc.mips_search(fake, None, ignore_indexes=None, k=3)
    assert calls == [4, 3]


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_merge_equals_unsharded(world):
    n = 1003
    x = synth.generate(9, 0, n, 64, synth.KIND_LATTICE)       # lattice: ties across shards
    q = synth.generate(10, 0, 7, 64, synth.KIND_LATTICE)
    ref_s, ref_i = orc.search_exact_bruteforce(q, x, 6)
    ps, pi = [], []
    covered = 0
    for r in range(world):
        lo, hi = orc.shard_bounds(n, world, r)
        covered += hi - lo
        s, i = orc.search_exact_bruteforce(q, x[lo:hi], 6, idx_offset=lo)
        ps.append(s)
        pi.append(i)
    assert covered == n
    s, i = orc.merge_topk(ps, pi, 6)
    assert np.array_equal(i, ref_i) and np.array_equal(s, ref_s)


def test_cosine_rescore_shape_and_value():
    q = torch.randn(3, 1, 16)
    m = torch.randn(3, 4, 16)
    out = orc.cosine_rescore(q, m)
    assert out.shape == (3, 4)
    exp = torch.nn.functional.cosine_similarity(q.expand(-1, 4, -1), m, dim=2)
    assert torch.allclose(out, exp, atol=1e-6)
