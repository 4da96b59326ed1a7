"""CPU tests of the product's host side: the C-ABI library builds, loads and exports every symbol
include/mips_hip.h declares (no compute without a GPU); the Mips facade's host logic against the
oracle (with the oracle standing in for the device index); sharding arithmetic and the packed
all-gather on a 2-rank gloo group."""
import os
import re

import numpy as np
import pytest
import torch

import retrieval_augmented_mds_amd as ram
from oracle import mips_oracle as orc
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    header = open(os.path.join(ROOT, "include", "mips_hip.h")).read()
    declared = set(re.findall(r"\b(mips_[a-z0-9_]+)\s*\(", header))
    declared.discard("mips_hip")
    assert declared, "no declarations parsed"
    lib = ram._lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in mips_hip.h but not exported"
    assert set(ram._lib.EXPORTS) == declared
    assert lib.mips_abi_version() == ram._lib.ABI_VERSION == 1
    assert int(re.search(r"#define MIPS_MAX_K (\d+)", header).group(1)) == ram.MAX_K


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu():
    with pytest.raises(RuntimeError, match="no CPU path"):
        ram.MipsIndex(16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ram.inner_product(np.zeros((1, 8), np.float32), np.zeros((2, 8), np.float32), 1)
    m = ram.Mips(ram.MipsArgs())
    with pytest.raises(RuntimeError):
        m.build_index(np.zeros((4, 8), np.float32))
    with pytest.raises(RuntimeError, match="no index"):
        m.search(np.zeros((1, 8), np.float32), k=1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "retrieval-augmented-mds_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                assert "mips_oracle" not in text, f


class OracleIndex:
    """Stands in for the device index in CPU tests: exact search on bf16-rounded data."""

    def __init__(self, x, metric=0):
        self.x = synth.round_to_bf16(np.asarray(x, dtype=np.float32))
        self.d = self.x.shape[1]
        self.metric = self.metric_type = metric  # faiss.Index.metric_type
        self.calls = []

    @property
    def ntotal(self):
        return len(self.x)

    def search(self, q, k, idx_offset=0):
        self.calls.append((np.asarray(q).shape, k))
        q = synth.round_to_bf16(np.asarray(q, dtype=np.float32))
        return orc.search_exact_bruteforce(q, self.x, k, metric=self.metric, idx_offset=idx_offset)


def _facade(metric=0, normalize=True, n=300, d=32, **kw):
    x = synth.generate(3, 0, n, d, synth.KIND_GAUSS)
    data = {"mips_column": [f"doc {i}" for i in range(n)], "aid": [f"a{i}" for i in range(n)]}
    m = ram.Mips(ram.MipsArgs(mips_metric_type=metric, mips_normalize=normalize, **kw), data=data)
    m.embeddings = ram.KnowledgeBase(dict(data), OracleIndex(x, metric), m.index_name)
    return m, x


def test_helpers_match_oracle():
    xb = synth.generate(7, 0, 64, 48, synth.KIND_GAUSS)
    xq = synth.generate(8, 0, 5, 48, synth.KIND_GAUSS)
    assert ram.get_phi(xb) == orc.get_phi(xb)
    assert np.array_equal(ram.augment_xb(xb), orc.augment_xb(xb))
    assert np.array_equal(ram.augment_xb(xb, phi=np.float32(99.0)), orc.augment_xb(xb, phi=np.float32(99.0)))
    assert np.array_equal(ram.augment_xq(xq), orc.augment_xq(xq))
    assert ram.augment_xb(xb).dtype == np.float32 and ram.augment_xq(xq).dtype == np.float32


def test_retriever_metrics_match_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_retriever_metrics.npz"))
    for n in range(int(g["ncases"])):
        out = ram.retriever_metrics(torch.tensor(g[f"pred{n}"]), torch.tensor(g[f"counts{n}"]))
        got = np.array([out["recall"], out["reciprocal_rank"], out["average_precision"]])
        np.testing.assert_allclose(got, g[f"out{n}"], rtol=1e-6)


@pytest.mark.parametrize("metric,normalize", [(0, True), (0, False), (1, True)])
def test_prepare_query_matches_oracle(metric, normalize):
    m, _ = _facade(metric, normalize)
    q = synth.generate(5, 0, 6, 32, synth.KIND_GAUSS)
    got = m._prepare_query(q.copy())
    exp = orc.prepare_query(q.copy(), normalize, metric)
    assert got.dtype == np.float32 and got.flags.c_contiguous and np.array_equal(got, exp)
    got = m._prepare_query(np.asfortranarray(q.copy()))
    assert np.array_equal(got, exp)


def test_l2_normalization_is_in_place():
    m, _ = _facade()
    q = synth.generate(5, 0, 4, 32, synth.KIND_GAUSS).copy()
    q[2] = 0
    ref = orc.l2_normalization(q.copy())
    out = m.l2_normalization(q)
    assert out is q and np.array_equal(q, ref)


def test_search_and_ignore_filter_match_oracle():
    m, x = _facade(0, False)
    q = synth.generate(5, 0, 6, 32, synth.KIND_GAUSS)
    index = m.embeddings.get_index(m.index_name).faiss_index
    s, i = m.search(q, k=4)
    es, ei = orc.mips_search(index.search, q, None, 4)
    assert np.array_equal(i, ei) and np.array_equal(s, es) and isinstance(s, np.ndarray)
    # ignore: k+1 fetched, equal id dropped, cut to k, lists of lists
    ignore = [int(ei[0][0]), 10 ** 6, int(ei[2][3]), int(ei[3][1]), -5, int(ei[5][0])]
    index.calls.clear()
    s, i = m.search(q, ignore_indexes=ignore, k=4)
    assert index.calls[-1][1] == 5
    es, ei2 = orc.mips_search(index.search, q, ignore, 4)
    assert isinstance(i, list) and all(len(r) == 4 for r in i)
    assert [list(map(int, r)) for r in i] == [list(map(int, r)) for r in ei2]
    assert [list(map(float, r)) for r in s] == [list(map(float, r)) for r in es]
    assert ignore[0] not in i[0] and ignore[2] not in i[2]
    # tensors are accepted for ignore_indexes like in the reference
    s2, i2 = m.search(q, ignore_indexes=torch.tensor(ignore), k=4)
    assert [list(map(int, r)) for r in i2] == [list(map(int, r)) for r in i]


def test_l2_mode_strips_the_augmentation_column():
    m, x = _facade(1, True)
    q = synth.generate(5, 0, 3, 32, synth.KIND_GAUSS)
    pq = m._prepare_query(q.copy())
    assert pq.shape == (3, 33)
    s, i = m.search(pq, k=5)
    es, ei = orc.search_exact_bruteforce(q, x, 5, metric=1)
    assert np.array_equal(i, ei) and np.array_equal(s, es)
    # and equals brute-force L2 on the reference's augmented vectors
    ab, aq = orc.augment_xb(x).astype(np.float64), orc.augment_xq(q).astype(np.float64)
    d2 = ((aq[:, None, :] - ab[None, :, :]) ** 2).sum(-1)
    assert np.array_equal(np.argsort(d2, axis=1)[:, :5], i)
    np.testing.assert_allclose(np.take_along_axis(d2, i, 1), s, rtol=1e-5)


def test_forward_variants():
    m, x = _facade(0, True, log_retriever_metrics=True)
    q = synth.generate(5, 0, 3, 32, synth.KIND_GAUSS)
    out = m.forward(q.copy(), k=2)
    assert isinstance(out, ram.MipsModelOutput)
    assert [len(e) for e in out.examples] == [2, 2, 2] and len(out.flat_texts) == 6
    assert out.examples[0][0] == f"doc {int(out.indices[0][0])}"
    assert np.array_equal(out.query_cls, orc.prepare_query(q.copy(), True, 0)) and out.metrics is None
    # metrics: aid of the first hit for query 0, unknown for the others
    hit = int(out.indices[0][1])
    out = m(q.copy(), aid=[f"a{hit}", "zz", "zz"], aid_counts=torch.tensor([1, 1, 1]), k=2)
    exp = orc.retriever_metrics(torch.tensor([[0., 1.], [0., 0.], [0., 0.]]), torch.tensor([1, 1, 1]))
    assert out.metrics == pytest.approx(exp)
    # target_only: no search at all
    m.args.memory_forcing = "target_only"
    out = m.forward(q.copy(), target_str=["t0", "t1", "t2"], k=2)
    assert out.scores is None and out.examples == [["t0", "t1", "t2"]] and out.flat_texts == ["t0", "t1", "t2"]
    # target_in with copy_forcing = 1 prepends the target
    m.args.memory_forcing, m.args.copy_forcing = "target_in", 2.0
    out = m.forward(q.copy(), target_str=["t0", "t1", "t2"], k=2)
    assert len(out.flat_texts) == 9 and out.flat_texts[0] == "t0" and out.flat_texts[3] == "t1"
    # dual mode fills up to k with retrieved texts after the input documents
    m.args.memory_forcing, m.args.multi_x_science_dataset_mode = "no_forcing", "dual"
    out = m.forward(q.copy(), input_str=["x <DOC_SEP> y <DOC_SEP> z", "x", "x <DOC_SEP> y"], k=2)
    assert out.flat_texts[:2] == ["x", "y"] and out.flat_texts[2] == "x" and out.flat_texts[3] == out.examples[1][0]


def test_build_index_rejects_non_flat_factories():
    m = ram.Mips(ram.MipsArgs(mips_string_factory="IVF4096,Flat"))
    with pytest.raises(NotImplementedError, match="Flat"):
        m.build_index(np.zeros((4, 8), np.float32))


def test_knowledge_base_nearest_examples_batch():
    x = synth.generate(3, 0, 50, 16, synth.KIND_GAUSS)
    kb = ram.KnowledgeBase({"mips_column": [str(i) for i in range(50)], "aid": [[i] for i in range(50)]},
                           OracleIndex(x[:3]), "mips_cls")
    q = synth.generate(4, 0, 2, 16, synth.KIND_GAUSS)
    scores, examples = kb.get_nearest_examples_batch("mips_cls", q, k=5)   # k > ntotal: -1 ids dropped
    assert [len(s) for s in scores] == [3, 3] and len(examples[0]["mips_column"]) == 3
    assert kb[4]["mips_column"] == "4" and kb[[1, 2]]["aid"] == [[1], [2]] and len(kb) == 50


def test_shard_bounds_cover_and_match_oracle():
    for n in (0, 1, 7, 8, 1003, 1 << 20):
        for w in (1, 2, 3, 4, 8):
            spans = [ram.shard_bounds(n, w, r) for r in range(w)]
            assert spans == [orc.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_pack_unpack_roundtrip():
    s = torch.tensor([[1.5, -0.0, float("-inf")], [3.25, 2.0, 1e-30]])
    i = torch.tensor([[7, 1 << 40, -1], [0, 5, 9]])
    p = ram.pack_topk(s, i)
    assert p.dtype == torch.int64 and p.shape == (2, 3, 2)
    g = torch.stack([p, p + 0])
    us, ui = ram.unpack_gathered(g, 2)
    assert us.shape == (2, 6) and torch.equal(us[:, :3], s) and torch.equal(us[:, 3:], s)
    assert torch.equal(ui[:, :3], i) and torch.equal(ui[:, 3:], i)
    assert np.signbit(us[0, 1].item())      # bit patterns survive (-0.0)


def _gloo_worker(rank, world, port, n, nq, d, k, ret):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = synth.generate(9, 0, n, d, synth.KIND_LATTICE)
        q = synth.generate(10, 0, nq, d, synth.KIND_LATTICE)
        lo, hi = ram.shard_bounds(n, world, rank)

        def local_search(qq, kk, off):      # oracle stands in for the device search of this shard
            assert off == lo
            return orc.search_exact_bruteforce(qq, x[lo:hi], kk, idx_offset=off)

        def merge(cs, ci, parts, kk, metric):  # oracle stands in for mips_merge_topk
            assert parts == world and cs.shape == (nq, world * kk)
            s, i = orc.merge_topk([cs.numpy()], [ci.numpy()], kk, metric)
            return torch.from_numpy(s), torch.from_numpy(i)

        ix = ram.ShardedMipsIndex(d, local_search=local_search, merge=merge)
        assert (ix.rank, ix.world) == (rank, world)
        ix.set_global_size(n)
        s, i = ix.search(q, k)
        es, ei = orc.search_exact_bruteforce(q, x, k)
        s2, i2 = ix.search_async(q, k).result()   # nothing to overlap on this backend: the synchronous path
        ret[rank] = bool(np.array_equal(i, ei) and np.array_equal(s, es) and np.array_equal(i2, ei) and np.array_equal(s2, es))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1003), (2, 3)])
def test_sharded_search_two_ranks_gloo(world, n):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + n % 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gloo_worker, args=(world, port, n, 5, 64, 4, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_encode_shard_bounds_follow_the_reference_chunking():
    # mips.py:227-229: chunck_size = N // num_rank + 1; stop = (rank + 1) * chunck_size if rank + 1 < num_rank else N
    for n in (10, 1003, 4096):
        for w in (1, 2, 3, 8):
            spans = [ram.Mips.encode_shard_bounds(n, r, w) for r in range(w)]
            chunk = n // w + 1
            for r, (a, b) in enumerate(spans):
                exp_stop = (r + 1) * chunk if r + 1 < w else n
                assert (a, b) == (min(n, r * chunk), max(min(n, r * chunk), min(n, exp_stop)))
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert sum(b - a for a, b in spans) == n


def test_bench_defaults_are_the_baseline_headline_config(monkeypatch):
    """bench.py with no flags = BASELINE config 2 on one GPU (2^20 x 768 bf16, Q = 4096, k = 5); the timed path
    must not import the oracle (only cpu_baseline() does)."""
    import ast
    import importlib
    import sys as _sys

    monkeypatch.setattr(_sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")
    a = bench.parse()
    assert (a.gpus, a.rows, a.dim, a.queries, a.k, a.index_dtype) == (1, 1 << 20, 768, 4096, 5, "bf16")
    assert a.steps >= 10 and a.warmup >= 1
    tree = ast.parse(open(bench.__file__).read())
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        imports_oracle = any(isinstance(n, ast.ImportFrom) and (n.module or "").startswith("oracle") for n in ast.walk(fn))
        assert imports_oracle == (fn.name == "cpu_baselines"), fn.name


def test_bench_gpus_n_launches_n_ranks_by_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must produce a 2-rank run by itself: it
    spawns a child torch.distributed.run of the same file BEFORE touching the GPU, relays its output and exits with
    its code.  --launch-check forms the process group, exchanges one tensor and reports what was formed (the search
    itself needs a GPU: -m gpu tests and the driver run it)."""
    import json
    import subprocess
    import sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"launch_check": True, "n_gpus": 2, "world_size": 2, "backend": "gloo", "rank_sum": 3, "self_launched": True}
    # a child that fails makes the parent fail (here: --gpus contradicts the WORLD_SIZE the launcher exports)
    bad = subprocess.run([_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", str(29400 + os.getpid() % 500), os.path.join(root, "bench.py"), "--gpus", "3",
                          "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0


def test_oracle_torch_topk_and_eval_consumer_restatements():
    import torch as _t

    x = synth.generate(1, 0, 5000, 64, synth.KIND_GAUSS)
    q = synth.generate(2, 0, 300, 64, synth.KIND_GAUSS)
    s, i = orc.torch_topk(_t.from_numpy(q), _t.from_numpy(x), 5, q_chunk=128, x_block=1024)
    es, ei = orc.search_exact(q, x, 5)
    assert np.array_equal(i.numpy(), ei)
    np.testing.assert_allclose(s.numpy(), es, rtol=1e-5)
    # nearest_examples_batch drops ids < 0 like datasets/search.py
    cols = {"aid": [f"a{t}" for t in range(3)]}
    sc, ex = orc.nearest_examples_batch(lambda qq, kk: orc.search_exact_bruteforce(qq, x[:3], kk), cols, q[:2], 5)
    assert [len(v) for v in sc] == [3, 3] and all(len(e["aid"]) == 3 for e in ex)
    col, metric = orc.full_kb_eval_index(x, inner_product=False)
    assert col.shape == (5000, 65) and metric == orc.METRIC_L2
    np.testing.assert_allclose((col.astype(np.float64) ** 2).sum(1), float(orc.get_phi(x)), rtol=1e-5)


def test_knowledge_base_add_faiss_index_argument_checks():
    kb = ram.KnowledgeBase({"cls": np.zeros((4, 8), np.float32), "aid": list("abcd")})
    with pytest.raises(NotImplementedError, match="Flat"):
        kb.add_faiss_index("cls", string_factory="IVF16,Flat")
    with pytest.raises(RuntimeError, match="no AMD GPU"):              # a legal build needs the GPU: no CPU fallback
        kb.add_faiss_index("cls", metric_type=ram.METRIC_IP)
    kb2 = ram.KnowledgeBase({"cls": synth.generate(1, 0, 6, 8, synth.KIND_GAUSS)})
    with pytest.raises(NotImplementedError, match="phi-augmented"):
        kb2.add_faiss_index("cls")                                      # metric None = faiss default L2, rows not augmented
    fake = OracleIndex(synth.generate(1, 0, 6, 8, synth.KIND_GAUSS), 0)
    kb2.add_faiss_index("cls", index_name="mine", custom_index=fake)    # HF's custom_index passthrough
    assert kb2.get_index("mine").faiss_index is fake
    s, e = kb2.get_nearest_examples_batch("mine", synth.generate(2, 0, 2, 8, synth.KIND_GAUSS), k=3)
    assert len(s) == 2 and len(e[0]["cls"]) == 3


def test_embedding_shard_files_roundtrip(tmp_path):
    """encode_text2 / build_index() without arguments: the per-rank shard files (mips.py:243-244, 291-295)."""
    n, d = 103, 16
    emb = synth.generate(5, 0, n, d, synth.KIND_GAUSS)
    data = {"mips_column": [f"text {t}" for t in range(n)], "aid": [f"a{t}" for t in range(n)]}
    m = ram.Mips(ram.MipsArgs(mips_tmp_folder=str(tmp_path), mips_batch_size=10), data=data)
    with pytest.raises(RuntimeError, match="encoder"):
        m.encode_text2(0, 3)
    m.encoder = lambda texts: emb[[int(t.split()[1]) for t in texts]]
    for r in (2, 0, 1):                                                  # any order on disk
        m.encode_text2(r, 3)
    embs, cols = m._load_embedding_shards()
    assert np.array_equal(np.concatenate(embs), emb) and cols == data    # rank order restored
    empty = ram.Mips(ram.MipsArgs(mips_tmp_folder=str(tmp_path / "none")))
    with pytest.raises(ValueError, match="no embedding shards"):
        empty._load_embedding_shards()


def test_faiss_shim_surface_without_a_gpu(tmp_path, monkeypatch):
    """faiss_shim: the module-level surface HF `datasets` and sotasum/mips.py touch -- constants, normalize_L2, the factory's
    refusal of anything but "Flat", argument checks, install() -- none of which needs the GPU (index storage is created lazily)."""
    import sys as _s

    fs = ram.faiss_shim
    assert (fs.METRIC_INNER_PRODUCT, fs.METRIC_L2) == (0, 1)
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    x[0] = 0
    ref = orc.l2_normalization(x.copy())
    fs.normalize_L2(x)
    assert np.array_equal(x, ref) and (x[0] == 0).all()
    with pytest.raises(TypeError):
        fs.normalize_L2(np.zeros((2, 3), np.float64))
    with pytest.raises(NotImplementedError, match="Flat"):
        fs.index_factory(8, "IVF16,Flat", fs.METRIC_INNER_PRODUCT)
    ix = fs.index_factory(8, "Flat", fs.METRIC_INNER_PRODUCT)
    assert isinstance(ix, fs.IndexFlat) and ix.d == 8 and ix.ntotal == 0 and ix.is_trained and ix.metric_type == 0
    ix.nprobe = 4
    ix.verbose = True
    assert ix.train(None) is None
    with pytest.raises(ValueError):
        ix.add(np.zeros((2, 7), np.float32))
    l2 = fs.IndexFlat(5)                                       # faiss's default metric is L2
    assert l2.metric_type == fs.METRIC_L2
    with pytest.raises(NotImplementedError, match="phi-augmented"):
        l2.add(np.array([[1, 0, 0, 0, 0], [3, 0, 0, 0, 1]], np.float32))   # rows of different norms: plain L2 is not this path
    monkeypatch.delitem(_s.modules, "faiss", raising=False)
    assert fs.install() is fs and _s.modules["faiss"] is fs
    import importlib.util
    assert importlib.util.find_spec("faiss") is not None      # what HF's `_has_faiss` asks
    _s.modules.pop("faiss", None)


def test_no_shipped_kernel_spills_or_uses_scratch():
    """Every gfx950 kernel of the built library allocates without a spill and without a private segment (a scratch reload
    behind LDS-DMA pieces drains the ring: the register budgets of the scan kernels are set to the last VGPR, and a small edit
    can tip one over -- tools/kernel_regs.py reads the code object's notes, no GPU needed), and the kernels that only exist for
    A/B measurements (`sub` instances, scan_kernel_v5 / scan_kernel_ks) are not in the shipped library."""
    import sys

    ram.build()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    try:
        import kernel_regs
    finally:
        sys.path.pop(0)
    if not os.path.exists(os.path.join(kernel_regs.LLVM, "clang-offload-bundler")):
        pytest.skip("llvm tools of the ROCm image not found")
    ks = kernel_regs.kernels()
    assert len(ks) > 100
    bad = [(k["demangled"], k["spill"], k["scratch"]) for k in ks if k["spill"] or k["scratch"]]
    assert not bad, bad
    names = " ".join(k["demangled"] for k in ks)
    assert "scan_kernel_v4<" in names and "scan_kernel_k3<" in names and "scan_kernel_e8<" in names and "tiny_search_kernel<" in names
    assert "scan_kernel_v5<" not in names and "scan_kernel_ks<" not in names
