"""`Mips` -- drop-in for the retrieval half of sotasum/mips.py:154-560 on the MI355X backend.

Kept call surface (what retriever_generator.py:145-153 and lightning_model.py:148-180 call):
    Mips(args)                      config knobs = the mips_* fields of sotasum/model_config.py:44-72
    ._prepare_query(query)          mips.py:368-375
    .search(queries, ignore_indexes, k)     mips.py:382-400 (k+1 fetch, equality filter, lists)
    .forward(queries, aid, aid_counts, target_str, input_str, ignore_indexes, k)   mips.py:402-463
    .build_index(...) / .save() / .load()   mips.py:290-345, 531-549
    .l2_normalization(x), .np_search(x, k), .max_norm, .rebuilt_steps, .embeddings
plus the module-level helpers get_phi / augment_xb / augment_xq (mips.py:55-70), inner_product
(mips.py:552-560) and retriever_metrics (pretrain.py:69-85).

Out of scope: the SPECTER2 / Longformer encoders that produce embeddings and
re-encode retrieved texts (mips.py:87-151, 226-288, 465-519) -- they need hub weights.  `forward`
therefore stops where the reference hands the retrieved texts to those encoders and returns the
texts (`examples`, `flat_texts`) with the scores, metrics and prepared queries.

The index itself is a MipsIndex (HIP, bf16 in HBM).  The reference's L2 mode searches
phi-augmented (d+1)-dimensional vectors (mips.py:316-331, 371-372); the same ranking and the same
squared distances are produced from the un-augmented d-dimensional index as |q|^2 + phi - 2 q.x, so
the kernel stays an inner-product kernel.  `_prepare_query` still appends the zero column exactly
like the reference; `search` strips it again.
"""
from __future__ import annotations

import json
import pickle
import shutil
import threading
from dataclasses import dataclass
from pathlib import Path
from random import random

import numpy as np

from . import _lib
from .index import MipsIndex, l2_normalize_, rows_max_sumsq, rows_max_sumsq_into

METRIC_INNER_PRODUCT = _lib.METRIC_IP
METRIC_L2 = _lib.METRIC_L2


@dataclass
class MipsArgs:
    """The subset of sotasum/model_config.py:4-82 this path reads (same names, same defaults)."""
    mips_disabled: bool = False
    mips_topk: int = 2
    mips_string_factory: str = "Flat"
    mips_nprobe: int = None
    mips_rebuild_every: int = 10000
    mips_train_size: int = -1
    mips_metric_type: int = 0  # 0 -> INNER_PRODUCT ; 1 -> L2
    mips_normalize: bool = True
    mips_db_max_size: int = None
    mips_tmp_folder: str = "./tmp"
    mips_batch_size: int = 32
    log_retriever_metrics: bool = False
    memory_forcing: str = "no_forcing"
    multi_x_science_dataset_mode: str = "original"
    copy_forcing: float = 0.0
    doc_sep: str = " <DOC_SEP> "
    # backend knobs (not in the reference)
    # "f32" (default): fp32-exact -- the neighbours an fp32 brute force (FAISS IndexFlat, mips.py:552-560) returns on the
    # caller's fp32 embeddings, which is what "drop-in" means.  "bf16" / "fp8_e4m3" store the rows rounded (2 / 1 B per
    # element, the fastest scans) and are exact ON THE ROUNDED VALUES: explicit opt-ins for knowledge bases that tolerate it.
    mips_index_dtype: str = "f32"
    mips_device: int = None
    mips_shard: bool = False  # under an initialised torch.distributed group: row-shard the index over the ranks
    #  instead of replicating it on every rank as lightning_model.py:180 does


@dataclass
class MipsModelOutput:
    """Carrier mirroring sotasum/mips.py:33-42 (encoder fields stay None: out of scope)."""
    scores: object = None
    mips_last_hidden_state: object = None
    memory_outputs: dict = None
    memory_input_ids: object = None
    memory_attention_mask: object = None
    metrics: dict = None
    examples: list = None
    query_cls: np.ndarray = None
    indices: object = None      # extension: the retrieved row ids
    flat_texts: list = None     # extension: what the reference tokenises next (mips.py:465)


# --------------------------------------------------------------------------- module-level helpers
def get_phi(xb: np.ndarray):
    """max_i |x_i|^2 (mips.py:55-56)."""
    return np.square(xb).sum(axis=1).max()


def augment_xb(xb: np.ndarray, phi=None) -> np.ndarray:
    """Append sqrt(phi - |x|^2) to every document (mips.py:59-65)."""
    sq = np.square(xb).sum(axis=1)
    if phi is None:
        phi = sq.max()
    return np.hstack((xb, np.sqrt(phi - sq).reshape(-1, 1)))


def augment_xq(xq: np.ndarray) -> np.ndarray:
    """Append a zero column to every query (mips.py:68-70)."""
    return np.hstack((xq, np.zeros((len(xq), 1), dtype="float32")))


def retriever_metrics(pred, counts) -> dict:
    """Recall / reciprocal rank / average precision of a 0/1 hit matrix (pretrain.py:69-85),
    including the reference's behaviour of scoring a rank-0 hit as reciprocal rank 0."""
    import torch

    pred = pred.float()
    counts = counts.to(pred.device)
    first = pred.argmax(dim=-1)
    rr = 1.0 / first.float()
    rr = torch.where(torch.isinf(rr), torch.zeros_like(rr), rr)
    ranks = torch.arange(1, pred.shape[-1] + 1, device=pred.device)
    ap = ((pred.cumsum(dim=-1) / ranks) * pred).sum(dim=-1) / counts
    return {
        "recall": (pred.sum(dim=-1) / counts).mean().item(),
        "reciprocal_rank": rr.mean().item(),
        "average_precision": ap.mean().item(),
    }


def inner_product(x: np.ndarray, y: np.ndarray, k: int = 1, normalize: bool = True, device: int = None,
                  dtype: str = "f32"):
    """Brute-force cross-check of mips.py:552-560, run on the GPU: optional row normalisation of
    both sides, exact top-k of x @ y.T.  Scores are descending; values are the canonical scores of
    the operands as stored (dtype "f32", the default: the caller's fp32 values, i.e. the reference's own
    arithmetic up to summation order; "bf16": rounded to bf16 first -- an opt-in)."""
    import torch

    assert len(x.shape) == len(y.shape) == 2
    dev = f"cuda:{_lib.require_gpu(device)}"
    xd = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
    yd = torch.as_tensor(np.ascontiguousarray(y, dtype=np.float32)).to(dev)
    if normalize:
        l2_normalize_(xd)
        l2_normalize_(yd)
    ix = MipsIndex(y.shape[1], metric=METRIC_INNER_PRODUCT, dtype=dtype, device=device)
    ix.add(yd)
    s, i = ix.search(xd, k)
    return s.cpu().numpy(), i.cpu().numpy()


_IN_BATCH_INDEXES: dict = {}
_IN_BATCH_LOCK = threading.Lock()


def in_batch_scores(query_cls, mips_cls, normalize: bool = False, dtype: str = "f32"):
    """The in-batch scoring of retriever_lightning.py:304-305 (`scores = query_cls @ mips_cls.T; _, i =
    scores.topk(1)`) and, with normalize=True, of :273-277 (`F.normalize(q) @ F.normalize(d).T`), through the
    same search path as everything else: the B (<= 16 in the reference, at most MAX_K here) in-batch documents
    become a B-row exact index, one search with k = B returns every score in rank order.
    CUDA tensors [B, d] in -> (scores float32 [B, B] dense as the reference's matrix, top1 int64 [B]) on the
    device.  dtype "f32" = fp32-exact index (the reference computes this in the model's own precision)."""
    import torch

    if not (isinstance(query_cls, torch.Tensor) and query_cls.is_cuda and isinstance(mips_cls, torch.Tensor) and mips_cls.is_cuda):
        raise ValueError("in_batch_scores expects CUDA tensors [B, d]")
    b, d = mips_cls.shape
    if b > _lib.MAX_K:
        raise NotImplementedError(f"in_batch_scores: {b} in-batch documents > MAX_K = {_lib.MAX_K}")
    q = query_cls.detach().float().contiguous()
    x = mips_cls.detach().float().contiguous()
    if normalize:
        q, x = l2_normalize_(q.clone()), l2_normalize_(x.clone())
    # one small index (+ its scratch) per (device, d, storage), kept for the life of the process: this runs every training /
    # validation step, and creating and destroying an index is a round of hipMalloc / hipFree that serialises the device
    key = (x.device.index, int(d), dtype)
    with _IN_BATCH_LOCK:
        ix = _IN_BATCH_INDEXES.get(key)
        if ix is None:
            ix = _IN_BATCH_INDEXES[key] = MipsIndex(d, metric=METRIC_INNER_PRODUCT, dtype=dtype, device=x.device.index)
            ix.reserve(_lib.MAX_K)
        ix.reset()
        ix.add(x)
        s, i = ix.search(q, b)
    dense = torch.empty_like(s).scatter_(1, i, s)
    return dense, i[:, 0].contiguous()


def _strip_augmentation_column(index, q: np.ndarray) -> np.ndarray:
    """Queries prepared for the reference's L2 mode carry the zero column of augment_xq (mips.py:68-70,
    371-372; retriever_lightning.py:313-315): [B, d + 1].  The index here stores the un-augmented d columns
    and produces the augmented distances itself, so the column is dropped -- after checking that it IS zero
    (anything else would change q.x' and cannot be answered from a stripped index)."""
    if index.metric_type == METRIC_L2 and q.ndim == 2 and q.shape[1] == index.d + 1:
        if np.any(q[:, -1] != 0):
            raise ValueError("L2 search: the queries' last (augmentation) column is not zero; this index stores "
                             "un-augmented rows and can only answer queries prepared by augment_xq")
        return q[:, :-1]
    return q


# --------------------------------------------------------------------------- the facade
class _IndexHolder:
    """`embeddings.get_index(name).faiss_index` shim (mips.py:343-345, 383)."""

    def __init__(self, index):
        self.faiss_index = index


class KnowledgeBase:
    """The few Dataset operations the path uses on `self.embeddings`: row fetch by id
    (`embeddings[i][text_column]`, mips.py:428, 458), get_index (mips.py:383) and len."""

    def __init__(self, columns: dict, index: MipsIndex = None, index_name: str = "mips_embeddings"):
        self.columns = columns
        self._indexes = {index_name: _IndexHolder(index)} if index is not None else {}

    def __len__(self):
        return len(next(iter(self.columns.values()))) if self.columns else 0

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.columns[key]
        if isinstance(key, (int, np.integer)):
            return {c: v[int(key)] for c, v in self.columns.items()}
        ids = [int(i) for i in key]
        return {c: [v[i] for i in ids] for c, v in self.columns.items()}

    def get_index(self, name: str):
        return self._indexes[name]

    def add_index(self, name: str, index: MipsIndex):
        self._indexes[name] = _IndexHolder(index)

    def drop_index(self, name: str):
        self._indexes.pop(name, None)

    def add_faiss_index(self, column: str, index_name: str = None, device: int = None, string_factory: str = None,
                        metric_type: int = None, custom_index=None, batch_size: int = 1000, train_size: int = None,
                        faiss_verbose: bool = False, dtype: str = "f32"):
        """HF Dataset.add_faiss_index as the reference calls it (retriever_lightning.py:395-404 / pretrain.py:
        470-479: `add_faiss_index(column="cls", index_name="mips_cls", metric_type=metric)`; mips.py:333-340 with
        string_factory / train_size / faiss_verbose).  Builds an exact MipsIndex over `self.columns[column]`
        ([N, d'] float32); dtype "f32" (default) = fp32-exact storage, the reference's neighbours on fp32 data.
        metric_type None = faiss.IndexFlat's default (L2), as in HF.
          * inner product: the column is stored as it is;
          * L2: the reference only ever indexes phi-AUGMENTED vectors here (augment_xb, retriever_lightning.py:
            373-389): rows of constant norm sqrt(phi) whose last column is sqrt(phi - |x|^2).  The backend keeps
            the un-augmented d = d' - 1 columns and reproduces the augmented distances |q|^2 + phi - 2 q.x, so a
            constant-norm column is stripped of its last element; any other L2 column is refused (plain L2 on rows
            of different norms is not this path)."""
        if string_factory not in (None, "Flat"):
            raise NotImplementedError(f"string_factory={string_factory!r}: only the exact 'Flat' index is implemented")
        index_name = index_name if index_name is not None else column
        if custom_index is not None:
            self._indexes[index_name] = _IndexHolder(custom_index)
            return self
        metric = METRIC_L2 if metric_type is None else int(metric_type)
        x = np.asarray(self.columns[column], dtype=np.float32)
        if x.ndim != 2:
            raise ValueError(f"add_faiss_index: column {column!r} must be a matrix [N, d], got shape {x.shape}")
        if metric == METRIC_L2:
            sq = np.square(x.astype(np.float64)).sum(axis=1)
            constant_norm = x.shape[0] > 0 and x.shape[1] >= 2 and (sq.max() - sq.min()) <= 1e-4 * max(sq.max(), 1e-30)
            if not (constant_norm and (x[:, -1] >= 0).all()):
                raise NotImplementedError(
                    "add_faiss_index(metric_type=L2): the rows are not phi-augmented (constant norm, last column "
                    "sqrt(phi - |x|^2) as augment_xb produces, sotasum/mips.py:59-65); this backend's L2 is that "
                    "MIPS->L2 reduction")
            x = x[:, :-1]
        elif metric != METRIC_INNER_PRODUCT:
            raise NotImplementedError(f"metric_type={metric_type!r}: inner product (0) and L2 (1) only")
        index = MipsIndex(x.shape[1], metric=metric, dtype=dtype, device=device)
        index.reserve(x.shape[0])
        for r0 in range(0, x.shape[0], max(int(batch_size), 1 << 16)):  # HF adds in batches; rows are converted on the device
            index.add(x[r0:r0 + max(int(batch_size), 1 << 16)])
        self._indexes[index_name] = _IndexHolder(index)
        return self

    def get_nearest_examples_batch(self, index_name: str, queries, k: int = 10):
        """HF Dataset.get_nearest_examples_batch as used at retriever_lightning.py:317-321: returns (scores
        per query, examples per query as dict of columns); ids < 0 are dropped like datasets/search.py does.
        L2 indexes take the reference's augmented queries ([B, d + 1], zero last column) as they come."""
        index = self.get_index(index_name).faiss_index
        q = _strip_augmentation_column(index, np.asarray(queries, dtype=np.float32))
        s, i = index.search(np.ascontiguousarray(q), k)
        scores, examples = [], []
        for row_s, row_i in zip(s, i):
            keep = row_i >= 0
            scores.append(row_s[keep])
            examples.append(self[row_i[keep]])
        return scores, examples


class Mips:
    def __init__(self, args: MipsArgs = None, data: dict = None) -> None:
        self.args = args if args is not None else MipsArgs()

        self.tmp_folder = Path(self.args.mips_tmp_folder)
        self.embeddings_tmp_folder = self.tmp_folder / "embeddings_tmp"
        self.mips_folder = self.tmp_folder / "mips"
        self.index_file = self.mips_folder / "index"          # directory (meta.json + rows.bf16)
        self.max_norm_file = self.mips_folder / "max_norm.pkl"
        self.embeddings_folder = self.mips_folder / "embeddings"

        # knowledge-base columns (text_column / index_column lists); the reference loads them from
        # Multi-XScience / arXiv (mips.py:167-185), which needs the hub -> supplied by the caller
        self.data = data
        if data is not None and isinstance(self.args.mips_db_max_size, int):
            self.data = {c: v[: self.args.mips_db_max_size] for c, v in data.items()}

        self.string_factory = self.args.mips_string_factory
        self.train_size = self.args.mips_train_size
        self.metric_type = self.args.mips_metric_type
        self.normalize = self.args.mips_normalize

        self.max_norm = None
        self.phi = None
        self.rebuilt_steps = [0]
        self.text_column = "mips_column"
        self.index_column = "aid"
        self.index_name = "mips_embeddings"
        self.embeddings: KnowledgeBase = None
        self.embeddings_column = "embeddings"
        self.scale_topk = 16

    # ------------------------------------------------------------------ index build (mips.py:290-345)
    def _dist(self):
        """(rank, world, group-initialised?) of the default torch.distributed group."""
        try:
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized():
                return dist.get_rank(), dist.get_world_size(), True
        except Exception:
            pass
        return 0, 1, False

    def _sharded(self) -> bool:
        return bool(self.args.mips_shard) and self._dist()[1] > 1

    def init_embeddings_folder(self) -> None:
        """mips.py:246-249 (@rank_zero_only there)."""
        if self._dist()[0] != 0:
            return
        shutil.rmtree(self.embeddings_folder, ignore_errors=True)
        self.embeddings_folder.mkdir(parents=True, exist_ok=True)

    @staticmethod
    def encode_shard_bounds(n_rows: int, rank: int, num_rank: int):
        """Row range [start, stop) rank `rank` encodes in the reference's corpus-sharded encoding
        (`encode_text2`, mips.py:227-229: chunks of N // num_rank + 1 rows, the last rank takes the
        rest).  The encoders themselves are out of scope; this is the arithmetic a caller needs to hand
        per-rank embedding shards to `build_index` in the reference's order."""
        chunk = (n_rows // num_rank) + 1
        start = min(n_rows, rank * chunk)
        stop = min(n_rows, (rank + 1) * chunk) if rank + 1 < num_rank else n_rows
        return start, max(start, stop)

    def save_embeddings_shard(self, rank: int, embeddings, columns: dict = None) -> None:
        """What `encode_text2` leaves behind for `build_index` (mips.py:243-244: save_to_disk(embeddings_tmp/
        <rank>)): this rank's CLS vectors [n, d] float32 (+ its slice of the text / id columns) under
        embeddings_tmp/<rank>/.  Format: embeddings.npy + columns.json."""
        folder = self.embeddings_tmp_folder / str(int(rank))
        folder.mkdir(parents=True, exist_ok=True)
        try:
            import torch

            if isinstance(embeddings, torch.Tensor):
                embeddings = embeddings.detach().float().cpu().numpy()
        except ImportError:
            pass
        np.save(folder / "embeddings.npy", np.ascontiguousarray(embeddings, dtype=np.float32))
        with open(folder / "columns.json", "w") as f:
            json.dump(columns if columns is not None else {}, f)

    def encode_text2(self, rank: int, num_rank: int) -> None:
        """mips.py:226-244: this rank encodes rows [rank * (N // num_rank + 1), ...) of the knowledge base and
        saves them for `build_index`.  The encoder (SPECTER2 / Longformer, mips.py:87-151) is out of scope:
        `self.encoder` must be a callable `list[str] -> float32 [n, d]` supplied by the caller."""
        if self.data is None or getattr(self, "encoder", None) is None:
            raise RuntimeError("encode_text2 needs the knowledge-base columns (data=...) and a callable self.encoder "
                               "(texts -> [n, d] embeddings); the reference's encoders need hub weights and are not built")
        n = len(self.data[self.text_column])
        start, stop = self.encode_shard_bounds(n, rank, num_rank)
        cols = {c: list(v[start:stop]) for c, v in self.data.items()}
        bs = max(1, int(self.args.mips_batch_size))
        parts = [np.asarray(self.encoder(cols[self.text_column][r0:r0 + bs]), dtype=np.float32)
                 for r0 in range(0, stop - start, bs)]
        emb = np.concatenate(parts, axis=0) if parts else np.zeros((0, 0), np.float32)
        self.save_embeddings_shard(rank, emb, cols)

    def _load_embedding_shards(self):
        """mips.py:291-295: concatenate the per-rank shards (in rank order) -> (embeddings [N, d], columns)."""
        folders = sorted((f for f in self.embeddings_tmp_folder.glob("*") if (f / "embeddings.npy").exists()),
                         key=lambda f: int(f.name) if f.name.isdigit() else 1 << 30)
        if not folders:
            raise ValueError(f"build_index(): no embedding shards under {self.embeddings_tmp_folder} "
                             "(encode_text2 / save_embeddings_shard write them) and no matrix was passed")
        embs, cols = [], {}
        for f in folders:
            e = np.load(f / "embeddings.npy")
            if e.size:
                embs.append(e)
            with open(f / "columns.json") as fh:
                for c, v in json.load(fh).items():
                    cols.setdefault(c, []).extend(v)
        return embs, cols

    def build_index(self, embeddings=None) -> None:
        """max_norm (mips.py:298-304) -> optional document normalisation for IP (:306-314) ->
        [L2: phi, mips.py:316-324; the augmentation column is implicit in the backend] ->
        Flat index (:333-340).
        `embeddings`: float32 [N, d] NumPy array / torch tensor, or a list of per-rank shards in rank order;
        None = the reference's own call text (`mips.build_index()`, lightning_model.py:176): the shards that
        encode_text2 / save_embeddings_shard left under embeddings_tmp/ are concatenated (mips.py:291-295).
        Like the reference (@rank_zero_only, mips.py:290) only rank 0 builds when a process group is up; the
        other ranks return at once and pick the index up in load()."""
        import torch

        if self.string_factory not in (None, "Flat"):
            raise NotImplementedError(
                f"mips_string_factory={self.string_factory!r}: only the exact 'Flat' index is implemented")
        rank, world, up = self._dist()
        if up and rank != 0:
            return
        shard_cols = None
        if embeddings is None:
            embeddings, shard_cols = self._load_embedding_shards()
        dev = f"cuda:{_lib.require_gpu(self.args.mips_device)}"
        if isinstance(embeddings, (list, tuple)):  # per-rank shards in rank order (mips.py:292-295 concatenates them)
            embeddings = torch.cat([torch.as_tensor(e).to(dev, dtype=torch.float32) for e in embeddings], dim=0)
        x = torch.as_tensor(embeddings)
        if isinstance(self.args.mips_db_max_size, int):
            x = x[: self.args.mips_db_max_size]
        x = x.to(dev, dtype=torch.float32).contiguous()
        if isinstance(embeddings, torch.Tensor) and x.data_ptr() == embeddings.data_ptr():
            x = x.clone()  # normalisation below is in place; never touch the caller's tensor

        self.max_norm = float(np.sqrt(rows_max_sumsq(x)))
        if self.normalize and self.metric_type == METRIC_INNER_PRODUCT:
            l2_normalize_(x)

        index = MipsIndex(x.shape[1], metric=self.metric_type, dtype=self.args.mips_index_dtype,
                          device=self.args.mips_device)
        index.reserve(x.shape[0])
        index.add(x)
        if self.metric_type == METRIC_L2:
            self.phi = index.phi()
        if isinstance(self.args.mips_nprobe, int):
            index.nprobe = self.args.mips_nprobe
        cols = dict(self.data) if self.data is not None else (shard_cols or {})
        self.embeddings = KnowledgeBase(cols, index, self.index_name)

    # ---- streaming form: CLS vectors go into the index AS THE ENCODER PRODUCES THEM, no Arrow-on-disk
    # exchange (mips.py:243-244, 292-295), no second pass: max-norm is a running maximum, the IP normalisation is per
    # row, phi is taken from the finished index -- the result is bit-identical to build_index(all rows at once)
    def begin_index_build(self, d: int) -> None:
        if self.string_factory not in (None, "Flat"):
            raise NotImplementedError(
                f"mips_string_factory={self.string_factory!r}: only the exact 'Flat' index is implemented")
        self._building = {"index": MipsIndex(int(d), metric=self.metric_type, dtype=self.args.mips_index_dtype,
                                             device=self.args.mips_device),
                          "max_sq": None, "cols": {}, "rows": 0}   # max_sq: one float64 ON THE DEVICE, read once in end_index_build

    def add_embeddings(self, batch, columns: dict = None) -> None:
        """One encoder batch: float32 [n, d] (CUDA tensor straight from the encoder, or NumPy) + its slice of the
        text / id columns.  Honours mips_db_max_size like the one-shot build."""
        import torch

        b = getattr(self, "_building", None)
        if b is None:
            raise RuntimeError("add_embeddings: call begin_index_build(d) first")
        dev = f"cuda:{_lib.require_gpu(self.args.mips_device)}"
        x = torch.as_tensor(batch)
        if isinstance(self.args.mips_db_max_size, int):
            room = max(0, self.args.mips_db_max_size - b["rows"])
            x = x[:room]
            if columns is not None:
                columns = {c: list(v)[:room] for c, v in columns.items()}
        if x.shape[0] == 0:
            return
        y = x.to(dev, dtype=torch.float32).contiguous()
        if isinstance(batch, torch.Tensor) and y.data_ptr() == batch.data_ptr():
            y = y.clone()  # the normalisation below is in place; never touch the encoder's tensor
        if b["max_sq"] is None:
            b["max_sq"] = torch.zeros(1, dtype=torch.float64, device=y.device)
        rows_max_sumsq_into(y, b["max_sq"])   # running maximum on the device: no host copy, no synchronisation per batch
        if self.normalize and self.metric_type == METRIC_INNER_PRODUCT:
            l2_normalize_(y)
        b["index"].add(y)
        b["rows"] += int(y.shape[0])
        for c, v in (columns or {}).items():
            b["cols"].setdefault(c, []).extend(list(v))

    def end_index_build(self) -> None:
        b = getattr(self, "_building", None)
        if b is None:
            raise RuntimeError("end_index_build: no build in progress")
        self._building = None
        self.max_norm = float(np.sqrt(b["max_sq"].item())) if b["max_sq"] is not None else 0.0   # the build's ONE host copy
        index = b["index"]
        if self.metric_type == METRIC_L2 and index.ntotal > 0:
            self.phi = index.phi()
        if isinstance(self.args.mips_nprobe, int):
            index.nprobe = self.args.mips_nprobe
        cols = b["cols"] if b["cols"] else (dict(self.data) if self.data is not None else {})
        self.embeddings = KnowledgeBase(cols, index, self.index_name)

    def build_index_sharded(self, embeddings) -> None:
        """Collective form of build_index for mips_shard=True: EVERY rank passes the full [N, d] matrix (array or
        memory map) and keeps rows [r * ceil(N / G), ...) only; max_norm and phi are all-reduced (MAX) so every
        shard normalises / measures distances like the unsharded index.  No disk round trip
        (the reference's is mips.py:243-244 + 292-295 + 531-549)."""
        import torch
        import torch.distributed as dist

        from .sharded import ShardedMipsIndex

        if self.string_factory not in (None, "Flat"):
            raise NotImplementedError(
                f"mips_string_factory={self.string_factory!r}: only the exact 'Flat' index is implemented")
        n = len(embeddings)
        if isinstance(self.args.mips_db_max_size, int):
            n = min(n, self.args.mips_db_max_size)
        dev_id = _lib.require_gpu(self.args.mips_device)
        sh = ShardedMipsIndex(int(np.shape(embeddings)[1]), metric=self.metric_type, dtype=self.args.mips_index_dtype,
                              device=dev_id)
        lo, hi = sh.set_global_size(n)
        x = torch.as_tensor(np.ascontiguousarray(embeddings[lo:hi], dtype=np.float32)
                            if not isinstance(embeddings, torch.Tensor) else embeddings[lo:hi]).to(f"cuda:{dev_id}", dtype=torch.float32).contiguous()
        if isinstance(embeddings, torch.Tensor) and x.data_ptr() == embeddings[lo:hi].data_ptr():
            x = x.clone()
        local_max = rows_max_sumsq(x) if hi > lo else 0.0
        if sh.world > 1:
            backend = dist.get_backend()
            t = torch.tensor([local_max], dtype=torch.float64, device=f"cuda:{dev_id}" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            local_max = float(t.item())
        self.max_norm = float(np.sqrt(local_max))
        if self.normalize and self.metric_type == METRIC_INNER_PRODUCT and hi > lo:
            l2_normalize_(x)
        if hi > lo:
            sh.local.reserve(hi - lo)
            sh.local.add(x)
        sh._sync_phi()
        if self.metric_type == METRIC_L2:
            self.phi = sh.phi()
        cols = dict(self.data) if self.data is not None else {}
        self.embeddings = KnowledgeBase(cols, sh, self.index_name)

    # ------------------------------------------------------------------ query side
    def l2_normalization(self, x: np.ndarray) -> np.ndarray:
        """mips.py:521-525 (faiss.normalize_L2: in place, rows of norm 0 untouched)."""
        if not x.flags.c_contiguous:
            x = np.asarray(x, order="C")
        nr = np.einsum("ij,ij->i", x, x, dtype=np.float32)
        scale = np.ones_like(nr)
        np.divide(np.float32(1.0), np.sqrt(nr), out=scale, where=nr > 0)
        x *= scale[:, None]
        return x

    def _prepare_query(self, query: np.ndarray) -> np.ndarray:
        """mips.py:368-375."""
        if self.normalize and self.metric_type == METRIC_INNER_PRODUCT:
            query = self.l2_normalization(query)
        if self.metric_type == METRIC_L2:
            query = augment_xq(query)
        if not query.flags.c_contiguous:
            query = np.asarray(query, order="C")
        return query.astype(np.float32)

    def _index(self) -> MipsIndex:
        if self.embeddings is None:
            raise RuntimeError("Mips: no index (call build_index() or load() first)")
        return self.embeddings.get_index(self.index_name).faiss_index

    def search(self, queries: np.ndarray, ignore_indexes: list = None, k: int = 10):
        """mips.py:382-400: k (or k+1 when filtering) nearest rows; with ignore_indexes the hit
        equal to ignore_indexes[j] is dropped per query and the rest cut to k (lists of lists)."""
        index = self._index()
        q = _strip_augmentation_column(index, np.asarray(queries))  # the zero column of augment_xq
        scores, indices = index.search(q, k + 1 if ignore_indexes is not None else k)
        if ignore_indexes is not None:
            out_s, out_i = [], []
            for j in range(len(indices)):
                banned = int(ignore_indexes[j])
                keep = [t for t in range(len(indices[j])) if indices[j][t] != banned][:k]
                out_s.append([scores[j][t] for t in keep])
                out_i.append([indices[j][t] for t in keep])
            scores, indices = out_s, out_i
        return scores, indices

    def search_device(self, queries, ignore_indexes=None, k: int = 10, prepare: bool = True):
        """Device-resident form of `_prepare_query` + `search`: takes the query CLS
        vectors as a CUDA tensor [B, d] straight from the encoder -- removing the
        `.detach().cpu().float().numpy()` hop of retriever_generator.py:143 -- normalises them on the
        device when the reference would (mips.py:369-370), searches, applies the ignore filter of
        mips.py:388-398 on the device and returns CUDA (scores [B, k], indices [B, k]); nothing
        synchronises."""
        import torch

        from .index import filter_ignore

        index = self._index()
        q = queries
        if not (isinstance(q, torch.Tensor) and q.is_cuda):
            raise ValueError("search_device expects a CUDA tensor [B, d]")
        want_norm = bool(prepare and self.normalize and self.metric_type == METRIC_INNER_PRODUCT)
        if isinstance(index, MipsIndex):
            # prepare + search + ignore filter in one library call: one kernel launch at the reference's own sizes
            # (B <= 16 queries, ~10^4 documents), the same separate steps otherwise -- identical results
            qq = q.detach()
            if qq.dim() == 2 and self.metric_type == METRIC_L2 and qq.shape[1] == index.d + 1:
                qq = qq[:, :-1]  # augment_xq's zero column (not checked here: that would synchronise)
            if want_norm:
                qq = qq.float()
            return index.search_fused(qq.contiguous(), k, normalize=want_norm, ignore=ignore_indexes)
        if want_norm:
            q = l2_normalize_(q.detach().float().contiguous().clone())
        if q.dim() == 2 and self.metric_type == METRIC_L2 and q.shape[1] == index.d + 1:
            q = q[:, :-1].contiguous()  # augment_xq's zero column (not checked here: that would synchronise)
        if ignore_indexes is None:
            return index.search(q, k)
        s, i = index.search(q, k + 1)
        return filter_ignore(s, i, ignore_indexes, k)

    def np_search(self, x, k: int = 2) -> tuple:
        """mips.py:527-529: exhaustive cross-check over the stored embeddings."""
        index = self._index()
        q = np.array(x, dtype=np.float32, copy=True)
        if self.normalize:
            q = self.l2_normalization(q)
        return index.search(q, k, force_ip=True)  # the reference's cross-check is always an inner product

    # ------------------------------------------------------------------ forward (mips.py:402-463)
    def forward(self, queries: np.ndarray, aid: list = None, aid_counts=None, target_str: list = None,
                input_str: list = None, ignore_indexes: list = None, k: int = 10) -> MipsModelOutput:
        a = self.args
        indices = None
        if a.memory_forcing == "target_only" and a.multi_x_science_dataset_mode == "original":
            flat_texts = target_str
            k = 1
            scores = None
            examples = [target_str]
        else:
            queries = self._prepare_query(query=queries)
            scores, indices = self.search(queries=queries, ignore_indexes=ignore_indexes, k=k)
            examples = [self.embeddings[[i for i in row if i >= 0]][self.text_column] for row in indices]

            if (a.memory_forcing == "target_in" and a.multi_x_science_dataset_mode == "original"
                    and a.copy_forcing > random() and isinstance(target_str, list)):
                flat_texts = [t for i, df in enumerate(examples) for t in ([target_str[i]] + df)]
                k += 1
            elif a.multi_x_science_dataset_mode == "dual" and input_str is not None:
                input_list = (i.split(a.doc_sep)[:k] for i in input_str)
                flat_texts = [j for e, i in zip(examples, input_list) for j in i + e[: (k - len(i))]]
            else:
                flat_texts = [t for df in examples for t in df]

        metrics = None
        if aid is not None and a.log_retriever_metrics and indices is not None:
            import torch

            full = [self.embeddings[[i for i in row if i >= 0]] for row in indices]
            pred = torch.tensor([[b == x for x in e[self.index_column]] for e, b in zip(full, aid)]).float()
            metrics = retriever_metrics(pred, torch.as_tensor(aid_counts).cpu())

        return MipsModelOutput(scores=scores, metrics=metrics, examples=examples, query_cls=queries,
                               indices=indices, flat_texts=flat_texts)

    __call__ = forward

    # ------------------------------------------------------------------ persistence (mips.py:531-549)
    def save(self) -> None:
        """mips.py:531-543 (@rank_zero_only there): index file + columns + max_norm.  With a row-sharded index
        (build_index_sharded) the save is collective: every rank appends its rows to the one file set."""
        from .sharded import ShardedMipsIndex

        rank, world, up = self._dist()
        index = self._index() if self.embeddings is not None else None
        collective = isinstance(index, ShardedMipsIndex) and index.world > 1
        if up and rank != 0 and not collective:
            return
        extra = {"phi": self.phi, "normalize": bool(self.normalize)}
        if rank == 0:
            shutil.rmtree(self.mips_folder, ignore_errors=True)
            self.mips_folder.mkdir(parents=True, exist_ok=True)
        index.save(str(self.index_file), extra=extra)
        if rank == 0:
            self.embeddings_folder.mkdir(parents=True, exist_ok=True)
            with open(self.embeddings_folder / "columns.json", "w") as f:
                json.dump(self.embeddings.columns, f)
            with open(self.max_norm_file, "wb") as f:
                pickle.dump(self.max_norm, f)
            shutil.rmtree(self.embeddings_tmp_folder, ignore_errors=True)
        self.embeddings = None

    def load(self) -> None:
        """mips.py:545-549: every rank loads.  The reference loads the WHOLE index on every rank
        (lightning_model.py:180); with mips_shard=True under a process group each rank keeps only its row range
        of the same files (ShardedMipsIndex.load) and searches go through the one all-gather + merge."""
        if self._sharded():
            from .sharded import ShardedMipsIndex

            index = ShardedMipsIndex.load(str(self.index_file), device=self.args.mips_device)
        else:
            index = MipsIndex.load(str(self.index_file), device=self.args.mips_device)
        with open(self.embeddings_folder / "columns.json") as f:
            cols = json.load(f)
        self.embeddings = KnowledgeBase(cols, index, self.index_name)
        self.phi = index.meta.get("phi")
        with open(self.max_norm_file, "rb") as f:
            self.max_norm = pickle.load(f)
