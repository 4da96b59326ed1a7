"""The slice of the `faiss` module surface that sotasum's retrieval path touches, answered by the MI355X backend.

The reference never calls its search code directly: it goes through HF `datasets`, which imports `faiss` by name
(sotasum/mips.py:1, 306, 316, 333-345, 383-386, 524, 536, 547; retriever_lightning.py:395-404, 317-321):

    faiss.METRIC_INNER_PRODUCT / faiss.METRIC_L2                      mips.py:306, 316, 369, 371
    faiss.normalize_L2(x)                                             mips.py:524
    Dataset.add_faiss_index(column=..., string_factory="Flat", metric_type=..., train_size=..., faiss_verbose=...)
        -> faiss.index_factory(d, "Flat", metric) / faiss.IndexFlat(d, metric), index.train, index.add   (HF search.py)
    Dataset.get_index(name).faiss_index.search(q, k) / .nprobe        mips.py:343-345, 383-386
    Dataset.get_nearest_examples_batch(name, q, k)                    retriever_lightning.py:317-321
    Dataset.save_faiss_index / load_faiss_index -> faiss.write_index / read_index through Python callbacks   mips.py:536, 547

`install()` registers this module as `faiss` (only if no real faiss is importable) and tells an already imported
`datasets.search` that faiss exists, after which the reference's call text runs UNCHANGED on a real `datasets.Dataset` and
lands in `MipsIndex` (fp32-exact storage by default: `DEFAULT_DTYPE`).  HF's own hook works as well without any of this:
`Dataset.add_faiss_index(column=..., custom_index=MipsIndex(d, metric))` -- it only needs `faiss` importable for its check.

Exact ("Flat") indexes only: every other factory string raises NotImplementedError, like Mips.build_index.

L2: the reference only ever puts phi-AUGMENTED vectors into an L2 index (augment_xb, mips.py:59-65, 316-331): rows of constant
norm sqrt(phi) whose last column is sqrt(phi - |x|^2), queries with a zero last column.  `IndexFlat(d + 1, METRIC_L2)` checks
that, stores the un-augmented d columns and reproduces the augmented squared distances |q|^2 + phi - 2 q.x (the kernel stays
an inner-product kernel); plain L2 on rows of different norms is not this path and is refused.
"""
from __future__ import annotations

import json
import struct
import sys

import numpy as np

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1

DEFAULT_DTYPE = "f32"   # storage of the MipsIndex behind IndexFlat: "f32" (exact on fp32 data) | "bf16" | "fp8_e4m3"
DEFAULT_DEVICE = None   # GPU ordinal (None = torch's current device)

__version__ = "mips-hip-shim"


def normalize_L2(x: np.ndarray) -> None:
    """faiss.normalize_L2: in-place row normalisation of a C-contiguous float32 matrix, rows of norm 0 untouched."""
    if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.ndim == 2 and x.flags.c_contiguous):
        raise TypeError("normalize_L2: a C-contiguous float32 matrix is expected (as faiss does)")
    nr = np.einsum("ij,ij->i", x, x, dtype=np.float32)
    scale = np.ones_like(nr)
    np.divide(np.float32(1.0), np.sqrt(nr), out=scale, where=nr > 0)
    x *= scale[:, None]


class IndexFlat:
    """faiss.IndexFlat(d, metric): d, ntotal, metric_type, is_trained, verbose, nprobe, add / train / search / reset."""

    is_trained = True

    def __init__(self, d: int, metric: int = METRIC_L2, dtype: str = None, device: int = None):
        if metric not in (METRIC_INNER_PRODUCT, METRIC_L2):
            raise NotImplementedError(f"metric {metric!r}: inner product (0) and L2 (1) only")
        self.d = int(d)
        self.metric_type = int(metric)
        self.verbose = False
        self.nprobe = 1
        self._dtype = dtype or DEFAULT_DTYPE
        self._device = DEFAULT_DEVICE if device is None else device
        self._inner = None

    # the MipsIndex behind it: d columns for inner product, d - 1 (augmentation column stripped) for L2
    def _index(self):
        if self._inner is None:
            from .index import MipsIndex

            dd = self.d - 1 if self.metric_type == METRIC_L2 else self.d
            if dd < 1:
                raise ValueError("IndexFlat(METRIC_L2) needs d >= 2 (phi-augmented vectors)")
            self._inner = MipsIndex(dd, metric=self.metric_type, dtype=self._dtype, device=self._device)
        return self._inner

    @property
    def ntotal(self) -> int:
        return 0 if self._inner is None else self._inner.ntotal

    @property
    def mips_index(self):
        """The backend object (MipsIndex): device-tensor searches, margin_stats(), set_param(), ..."""
        return self._index()

    def train(self, x=None) -> None:   # exact index: nothing to train (HF calls it when train_size is given)
        return None

    def reset(self) -> None:
        if self._inner is not None:
            self._inner.reset()

    def add(self, x) -> None:
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add: expected [n, {self.d}], got {x.shape}")
        if self.metric_type == METRIC_L2:
            sq = np.square(x.astype(np.float64)).sum(axis=1)
            phi = getattr(self, "_phi_seen", None)
            ref = float(sq.max()) if phi is None else phi
            if not (len(sq) > 0 and np.all(np.abs(sq - ref) <= 1e-4 * max(ref, 1e-30)) and (x[:, -1] >= 0).all()):
                raise NotImplementedError(
                    "IndexFlat(METRIC_L2).add: the rows are not phi-augmented (constant norm, last column sqrt(phi - |x|^2) as "
                    "augment_xb produces, sotasum/mips.py:59-65); this backend's L2 is that MIPS->L2 reduction")
            self._phi_seen = ref
            x = np.ascontiguousarray(x[:, :-1])
        self._index().add(x)

    def search(self, x, k: int, **kwargs):
        """(D float32 [nq, k], I int64 [nq, k]); NumPy in, NumPy out -- the call of mips.py:383-386 and of HF's search_batch."""
        ix = self._index()
        q = np.ascontiguousarray(x, dtype=np.float32)
        if self.metric_type == METRIC_L2:
            if q.ndim != 2 or q.shape[1] != self.d:
                raise ValueError(f"search: expected [nq, {self.d}], got {q.shape}")
            if np.any(q[:, -1] != 0):
                raise ValueError("L2 search: the queries' last (augmentation) column is not zero; this index answers queries "
                                 "prepared by augment_xq (sotasum/mips.py:68-70)")
            q = np.ascontiguousarray(q[:, :-1])
        return ix.search(q, int(k))


class IndexFlatIP(IndexFlat):
    def __init__(self, d: int):
        super().__init__(d, METRIC_INNER_PRODUCT)


class IndexFlatL2(IndexFlat):
    def __init__(self, d: int):
        super().__init__(d, METRIC_L2)


Index = IndexFlat   # (type annotations in HF spell `faiss.Index`)


def index_factory(d: int, description: str, metric: int = METRIC_L2) -> IndexFlat:
    """faiss.index_factory for the one exact factory string the reference ships with (mips_string_factory = "Flat")."""
    if description != "Flat":
        raise NotImplementedError(f"index_factory({description!r}): only the exact 'Flat' index is implemented")
    return IndexFlat(d, metric)


# ---- write_index / read_index as HF drives them: faiss.write_index(index, faiss.BufferedIOWriter(faiss.PyCallbackIOWriter(f.write)))
class PyCallbackIOWriter:
    def __init__(self, write, bs: int = 1 << 20):
        self.write = write


class PyCallbackIOReader:
    def __init__(self, read, bs: int = 1 << 20):
        self.read = read


class BufferedIOWriter:
    def __init__(self, writer, bsz: int = 1 << 20):
        self.write = writer.write


class BufferedIOReader:
    def __init__(self, reader, bsz: int = 1 << 20):
        self.read = reader.read


_MAGIC = b"MIPSHIP1"


def write_index(index: IndexFlat, writer) -> None:
    """Own byte format (the on-disk format is free): magic, JSON header, raw rows in the storage dtype."""
    if isinstance(writer, str):
        with open(writer, "wb") as f:
            return write_index(index, PyCallbackIOWriter(f.write))
    inner = index._index()
    head = {"d": index.d, "metric": index.metric_type, "dtype": inner.dtype, "ntotal": inner.ntotal, "inner_d": inner.d,
            "phi_seen": getattr(index, "_phi_seen", None)}
    hb = json.dumps(head).encode()
    writer.write(_MAGIC + struct.pack("<q", len(hb)) + hb)
    for r0 in range(0, inner.ntotal, 1 << 16):
        writer.write(inner.rows_raw(r0, min(1 << 16, inner.ntotal - r0)).tobytes())


def _read_exact(reader, n: int) -> bytes:
    out = bytearray()
    while len(out) < n:
        b = reader.read(n - len(out))
        if not b:
            raise EOFError("read_index: truncated index file")
        out += b
    return bytes(out)


def read_index(reader) -> IndexFlat:
    if isinstance(reader, str):
        with open(reader, "rb") as f:
            return read_index(PyCallbackIOReader(f.read))
    if _read_exact(reader, 8) != _MAGIC:
        raise ValueError("read_index: not an index written by this backend's write_index")
    (hl,) = struct.unpack("<q", _read_exact(reader, 8))
    head = json.loads(_read_exact(reader, hl))
    index = IndexFlat(head["d"], head["metric"], dtype=head["dtype"])
    if head.get("phi_seen") is not None:
        index._phi_seen = head["phi_seen"]
    inner = index._index()
    npdt = {"bf16": np.uint16, "fp8_e4m3": np.uint8, "fp8_e4m3_docs": np.uint8, "f32": np.float32}[head["dtype"]]
    row_bytes = head["inner_d"] * np.dtype(npdt).itemsize
    inner.reserve(head["ntotal"])
    for r0 in range(0, head["ntotal"], 1 << 16):
        n = min(1 << 16, head["ntotal"] - r0)
        inner.add(np.frombuffer(_read_exact(reader, n * row_bytes), dtype=npdt).reshape(n, head["inner_d"]))
    return index


def index_gpu_to_cpu(index):   # (HF calls it when a device was given: the index lives on the GPU either way)
    return index


def index_cpu_to_gpu(resources, device, index):
    return index


def index_cpu_to_all_gpus(index):
    return index


class StandardGpuResources:
    pass


def install(force: bool = False):
    """Make `import faiss` resolve to this module (unless a real faiss is importable and force is False) and let an already
    imported HF `datasets.search` know.  Returns the module."""
    import importlib.util

    me = sys.modules[__name__]
    real = None
    if "faiss" in sys.modules and sys.modules["faiss"] is not me:
        real = sys.modules["faiss"]
    elif "faiss" not in sys.modules:
        try:
            real = importlib.util.find_spec("faiss")
        except (ImportError, ValueError):
            real = None
    if real is not None and not force:
        raise RuntimeError("a real `faiss` is importable: pass force=True to shadow it with the MI355X backend")
    sys.modules["faiss"] = me
    search = sys.modules.get("datasets.search")
    if search is not None:
        search._has_faiss = True
    return me
