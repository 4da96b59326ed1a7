"""MipsIndex -- the faiss.IndexFlat-shaped object the reference reaches through
`self.embeddings.get_index(self.index_name).faiss_index` (sotasum/mips.py:383-386, 342-345).

Duck type kept:  d, ntotal, metric_type, nprobe (settable, ignored: the index is exact),
add(x), search(x, k) -> (D float32 [nq,k], I int64 [nq,k]), reset().
Extensions: torch CUDA tensors in -> torch CUDA tensors out (no host hop), add_synthetic,
save/load of the raw bf16 shard, last_scan_ms for the bench.

All arithmetic happens in libmips_hip.so (hand-written HIP, gfx950); this file only moves
pointers.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
import json
import os
import threading

import numpy as np

from . import _lib

_FORMAT_VERSION = 1


def _stream_handle(device: int) -> int:
    import torch

    return int(torch.cuda.current_stream(device).cuda_stream)


class MipsIndex:
    def __init__(self, d: int, metric: int = _lib.METRIC_IP, dtype: str = "bf16", device: int | None = None):
        if dtype not in ("bf16", "fp8_e4m3", "fp8_e4m3_docs", "f32"):
            raise NotImplementedError(f"index dtype {dtype!r}: this build stores 'bf16', 'fp8_e4m3' (queries e4m3 too), "
                                      "'fp8_e4m3_docs' (e4m3 rows, bf16 queries) or 'f32'")
        self._lib = _lib.load()
        self.device = _lib.require_gpu(device)
        self._h = ctypes.c_void_p()
        self._code = {"bf16": _lib.DTYPE_BF16, "fp8_e4m3": _lib.DTYPE_FP8_E4M3, "fp8_e4m3_docs": _lib.DTYPE_FP8_E4M3_DOCS,
                      "f32": _lib.DTYPE_F32}[dtype]
        self._f8 = self._code in (_lib.DTYPE_FP8_E4M3, _lib.DTYPE_FP8_E4M3_DOCS)   # e4m3 storage
        _lib.check(self._lib.mips_index_create(ctypes.byref(self._h), self.device, int(d), self._code,
                                               int(metric)), "mips_index_create")
        self._d = int(d)
        self._metric = int(metric)
        self.dtype = dtype
        self.nprobe = 1  # accepted for drop-in compatibility (mips.py:342-345); exact search ignores it
        self._mutex = threading.Lock()

    # ------------------------------------------------------------------ faiss-like attributes
    @property
    def d(self) -> int:
        return self._d

    @property
    def ntotal(self) -> int:
        return int(self._lib.mips_index_ntotal(self._h))

    @property
    def metric_type(self) -> int:
        return self._metric

    is_trained = True

    def __len__(self) -> int:
        return self.ntotal

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.mips_index_destroy(h)
            except Exception:
                pass
            h.value = None

    # ------------------------------------------------------------------ helpers
    def _as_buffer(self, x, what: str):
        """-> (pointer, dtype code, is_device, n, keepalive)"""
        import torch

        if isinstance(x, torch.Tensor):
            if x.dim() != 2 or x.shape[1] != self._d:
                raise ValueError(f"{what}: expected [n, {self._d}], got {tuple(x.shape)}")
            if x.dtype == torch.float32:
                code = _lib.DTYPE_F32
            elif x.dtype == torch.bfloat16:
                code = _lib.DTYPE_BF16
            elif x.dtype == getattr(torch, "float8_e4m3fn", None) and self._f8 and not (what == "search" and self._code == _lib.DTYPE_FP8_E4M3_DOCS):
                code = _lib.DTYPE_FP8_E4M3  # raw e4m3 bytes
            else:
                x = x.float()
                code = _lib.DTYPE_F32
            x = x.contiguous()
            if x.is_cuda:
                if x.device.index != self.device:
                    x = x.to(f"cuda:{self.device}")
                return x.data_ptr(), code, 1, x.shape[0], x
            return x.data_ptr(), code, 0, x.shape[0], x
        a = np.asarray(x)
        if a.ndim != 2 or a.shape[1] != self._d:
            raise ValueError(f"{what}: expected [n, {self._d}], got {a.shape}")
        if a.dtype == np.uint16:  # raw bf16 bit patterns
            a = np.ascontiguousarray(a)
            return a.ctypes.data, _lib.DTYPE_BF16, 0, a.shape[0], a
        if a.dtype == np.uint8 and self._f8 and not (what == "search" and self._code == _lib.DTYPE_FP8_E4M3_DOCS):  # raw e4m3 codes
            a = np.ascontiguousarray(a)
            return a.ctypes.data, _lib.DTYPE_FP8_E4M3, 0, a.shape[0], a
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a.ctypes.data, _lib.DTYPE_F32, 0, a.shape[0], a

    # ------------------------------------------------------------------ building
    def reserve(self, n: int) -> None:
        _lib.check(self._lib.mips_index_reserve(self._h, int(n)), "mips_index_reserve")

    def add(self, x) -> None:
        """faiss Index.add: append rows (np.float32 / np.uint16 bf16 bits / torch float32|bfloat16,
        host or device).  float32 is rounded to bf16 (RNE) on the device."""
        ptr, code, is_dev, n, keep = self._as_buffer(x, "add")
        with self._mutex:
            _lib.check(self._lib.mips_index_add(self._h, ptr, n, code, is_dev, _stream_handle(self.device)),
                       "mips_index_add")
        del keep

    def add_synthetic(self, n: int, row0: int = 0, seed: int = 0xD0C5, kind: int = _lib.SYNTH_GAUSS) -> None:
        with self._mutex:
            _lib.check(self._lib.mips_index_add_synthetic(self._h, int(n), int(row0), int(seed), int(kind),
                                                          _stream_handle(self.device)), "mips_index_add_synthetic")

    def reset(self) -> None:
        _lib.check(self._lib.mips_index_reset(self._h), "mips_index_reset")

    def phi(self) -> float:
        out = ctypes.c_double()
        _lib.check(self._lib.mips_index_phi(self._h, ctypes.byref(out), _stream_handle(self.device)), "mips_index_phi")
        return out.value

    def set_phi(self, phi: float) -> None:
        """Override phi (row-sharded L2 indexes: the maximum over all shards)."""
        _lib.check(self._lib.mips_index_set_phi(self._h, float(phi)), "mips_index_set_phi")

    def clear_phi(self) -> None:
        """Drop a set_phi override: phi() is this shard's own maximum again (recomputed from the stored rows)."""
        _lib.check(self._lib.mips_index_set_phi(self._h, -1.0), "mips_index_set_phi")

    def rows_raw(self, row0: int = 0, n: int | None = None) -> np.ndarray:
        """Stored rows in the index dtype: np.uint16 bf16 bits, np.uint8 e4m3 codes or np.float32, [n, d]."""
        n = self.ntotal - row0 if n is None else n
        npdt = {_lib.DTYPE_BF16: np.uint16, _lib.DTYPE_FP8_E4M3: np.uint8, _lib.DTYPE_FP8_E4M3_DOCS: np.uint8, _lib.DTYPE_F32: np.float32}[self._code]
        out = np.empty((n, self._d), dtype=npdt)
        _lib.check(self._lib.mips_index_read_rows(self._h, int(row0), int(n), out.ctypes.data,
                                                  _stream_handle(self.device)), "mips_index_read_rows")
        return out

    def rows_bf16(self, row0: int = 0, n: int | None = None) -> np.ndarray:
        """Stored rows as bf16 bit patterns, np.uint16 [n, d]."""
        if self._code != _lib.DTYPE_BF16:
            raise TypeError("rows_bf16 on an fp8 index: use rows_raw")
        n = self.ntotal - row0 if n is None else n
        out = np.empty((n, self._d), dtype=np.uint16)
        _lib.check(self._lib.mips_index_read_rows(self._h, int(row0), int(n), out.ctypes.data,
                                                  _stream_handle(self.device)), "mips_index_read_rows")
        return out

    # ------------------------------------------------------------------ search
    def search(self, x, k: int, idx_offset: int = 0, force_ip: bool = False, tail_stream=None):
        """faiss Index.search(x, k) -> (D, I)  (sotasum/mips.py:383-386).
        NumPy in -> NumPy out; torch CUDA tensor in -> torch CUDA tensors out (stream-ordered, no
        synchronisation).  tail_stream (a torch.cuda.Stream, CUDA tensors only): candidate selection and exact
        re-score run there instead of on the current stream (mips_search_split) -- the results are complete on
        THAT stream."""
        import torch

        k = int(k)
        if k < 0:
            raise ValueError("k must be >= 0")
        if k > _lib.MAX_K:
            raise NotImplementedError(f"k = {k} > {_lib.MAX_K} is not supported by this build")
        ptr, code, is_dev, nq, keep = self._as_buffer(x, "search")
        stream = _stream_handle(self.device)
        if is_dev:
            dev = f"cuda:{self.device}"
            D = torch.empty((nq, k), dtype=torch.float32, device=dev)
            I = torch.empty((nq, k), dtype=torch.int64, device=dev)
            flags = _lib.Q_DEVICE | _lib.OUT_DEVICE
            ds, di = D.data_ptr(), I.data_ptr()
        else:
            D = np.empty((nq, k), dtype=np.float32)
            I = np.empty((nq, k), dtype=np.int64)
            flags = 0
            ds, di = D.ctypes.data, I.ctypes.data
        if force_ip:
            flags |= _lib.FORCE_IP  # inner-product ranking on an L2 index (Mips.np_search)
        with self._mutex:
            if tail_stream is not None and is_dev:
                D.record_stream(tail_stream)
                I.record_stream(tail_stream)
                _lib.check(self._lib.mips_search_split(self._h, ptr, code, nq, k, ds, di, int(idx_offset), flags, stream,
                                                       int(tail_stream.cuda_stream)), "mips_search_split")
            else:
                _lib.check(self._lib.mips_search(self._h, ptr, code, nq, k, ds, di, int(idx_offset), flags, stream),
                           "mips_search")
        del keep
        return D, I

    def search_fused(self, x, k: int, normalize: bool = False, ignore=None, idx_offset: int = 0):
        """The scoring hook's search in one call on CUDA tensors (include/mips_hip.h, mips_search_fused): optional
        row normalisation of float32 queries (the caller's tensor is not modified), top-k, and the ignore filter of
        sotasum/mips.py:388-398 (`ignore`: int64 ids, one per query).  One kernel launch for <= 16 queries on a small
        index; nothing synchronises."""
        import torch

        k = int(k)
        ptr, code, is_dev, nq, keep = self._as_buffer(x, "search")
        if not is_dev:
            raise ValueError("search_fused needs a CUDA tensor")
        dev = f"cuda:{self.device}"
        ig = None
        if ignore is not None:
            ig = torch.as_tensor(ignore, device=dev, dtype=torch.int64).contiguous()
            if ig.shape != (nq,):
                raise ValueError(f"ignore_indexes: expected {nq} ids, got {tuple(ig.shape)}")
        D = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I = torch.empty((nq, k), dtype=torch.int64, device=dev)
        with self._mutex:
            _lib.check(self._lib.mips_search_fused(self._h, ptr, code, nq, k, int(bool(normalize)),
                                                   ig.data_ptr() if ig is not None else None, D.data_ptr(), I.data_ptr(),
                                                   int(idx_offset), _stream_handle(self.device)), "mips_search_fused")
        del keep, ig
        return D, I

    def search_packed(self, x, k: int, idx_offset: int = 0, tail_stream=None):
        """Device-only search returning the all-gather payload: CUDA int64 [nq, k, 2] =
        {float32 score bits, index + idx_offset} (sharded.py).  tail_stream: as in search()."""
        import torch

        k = int(k)
        if k > _lib.MAX_K:
            raise NotImplementedError(f"k = {k} > {_lib.MAX_K} is not supported by this build")
        ptr, code, is_dev, nq, keep = self._as_buffer(x, "search")
        if not is_dev:
            raise ValueError("search_packed needs a CUDA tensor")
        out = torch.empty((nq, k, 2), dtype=torch.int64, device=f"cuda:{self.device}")
        flags = _lib.Q_DEVICE | _lib.OUT_DEVICE | _lib.OUT_PACKED
        with self._mutex:
            if tail_stream is not None:
                out.record_stream(tail_stream)
                _lib.check(self._lib.mips_search_split(self._h, ptr, code, nq, k, None, out.data_ptr(), int(idx_offset), flags,
                                                       _stream_handle(self.device), int(tail_stream.cuda_stream)),
                           "mips_search_split")
            else:
                _lib.check(self._lib.mips_search(self._h, ptr, code, nq, k, None, out.data_ptr(), int(idx_offset), flags,
                                                 _stream_handle(self.device)), "mips_search")
        del keep
        return out

    def set_param(self, name: str, value: int) -> None:
        """Launch tuning knob ("nsplit", "qgroups", "variant", "f32_fast": two-stage search of an fp32-exact index,
        0 off / 1 when the call may synchronise / 2 always); never changes certified results.  ("spin_limit" is the
        test-only bound of the scan's block barrier, include/mips_hip.h.)"""
        _lib.check(self._lib.mips_index_set_param(self._h, name.encode(), int(value)), "mips_index_set_param")

    def check(self, synchronize: bool = True) -> None:
        """Raise RuntimeError if a scan kernel of an earlier search on this index gave up on its block barrier
        (that search returned idx = IDX_POISON / NaN in every slot).  synchronize=True first waits for the
        current stream, so every search enqueued so far is covered; False only looks at the host-visible flag.
        Device-output searches never synchronise by themselves: call this where a sync is affordable (end of a
        batch, before results leave the process).  The next search on the index performs the same check."""
        _lib.check(self._lib.mips_index_check_error(self._h, int(bool(synchronize)), _stream_handle(self.device)),
                   "mips_index_check_error")

    def margin_stats(self, synchronize: bool = True) -> dict:
        """Margin check of the LAST search (include/mips_hip.h, mips_index_margin_stats): {"flagged": queries whose
        candidate pool was not provably wide enough, "rescanned": of those settled (exactly, by the brute-force pass, or by the
        re-scan with the widest lists), "unresolved": left with their first result}.  set_param("margin_check", m): 0 off,
        1 (default) certify -- NumPy searches with a synchronisation, CUDA searches stream-ordered, without one --,
        2 certify and synchronise, 3 stream-ordered explicitly, 4 count only."""
        f, r, u = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        _lib.check(self._lib.mips_index_margin_stats(self._h, ctypes.byref(f), ctypes.byref(r), ctypes.byref(u),
                                                     int(bool(synchronize)), _stream_handle(self.device)),
                   "mips_index_margin_stats")
        return {"flagged": f.value, "rescanned": r.value, "unresolved": u.value}

    @property
    def last_kernel(self) -> str:
        """Scan-kernel instance the last search dispatched to (rocprofv3 spelling)."""
        return self._lib.mips_index_last_kernel(self._h).decode()

    def scan_timing(self, reset: bool = False):
        """(summed ms, launches) of the fused scan kernel in the current measurement window, from HIP events
        on the search stream (synchronise first).  reset=True opens a new window (at most 128 launches);
        outside a window no events are recorded."""
        ms, cnt = ctypes.c_float(), ctypes.c_int()
        _lib.check(self._lib.mips_scan_timing(self._h, ctypes.byref(ms), ctypes.byref(cnt), int(reset)),
                   "mips_scan_timing")
        return ms.value, cnt.value

    def last_scan_ms(self) -> float:
        """Mean scan-kernel time of the searches since scan_timing(reset=True) opened a measurement window
        (-1.0 outside a window: events are not recorded then)."""
        import torch

        torch.cuda.synchronize(self.device)
        ms, cnt = self.scan_timing()
        return ms / cnt if cnt else -1.0

    # ------------------------------------------------------------------ persistence
    # Own format (the on-disk format is free, the call surface is kept):
    #   <path>/meta.json   {"format":1,"d":..,"ntotal":..,"metric":..,"dtype":"bf16", ...}
    #   <path>/rows.bf16   raw little-endian bf16 bit patterns [ntotal, d]   (rows.e4m3: e4m3 bytes)
    def save(self, path: str, extra: dict | None = None, chunk_rows: int = 1 << 16) -> None:
        """Replaces Dataset.save_faiss_index (sotasum/mips.py:536)."""
        os.makedirs(path, exist_ok=True)
        n = self.ntotal
        with open(os.path.join(path, "rows." + {"bf16": "bf16", "fp8_e4m3": "e4m3", "fp8_e4m3_docs": "e4m3", "f32": "f32"}[self.dtype]), "wb") as f:
            for r0 in range(0, n, chunk_rows):
                f.write(self.rows_raw(r0, min(chunk_rows, n - r0)).tobytes())
        meta = {"format": _FORMAT_VERSION, "d": self._d, "ntotal": n, "metric": self._metric, "dtype": self.dtype}
        if self._metric == _lib.METRIC_L2 and n > 0:
            # phi of the WHOLE file: a rank that loads one row range of it must not fall back to its own rows' maximum
            # (L2 distances of different shards would not be comparable)
            meta["phi"] = self.phi()
        if extra:
            meta.update({k: v for k, v in extra.items() if v is not None or k not in meta})
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def load(cls, path: str, device: int | None = None, row_range: tuple | None = None,
             chunk_rows: int = 1 << 16) -> "MipsIndex":
        """Replaces Dataset.load_faiss_index (sotasum/mips.py:547).  row_range=(lo, hi) loads one
        row shard of the file (multi-GPU: every rank maps the same file and keeps its own rows)."""
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        if meta.get("format") != _FORMAT_VERSION:
            raise ValueError(f"unknown index format {meta.get('format')}")
        ix = cls(meta["d"], metric=meta["metric"], dtype=meta["dtype"], device=device)
        n, d = meta["ntotal"], meta["d"]
        lo, hi = (0, n) if row_range is None else row_range
        if hi > lo:
            ext, npdt = {"bf16": ("bf16", np.uint16), "fp8_e4m3": ("e4m3", np.uint8), "fp8_e4m3_docs": ("e4m3", np.uint8),
                         "f32": ("f32", np.float32)}[meta["dtype"]]
            mm = np.memmap(os.path.join(path, "rows." + ext), dtype=npdt, mode="r", shape=(n, d))
            ix.reserve(hi - lo)
            for r0 in range(lo, hi, chunk_rows):
                ix.add(np.asarray(mm[r0:min(hi, r0 + chunk_rows)]))
            del mm
        if row_range is not None and meta["metric"] == _lib.METRIC_L2 and meta.get("phi") is not None:
            ix.set_phi(float(meta["phi"]))  # the file's phi, not this row range's
        ix.meta = meta
        return ix


def synth_fill(n: int, d: int, row0: int, seed: int, kind: int, dtype="bf16", device: int | None = None):
    """Device tensor [n, d] of generator values: torch.bfloat16 ("bf16"), torch.float32 ("f32") or
    e4m3 codes as torch.uint8 ("fp8_e4m3")."""
    import torch

    lib = _lib.load()
    dev = _lib.require_gpu(device)
    tdt = {"bf16": torch.bfloat16, "f32": torch.float32, "fp8_e4m3": torch.uint8}[dtype]
    code = {"bf16": _lib.DTYPE_BF16, "f32": _lib.DTYPE_F32, "fp8_e4m3": _lib.DTYPE_FP8_E4M3}[dtype]
    out = torch.empty((n, d), dtype=tdt, device=f"cuda:{dev}")
    _lib.check(lib.mips_synth_fill(out.data_ptr(), n, d, row0, seed, kind, code, dev, _stream_handle(dev)),
               "mips_synth_fill")
    return out


def l2_normalize_(x):
    """In-place row normalisation of a CUDA float32 tensor (faiss.normalize_L2 semantics,
    sotasum/mips.py:521-525)."""
    import torch

    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2):
        raise ValueError("l2_normalize_: expected a contiguous CUDA float32 matrix")
    lib = _lib.load()
    _lib.check(lib.mips_l2_normalize(x.data_ptr(), x.shape[0], x.shape[1], x.device.index,
                                     _stream_handle(x.device.index)), "mips_l2_normalize")
    return x


def rows_max_sumsq(x) -> float:
    """max_i |x_i|^2 of a CUDA float32 matrix (its sqrt is max_norm, sotasum/mips.py:298-304)."""
    import torch

    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2):
        raise ValueError("rows_max_sumsq: expected a contiguous CUDA float32 matrix")
    lib = _lib.load()
    out = ctypes.c_double()
    _lib.check(lib.mips_rows_max_sumsq(x.data_ptr(), x.shape[0], x.shape[1], ctypes.byref(out), x.device.index,
                                       _stream_handle(x.device.index)), "mips_rows_max_sumsq")
    return out.value


def rows_max_sumsq_into(x, acc) -> None:
    """acc = max(acc, max_i |x_i|^2): `acc` is a CUDA float64 tensor of one element that stays on the device -- no host copy,
    no synchronisation (the running max-norm of a streaming index build; read it once, at the end)."""
    import torch

    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2):
        raise ValueError("rows_max_sumsq_into: expected a contiguous CUDA float32 matrix")
    if not (isinstance(acc, torch.Tensor) and acc.is_cuda and acc.dtype == torch.float64 and acc.numel() == 1 and acc.device == x.device):
        raise ValueError("rows_max_sumsq_into: the accumulator must be a one-element CUDA float64 tensor on the rows' device")
    lib = _lib.load()
    _lib.check(lib.mips_rows_max_sumsq_device(x.data_ptr(), x.shape[0], x.shape[1], acc.data_ptr(), x.device.index,
                                              _stream_handle(x.device.index)), "mips_rows_max_sumsq_device")


def merge_topk_packed(gathered, nq: int, parts: int, k: int, metric: int = _lib.METRIC_IP):
    """Device merge straight from the gathered payload: CUDA int64 [parts * nq, k, 2] (rank-major)."""
    import torch

    lib = _lib.load()
    dev = gathered.device.index
    out_s = torch.empty((nq, k), dtype=torch.float32, device=gathered.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=gathered.device)
    _lib.check(lib.mips_merge_topk_packed(gathered.data_ptr(), nq, parts, k, metric, out_s.data_ptr(), out_i.data_ptr(),
                                          dev, _stream_handle(dev)), "mips_merge_topk_packed")
    return out_s, out_i


def merge_topk(cand_s, cand_i, parts: int, k: int, metric: int = _lib.METRIC_IP):
    """Device merge of `parts` per-shard top-k lists: cand_* CUDA [nq, parts*k] -> ([nq,k], [nq,k])."""
    import torch

    lib = _lib.load()
    nq = cand_s.shape[0]
    if cand_s.shape != cand_i.shape or cand_s.shape[1] != parts * k:
        raise ValueError(f"merge_topk: candidates {tuple(cand_s.shape)} / {tuple(cand_i.shape)} != [nq, parts*k = {parts * k}]")
    cand_s = cand_s.contiguous()
    cand_i = cand_i.contiguous()
    dev = cand_s.device.index
    out_s = torch.empty((nq, k), dtype=torch.float32, device=cand_s.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=cand_s.device)
    _lib.check(lib.mips_merge_topk(cand_s.data_ptr(), cand_i.data_ptr(), nq, parts, k, metric, out_s.data_ptr(),
                                   out_i.data_ptr(), dev, _stream_handle(dev)), "mips_merge_topk")
    return out_s, out_i


def filter_ignore(scores, idx, ignore, k: int):
    """Device form of the ignore filter of sotasum/mips.py:388-398: CUDA scores/idx [nq, k+1] and
    ignore [nq] -> ([nq, k], [nq, k])."""
    import torch

    lib = _lib.load()
    nq, k1 = scores.shape
    dev = scores.device.index
    scores, idx = scores.contiguous(), idx.contiguous()
    ignore = torch.as_tensor(ignore, device=scores.device, dtype=torch.int64).contiguous()
    if ignore.shape != (nq,):
        raise ValueError(f"ignore_indexes: expected {nq} ids, got {tuple(ignore.shape)}")
    out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    _lib.check(lib.mips_filter_ignore(scores.data_ptr(), idx.data_ptr(), ignore.data_ptr(), nq, k1, k, out_s.data_ptr(),
                                      out_i.data_ptr(), dev, _stream_handle(dev)), "mips_filter_ignore")
    return out_s, out_i


def _cosine_rescore_raw(query, mips_cls, memory_seq_len: int):
    """One launch: float32 scores [B, k] (+ memory_bias [B, k * memory_seq_len]) of CUDA query [B, d], cls [B, k, d]."""
    import torch

    lib = _lib.load()
    b, k, d = mips_cls.shape
    code = _lib.DTYPE_BF16 if mips_cls.dtype == torch.bfloat16 else _lib.DTYPE_F32
    dev = query.device.index
    out = torch.empty((b, k), dtype=torch.float32, device=query.device)
    if memory_seq_len == 0:
        _lib.check(lib.mips_cosine_rescore(query.data_ptr(), mips_cls.data_ptr(), code, b, k, d, out.data_ptr(), dev,
                                           _stream_handle(dev)), "mips_cosine_rescore")
        return out, None
    bias = torch.empty((b, k * memory_seq_len), dtype=torch.float32, device=query.device)
    _lib.check(lib.mips_cosine_rescore_bias(query.data_ptr(), mips_cls.data_ptr(), code, b, k, d, out.data_ptr(),
                                            memory_seq_len, bias.data_ptr(), dev, _stream_handle(dev)),
               "mips_cosine_rescore_bias")
    return out, bias


def _make_cosine_function():
    import torch

    class _CosineRescore(torch.autograd.Function):
        """Autograd wrapper of the hook's re-score: backward = the reference's (retriever_generator.py:158-172) --
        gradient of `query @ mips_cls.transpose(1, 2)` only, the norms are constants (they are computed under
        torch.no_grad() there).  memory_bias is an expand of the scores, its gradient folds back over the tokens."""

        @staticmethod
        def forward(ctx, query, mips_cls, memory_seq_len):
            ctx.save_for_backward(query, mips_cls)
            ctx.memory_seq_len = int(memory_seq_len)
            out, bias = _cosine_rescore_raw(query, mips_cls, ctx.memory_seq_len)
            if bias is None:
                return out
            return out, bias

        @staticmethod
        def backward(ctx, g_scores, g_bias=None):
            query, mips_cls = ctx.saved_tensors
            lib = _lib.load()
            b, k, d = mips_cls.shape
            code = _lib.DTYPE_BF16 if mips_cls.dtype == torch.bfloat16 else _lib.DTYPE_F32
            dev = query.device.index
            gs = g_scores.float().contiguous() if g_scores is not None else None
            gb = g_bias.float().contiguous() if (g_bias is not None and ctx.memory_seq_len > 0) else None
            gq = torch.empty((b, d), dtype=torch.float32, device=query.device)
            gc = torch.empty((b, k, d), dtype=torch.float32, device=query.device)
            if gs is None and gb is None:
                return None, None, None
            _lib.check(lib.mips_cosine_rescore_backward(query.data_ptr(), mips_cls.data_ptr(), code, b, k, d,
                                                        gs.data_ptr() if gs is not None else None,
                                                        gb.data_ptr() if gb is not None else None, ctx.memory_seq_len,
                                                        gq.data_ptr(), gc.data_ptr(), dev, _stream_handle(dev)),
                       "mips_cosine_rescore_backward")
            return gq.to(query.dtype), gc.to(mips_cls.dtype), None

    return _CosineRescore


_COSINE_FN = None


def cosine_rescore(query, mips_cls, memory_seq_len: int = 0):
    """retriever_generator.py:158-172 on the device: query [B, 1, d] or [B, d], mips_cls [B, k, d]
    (CUDA float32 or bfloat16) -> float32 [B, k] = q . c / (|q| |c|).  With memory_seq_len > 0 the
    same launch also writes the hook's memory_bias (retriever_generator.py:188-192), float32
    [B, k * memory_seq_len], and (scores, memory_bias) is returned.
    Differentiable like the reference's expression: when an input requires grad the call goes through a
    torch.autograd.Function whose backward (a HIP kernel as well) is the gradient of the dot product with the
    norms held constant -- memory_bias is how the retriever's encoders are trained."""
    import torch

    global _COSINE_FN
    squeeze_grad_shape = query.dim() == 3
    if squeeze_grad_shape:
        query = query[:, 0, :]
    b, k, d = mips_cls.shape
    if query.shape != (b, d) or not (query.is_cuda and mips_cls.is_cuda):
        raise ValueError("cosine_rescore: expected CUDA query [B,(1,)d] and mips_cls [B,k,d]")
    if memory_seq_len < 0:
        raise ValueError("cosine_rescore: memory_seq_len must be >= 0")
    if not (mips_cls.dtype == torch.bfloat16 and query.dtype == torch.bfloat16):
        query, mips_cls = query.float(), mips_cls.float()
    query, mips_cls = query.contiguous(), mips_cls.contiguous()
    if torch.is_grad_enabled() and (query.requires_grad or mips_cls.requires_grad):
        if _COSINE_FN is None:
            _COSINE_FN = _make_cosine_function()
        return _COSINE_FN.apply(query, mips_cls, int(memory_seq_len))
    out, bias = _cosine_rescore_raw(query, mips_cls, int(memory_seq_len))
    return out if bias is None else (out, bias)
