"""Row-sharded search over the GPUs of one node.

The reference replicates the whole FAISS index on every rank (lightning_model.py:180,
mips.py:545-549).  Here rank r of G keeps rows [r*ceil(N/G), min(N, (r+1)*ceil(N/G))) in its own
HBM, every rank scores ALL queries against its shard with the fused HIP kernel, and ONE collective
-- an all-gather (RCCL over xGMI when the backend is "nccl") of the packed per-rank top-k -- is
followed by a replicated k-way merge, identical on every rank.  No other data-path collective.

The all-gather payload is one int64 tensor [nq, k, 2] per rank (float32 score bits, global row id):
4096 x 5 x 16 B = 320 KiB at the headline config -- latency bound on xGMI.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous row partition, ceil(N/G) rows per rank (same contiguous-chunk idea as the
    reference's encode sharding, sotasum/mips.py:227-229)."""
    per = -(-int(n) // int(world))
    lo = min(int(n), rank * per)
    hi = min(int(n), (rank + 1) * per)
    return lo, hi


def pack_topk(scores, idx):
    """(float32 [nq,k], int64 [nq,k]) torch tensors -> int64 [nq,k,2]."""
    import torch

    bits = scores.contiguous().view(torch.int32).to(torch.int64)
    return torch.stack((bits, idx.to(torch.int64)), dim=-1).contiguous()


def unpack_gathered(gathered, world: int):
    """int64 [world, nq, k, 2] -> (float32 [nq, world*k], int64 [nq, world*k]), shard-major rows."""
    import torch

    nq, k = gathered.shape[1], gathered.shape[2]
    g = gathered.permute(1, 0, 2, 3).reshape(nq, world * k, 2)
    s = g[..., 0].to(torch.int32).contiguous().view(torch.float32)
    return s, g[..., 1].contiguous()


class ShardedMipsIndex:
    """One logical exact index, row-sharded across the ranks of a torch.distributed group.

    local_search(q, k, idx_offset) and merge(cand_s, cand_i, parts, k, metric) default to the HIP
    path (MipsIndex.search / mips_merge_topk).  They are injectable so that the partition + pack +
    all-gather + unpack plumbing can be exercised with the gloo backend on CPU-only machines, where
    tests substitute the CPU oracle for the two device steps.
    """

    def __init__(self, d: int, metric: int = _lib.METRIC_IP, dtype: str = "bf16", group=None, device=None,
                 local_search=None, merge=None):
        import torch.distributed as dist

        self.d = int(d)
        self.metric_type = int(metric)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.ntotal_global = 0
        self.lo = self.hi = 0
        self.local = None
        if local_search is None:
            from .index import MipsIndex

            self.local = MipsIndex(d, metric=metric, dtype=dtype, device=device)
            local_search = self.local.search
        self._fast = self.local is not None and merge is None  # both device steps are the library's own
        self._device_merge = merge is None  # mips_merge_topk: candidates must be on the GPU whatever moved them
        if merge is None:
            from .index import merge_topk as merge
        self._local_search = local_search
        self._merge = merge

    # ------------------------------------------------------------------ building
    def set_global_size(self, n: int):
        self.ntotal_global = int(n)
        self.lo, self.hi = shard_bounds(n, self.world, self.rank)
        return self.lo, self.hi

    def add_global(self, x) -> None:
        """Every rank sees the full [N, d] array (NumPy / memmap / torch) and keeps its own rows."""
        lo, hi = self.set_global_size(len(x))
        if self.local is not None and hi > lo:
            self.local.reserve(hi - lo)
            self.local.add(x[lo:hi])
        self._sync_phi()

    def add_synthetic_global(self, n: int, seed: int, kind: int) -> None:
        lo, hi = self.set_global_size(n)
        if hi > lo:
            self.local.reserve(hi - lo)
            self.local.add_synthetic(hi - lo, row0=lo, seed=seed, kind=kind)
        self._sync_phi()

    def _sync_phi(self) -> None:
        """L2 mode: phi = max_i |x_i|^2 must be the GLOBAL maximum (mips.py:316-324 computes it over the
        whole knowledge base); one scalar all-reduce at build time, not on the search path."""
        import torch
        import torch.distributed as dist

        if self.metric_type != _lib.METRIC_L2 or self.local is None or self.world == 1:
            return
        self.local.clear_phi()  # a previous global value must not mask rows added since (incremental add_global)
        local = self.local.phi() if self.local.ntotal > 0 else 0.0
        backend = dist.get_backend(self.group)
        t = torch.tensor([local], dtype=torch.float64, device=f"cuda:{self.local.device}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.local.set_phi(float(t.item()))

    @property
    def ntotal(self) -> int:
        return self.ntotal_global

    @property
    def dtype(self):
        return self.local.dtype if self.local is not None else None

    nprobe = 1  # accepted like MipsIndex.nprobe (mips.py:342-345); exact search ignores it

    def phi(self) -> float:
        """The global phi (after _sync_phi every shard holds the same override)."""
        return self.local.phi()

    # ------------------------------------------------------------------ persistence
    def save(self, path: str, extra: dict = None) -> None:
        """Collective: the shards are written into ONE file set in rank order (rank 0 writes meta.json), so the
        result is the same directory MipsIndex.save produces for the unsharded index and can be loaded by any
        number of ranks (or by a single MipsIndex.load)."""
        import json
        import os

        import torch.distributed as dist

        ext = {"bf16": "bf16", "fp8_e4m3": "e4m3", "fp8_e4m3_docs": "e4m3", "f32": "f32"}[self.local.dtype]
        if self.rank == 0:
            os.makedirs(path, exist_ok=True)
            open(os.path.join(path, "rows." + ext), "wb").close()
        for r in range(self.world):  # append in rank order; barriers keep the order
            if self.world > 1:
                dist.barrier(group=self.group)
            if r == self.rank and self.local.ntotal > 0:
                with open(os.path.join(path, "rows." + ext), "ab") as f:
                    n = self.local.ntotal
                    for r0 in range(0, n, 1 << 16):
                        f.write(self.local.rows_raw(r0, min(1 << 16, n - r0)).tobytes())
        if self.world > 1:
            dist.barrier(group=self.group)
        if self.rank == 0:
            meta = {"format": 1, "d": self.d, "ntotal": self.ntotal_global, "metric": self.metric_type,
                    "dtype": self.local.dtype}
            if self.metric_type == _lib.METRIC_L2 and self.ntotal_global > 0:
                meta["phi"] = self.phi()
            if extra:
                meta.update({k: v for k, v in extra.items() if v is not None or k not in meta})
            with open(os.path.join(path, "meta.json"), "w") as f:
                json.dump(meta, f)
        if self.world > 1:
            dist.barrier(group=self.group)

    @classmethod
    def load(cls, path: str, group=None, device=None) -> "ShardedMipsIndex":
        """Every rank maps the same file set and keeps its own row range (MipsIndex.load(row_range)); an L2
        index takes the file's global phi.  Replaces `load()` on every rank of lightning_model.py:180."""
        import json
        import os

        from .index import MipsIndex

        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        self = cls.__new__(cls)
        ShardedMipsIndex.__init__(self, meta["d"], metric=meta["metric"], dtype=meta["dtype"], group=group, device=device,
                                  local_search=lambda *a, **k: None)  # placeholder: the local index comes from the file
        lo, hi = self.set_global_size(meta["ntotal"])
        self.local = MipsIndex.load(path, device=device, row_range=(lo, hi))
        self._local_search = self.local.search
        self._fast = True
        self.meta = meta
        if self.metric_type == _lib.METRIC_L2 and meta.get("phi") is None:
            self._sync_phi()  # files written before phi was persisted
        return self

    def set_param(self, name: str, value: int) -> None:
        """Knob of the LOCAL scan (MipsIndex.set_param) -- every rank sets its own.  The default margin mode already certifies
        the local device-output searches without synchronising (the exact pass of the queries they flag is enqueued behind the
        scan), which keeps the optimistic / two-stage paths (8 <= k <= 29, fp32-exact shards) open to the sharded search."""
        self.local.set_param(name, value)

    def margin_stats(self, synchronize: bool = True, reduce: bool = True) -> dict:
        """Margin statistics of the last search: the shards' counts (MipsIndex.margin_stats) SUMMED over the ranks -- a query
        left unresolved on any shard is unresolved in the merged result, so `unresolved` must be visible everywhere.  The sum
        is one small all-reduce: COLLECTIVE, every rank calls it (synchronize=True only; reduce=False or synchronize=False
        return this rank's own counts without communicating)."""
        st = self.local.margin_stats(synchronize)
        if not (reduce and synchronize and self.world > 1):
            return st
        import torch
        import torch.distributed as dist

        backend = dist.get_backend(self.group)
        t = torch.tensor([st["flagged"], st["rescanned"], st["unresolved"]], dtype=torch.int64,
                         device=f"cuda:{self.local.device}" if backend == "nccl" else "cpu")
        t = torch.clamp(t, min=0)  # (-1 = "only counted on the device" cannot survive a synchronising read)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        f, r, u = (int(v) for v in t.cpu().tolist())
        return {"flagged": f, "rescanned": r, "unresolved": u}

    def check(self, synchronize: bool = True) -> None:
        """MipsIndex.check for this rank's shard.  A shard whose scan timed out hands poisoned rows (idx -2, NaN)
        to the all-gather and the merge kernel propagates them, so every rank SEES the failure in its results;
        the rank it happened on also raises here (and on its next search)."""
        if self.local is not None:
            self.local.check(synchronize)

    # ------------------------------------------------------------------ pipelined search
    class _Pending:
        """Result of search_async: .result() makes the CURRENT stream wait for the merged top-k and returns it."""

        def __init__(self, out, done_event, keep, index=None):
            self._out, self._done, self._keep, self._index = out, done_event, keep, index

        def result(self):
            import torch

            if self._done is not None:
                cur = torch.cuda.current_stream(self._out[0].device)
                cur.wait_event(self._done)
                for t in self._out:
                    t.record_stream(cur)  # allocated on the side stream, consumed on this one
                self._done, self._keep = None, None
            if self._index is not None:
                self._index.check(synchronize=False)  # host-visible flag only: no synchronisation on this path
            return self._out

    def search_async(self, q, k: int, _force_collective: bool = False):
        """search() split over two streams so that consecutive, independent query batches overlap:
          current stream   query staging + the fused scan of this shard (mips_search_split);
          side stream      candidate selection + exact re-score, then the exchange step -- the ONE all-gather and the
                           replicated merge (ranks > 1).
        The caller can enqueue the NEXT batch's scan before asking for this batch's result(): the scan of batch t + 1
        starts right behind the scan of batch t, the ~45 us tail and the latency-bound collective (tens of us against a
        ~0.6 ms shard scan at 8 GPUs) run beside it.  Falls back to the synchronous path when there is nothing to
        overlap with (host queries, injected device steps, gloo)."""
        import torch
        import torch.distributed as dist

        backend = dist.get_backend(self.group) if dist.is_initialized() else None
        collective = self.world > 1 or _force_collective
        fast = (self.local is not None and self._fast and isinstance(q, torch.Tensor) and q.is_cuda
                and (backend == "nccl" or not collective))
        if not fast:
            return ShardedMipsIndex._Pending(self.search(q, k), None, None)
        from .index import merge_topk_packed

        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=q.device, priority=-1)  # its short kernels go first when CUs free up
        side = self._side
        main = torch.cuda.current_stream(q.device)
        if not collective:
            out = self.local.search(q, k, self.lo, tail_stream=side)  # scan here, select + re-score on the side stream
            done = torch.cuda.Event()
            done.record(side)
            return ShardedMipsIndex._Pending(out, done, (q,), self.local)
        packed = self.local.search_packed(q, k, self.lo, tail_stream=side)
        nq = packed.shape[0]
        ready = torch.cuda.Event()
        ready.record(main)  # (the one-launch kernel of tiny searches writes its results on the main stream)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            gathered = torch.empty((self.world * nq, k, 2), dtype=torch.int64, device=packed.device)
            dist.all_gather_into_tensor(gathered, packed, group=self.group)  # the ONE collective of the path
            out = merge_topk_packed(gathered, nq, self.world, k, self.metric_type)
            done = torch.cuda.Event()
            done.record(side)
        return ShardedMipsIndex._Pending(out, done, (packed, gathered, q), self.local)

    # ------------------------------------------------------------------ search
    def search(self, q, k: int, idx_offset: int = 0, force_ip: bool = False):
        """Replicated queries in, global top-k out (same on every rank).  force_ip: rank by inner product on an
        L2 index (Mips.np_search); idx_offset exists for signature compatibility with MipsIndex.search and must
        be 0 (global row numbers are the shard offsets' business)."""
        import torch
        import torch.distributed as dist

        if idx_offset:
            raise ValueError("ShardedMipsIndex.search returns global row numbers; idx_offset must be 0")
        if force_ip:
            return self._search_force_ip(q, k)
        backend = dist.get_backend(self.group) if self.world > 1 else None
        if (self.world > 1 and self.local is not None and self._fast and isinstance(q, torch.Tensor) and q.is_cuda):
            # device fast path: the re-score kernel writes the all-gather payload, the merge kernel reads
            # the gathered buffer as it arrives -- no tensor reshuffling between scan and collective
            from .index import merge_topk_packed

            packed = self.local.search_packed(q, k, self.lo)
            nq = packed.shape[0]
            if backend == "gloo":
                packed = packed.cpu()
            gathered = torch.empty((self.world * nq, k, 2), dtype=torch.int64, device=packed.device)
            dist.all_gather_into_tensor(gathered, packed, group=self.group)  # the ONE collective of the path
            if not gathered.is_cuda:
                gathered = gathered.to(q.device)
            return merge_topk_packed(gathered, nq, self.world, k, self.metric_type)
        s, i = self._local_search(q, k, self.lo)
        return self._exchange(s, i, k, self.metric_type)

    def _search_force_ip(self, q, k: int):
        s, i = self.local.search(q, k, self.lo, force_ip=True)
        return self._exchange(s, i, k, _lib.METRIC_IP)

    def _exchange(self, s, i, k: int, metric: int):
        """Generic form of the exchange step: pack the local top-k, ONE all-gather, unpack, merge."""
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return s, i
        backend = dist.get_backend(self.group)
        as_numpy = not isinstance(s, torch.Tensor)
        if as_numpy:
            s, i = torch.from_numpy(np.ascontiguousarray(s)), torch.from_numpy(np.ascontiguousarray(i))
        home = s.device
        if (backend == "nccl" or self._device_merge) and not s.is_cuda:
            home = torch.device(f"cuda:{self.local.device}" if self.local is not None else "cuda")
            s, i = s.to(home), i.to(home)
        packed = pack_topk(s, i)
        if backend == "gloo" and packed.is_cuda:
            packed = packed.cpu()  # rehearsal / CPU clusters: gloo moves host memory; RCCL takes the device tensor
        nq = packed.shape[0]
        # rank-major concatenation along dim 0 (the layout both RCCL and gloo accept)
        gathered = torch.empty((self.world * nq,) + tuple(packed.shape[1:]), dtype=torch.int64, device=packed.device)
        dist.all_gather_into_tensor(gathered, packed, group=self.group)  # the ONE collective of the path
        if gathered.device != home:
            gathered = gathered.to(home)
        cs, ci = unpack_gathered(gathered.view((self.world, nq) + tuple(packed.shape[1:])), self.world)
        out_s, out_i = self._merge(cs, ci, self.world, k, metric)
        if as_numpy:
            return out_s.cpu().numpy(), out_i.cpu().numpy()
        return out_s, out_i
