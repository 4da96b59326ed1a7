"""MI355X-native dense retrieval (exact MIPS) backend for sotasum's `Mips` path.

    import retrieval_augmented_mds_amd as ram          # alias module at the repo root
    index = ram.MipsIndex(768); index.add(x); D, I = index.search(q, 5)
    mips = ram.Mips(ram.MipsArgs(mips_topk=5), data={"mips_column": texts, "aid": aids})

Everything numeric runs in libmips_hip.so (hand-written HIP for gfx950, csrc/); there is no CPU
fallback.  See DESIGN.md and include/mips_hip.h.
"""
from . import _lib, faiss_shim
from ._lib import (DTYPE_BF16, DTYPE_F32, DTYPE_FP8_E4M3, DTYPE_FP8_E4M3_DOCS, IDX_POISON, MAX_K, METRIC_IP, METRIC_L2, SEED_DOCS, SEED_QUERIES, SYNTH_GAUSS,
                   SYNTH_LATTICE, SYNTH_LATTICE_FP8, build)
from .index import (MipsIndex, cosine_rescore, filter_ignore, l2_normalize_, merge_topk, merge_topk_packed,
                    rows_max_sumsq, synth_fill)
from .mips import (KnowledgeBase, Mips, MipsArgs, MipsModelOutput, augment_xb, augment_xq, get_phi,
                   in_batch_scores, inner_product, retriever_metrics)
from .sharded import ShardedMipsIndex, pack_topk, shard_bounds, unpack_gathered

__all__ = [
    "MipsIndex", "ShardedMipsIndex", "Mips", "MipsArgs", "MipsModelOutput", "KnowledgeBase",
    "get_phi", "augment_xb", "augment_xq", "inner_product", "in_batch_scores", "retriever_metrics", "IDX_POISON",
    "l2_normalize_", "rows_max_sumsq", "merge_topk", "merge_topk_packed", "filter_ignore", "cosine_rescore", "synth_fill", "shard_bounds", "pack_topk",
    "unpack_gathered", "build", "METRIC_IP", "METRIC_L2", "MAX_K", "faiss_shim",
]
