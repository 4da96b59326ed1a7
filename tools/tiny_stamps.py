"""Phase time stamps of tiny_search_kernel (experimental build only: MIPS_HIP_EXPERIMENTAL=1 MIPS_TINY_DBG=1).
    MIPS_HIP_EXPERIMENTAL=1 MIPS_TINY_DBG=1 python tools/tiny_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import retrieval_augmented_mds_amd as ram

for rows, dtype in ((10000, "bf16"), (65536, "bf16"), (10000, "f32"), (65536, "f32")):
    ix = ram.MipsIndex(768, dtype=dtype)
    ix.add_synthetic(rows, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
    q = ram.synth_fill(8, 768, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS)
    print("rows", rows, dtype, file=sys.stderr, flush=True)
    for _ in range(6):
        ix.search(q, 5)
        torch.cuda.synchronize()
    qf = q.float()
    ig = torch.arange(8, device="cuda")
    print("fused (normalise + ignore)", file=sys.stderr, flush=True)
    for _ in range(3):
        ix.search_fused(qf, 5, normalize=True, ignore=ig)
        torch.cuda.synchronize()
