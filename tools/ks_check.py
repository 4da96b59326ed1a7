"""scan_kernel_ks (K split over a wave pair, pitch 1024) against scan_kernel_v3's 1024 configuration and the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import retrieval_augmented_mds_amd as ram
from oracle import mips_oracle as orc, synth
ok = True
for n, nq, d, k in ((70001, 300, 1024, 5), (33, 1, 1024, 3), (150001, 700, 1000, 5), (64 * 37 + 5, 129, 800, 4), (40000, 4096, 1024, 5)):
    ix = ram.MipsIndex(d)
    ix.add_synthetic(n, row0=0, seed=171, kind=synth.KIND_GAUSS)
    q = ram.synth_fill(nq, d, 0, 172, synth.KIND_GAUSS)
    ix.set_param("variant", 3)
    rs, ri = ix.search(q, k)
    name3 = ix.last_kernel
    ix.set_param("variant", 6)
    s, i = ix.search(q, k)
    torch.cuda.synchronize(); ix.check()
    same = torch.equal(i, ri) and torch.equal(s, rs)
    print(n, nq, d, k, name3, "->", ix.last_kernel, "same:", same, flush=True)
    ok &= same
    for ns in (8, 40):
        ix.set_param("nsplit", ns)
        s, i = ix.search(q, k)
        ok &= torch.equal(i, ri) and torch.equal(s, rs)
    ix.set_param("nsplit", 0)
    if n <= 70001:
        x = synth.generate(171, 0, n, d, synth.KIND_GAUSS)
        es, ei = orc.search_exact(q.float().cpu().numpy(), x, k)
        okk = np.array_equal(i.cpu().numpy(), ei) and np.array_equal(s.cpu().numpy(), es)
        print("   oracle:", okk); ok &= okk
x = synth.generate(5, 0, 3000, 1024, synth.KIND_LATTICE); ql = synth.generate(6, 0, 300, 1024, synth.KIND_LATTICE)
x[10] = x[700]; x[333] = x[700]; ql[0] = x[700]
ix = ram.MipsIndex(1024); ix.add(x); ix.set_param("variant", 6)
s, i = ix.search(ql, 5)
es, ei = orc.search_exact_bruteforce(ql, x, 5)
print("lattice ties:", np.array_equal(i, ei) and np.array_equal(s, es), ix.last_kernel)
ok &= np.array_equal(i, ei) and np.array_equal(s, es)
print("ALL OK" if ok else "MISMATCH")
