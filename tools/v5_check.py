import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import retrieval_augmented_mds_amd as ram
from oracle import mips_oracle as orc, synth
ok = True
for n, nq, d in ((150001, 700, 768), (40000, 300, 768), (5000, 520, 768), (64 * 37 + 5, 300, 512), (33, 400, 768), (100003, 1024, 768)):
    ix = ram.MipsIndex(d)
    ix.add_synthetic(n, row0=0, seed=171, kind=synth.KIND_GAUSS)
    q = ram.synth_fill(nq, d, 0, 172, synth.KIND_GAUSS)
    ix.set_param("variant", 4)
    rs, ri = ix.search(q, 5)
    ix.set_param("variant", 5)
    s, i = ix.search(q, 5)
    torch.cuda.synchronize(); ix.check()
    same = torch.equal(i, ri) and torch.equal(s, rs)
    print(n, nq, d, ix.last_kernel, "same as v4:", same, flush=True)
    ok &= same
    for ns in (8, 40):
        ix.set_param("nsplit", ns)
        s, i = ix.search(q, 5)
        ok &= torch.equal(i, ri) and torch.equal(s, rs)
    ix.set_param("nsplit", 0)
x = synth.generate(5, 0, 3000, 768, synth.KIND_LATTICE); ql = synth.generate(6, 0, 300, 768, synth.KIND_LATTICE)
x[10] = x[700]; x[333] = x[700]; ql[0] = x[700]
ix = ram.MipsIndex(768); ix.add(x); ix.set_param("variant", 5)
s, i = ix.search(ql, 5)
es, ei = orc.search_exact_bruteforce(ql, x, 5)
print("lattice ties:", np.array_equal(i, ei) and np.array_equal(s, es), ix.last_kernel)
ok &= np.array_equal(i, ei) and np.array_equal(s, es)
print("ALL OK" if ok else "MISMATCH")
