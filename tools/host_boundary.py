"""PCIe-inclusive rates of the drop-in call shape (GPU box): NumPy float32 queries in, NumPy results out,
exactly what `faiss_index.search(queries, k)` hands over at sotasum/mips.py:383-386.  Not bench.py's `value`
(that one starts with the queries resident in HBM); DESIGN.md section 5 quotes these numbers.
usage: python tools/host_boundary.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import retrieval_augmented_mds_amd as ram


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(iters):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return float(np.median(t)), float(np.min(t))


out = []
for rows, nq, iters, name, dtype in ((1 << 20, 4096, 20, "BASELINE config 2", "bf16"),
                                     (10000, 8, 200, "BASELINE config 1 shape (bf16 storage, opt-in)", "bf16"),
                                     (10000, 8, 200, "BASELINE config 1 shape (fp32-exact storage: the facade's default)", "f32")):
    ix = ram.MipsIndex(768, dtype=dtype)
    if dtype == "f32":   # rows and queries that are NOT bf16 values (the synthetic generator emits bf16 values)
        g = torch.Generator(device="cuda").manual_seed(0xD0C5)
        ix.add(torch.randn(rows, 768, device="cuda", generator=g))
        q_dev = torch.randn(nq, 768, device="cuda", generator=g)
    else:
        ix.add_synthetic(rows, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
        q_dev = ram.synth_fill(nq, 768, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS)
    q_host = q_dev.float().cpu().numpy()
    med_h, min_h = timed(lambda: ix.search(q_host, 5), iters)

    def dev_call():
        ix.search(q_dev, 5); torch.cuda.synchronize()
    med_d, min_d = timed(dev_call, iters)
    rec = {"workload": f"{name}: {rows}x768 {dtype} index, Q={nq}, k=5", "kernel": ix.last_kernel, "margin": ix.margin_stats(),
           "host_numpy_f32_in_out_ms": {"median": med_h * 1e3, "min": min_h * 1e3}, "host_queries_per_s": nq / med_h,
           "device_resident_ms": {"median": med_d * 1e3, "min": min_d * 1e3}, "device_queries_per_s": nq / med_d}
    if nq <= 16:
        # the same call through the general path (separate launches), and the GPU time alone from HIP events
        def gpu_ms(fn, n=200):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n
        rec["one_launch_gpu_ms_per_call_back_to_back"] = gpu_ms(lambda: ix.search(q_dev, 5))
        ig = torch.arange(nq, device="cuda")
        rec["fused_hook_normalize_ignore_gpu_ms"] = gpu_ms(lambda: ix.search_fused(q_dev.float(), 5, normalize=True, ignore=ig))
        ix.set_param("margin_check", 4)   # count only: what the certificate's launches (exact pass behind the one launch) cost
        if dtype != "f32":                # (the fp32-exact index takes the one-launch kernel only when it certifies)
            med_c, min_c = timed(dev_call, iters)
            rec["count_only_device_resident_ms"] = {"median": med_c * 1e3, "min": min_c * 1e3}
            rec["count_only_gpu_ms_per_call_back_to_back"] = gpu_ms(lambda: ix.search(q_dev, 5))
        ix.set_param("margin_check", 1)
        ix.set_param("tiny", 0)
        med_g, min_g = timed(dev_call, iters)
        rec["general_path_device_resident_ms"] = {"median": med_g * 1e3, "min": min_g * 1e3}
        rec["general_path_gpu_ms_per_call_back_to_back"] = gpu_ms(lambda: ix.search(q_dev, 5))
        rec["general_path_kernel"] = ix.last_kernel
        ix.set_param("tiny", 1)
    out.append(rec)
print(json.dumps(out, indent=1))
