#!/usr/bin/env python3
"""Headline benchmark: exact top-5 MIPS queries/second (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF (a child
`python -m torch.distributed.run ...` of this same file, spawned before this process touches the GPU), relays
the child's output and exits with its code -- so the same verb produces a 1-GPU and an N-GPU line.

A step = one pass of the hot path over one batch: Q = 4096 synthetic bf16 queries (already resident
in HBM) against the HBM-resident 2^20 x 768 bf16 index (BASELINE config 2), top-k = 5, through
MipsIndex.search -> mips_search (query staging into the padded tile buffer, fused MFMA score +
top-K scan, split merge + exact re-score).  With N > 1 the SAME index is row-sharded over the ranks
(rank r keeps rows [r*ceil(n/N), ...)), every rank scores all queries against its shard and one
RCCL all-gather + merge produces the replicated global top-k: total work is fixed => "strong".

One JSON line on rank 0.  `roofline` describes the dominant kernel (the fused scan; its name comes from the
library: mips_index_last_kernel): this workload has Q = 4096 flop per index byte, far above the ~310 flop/B
ridge, so the binding roof is MFMA; the HBM-read fraction the north star asks for is reported next to it
(hbm_* keys) and measured where it is meaningful in `regimes` (2^24 x 768 index at Q = 8: HBM-bound; `regimes` also carries
the fp32-exact index and k = 10 at the headline's size: the paths that rest on the per-query certificate).
`cpu_baseline` times the oracle's literal restatement of the reference's brute force (sotasum/mips.py:552-560:
fp32 matmul + full argsort), `cpu_baseline_torch` the reference's CPU torch idiom (retriever_lightning.py:304-305,
`topk(q @ d.T)`) on all host cores; rank 0, N = 1 only, bounded samples, >= 3 repeats, median.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0    # dense fp8 MFMA (block-scaled K=64 instruction)
PEAK_HBM_GBS = 8000.0       # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1 << 20, help="index rows in total (default: BASELINE config 2)")
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--cpu-queries", type=int, default=256, help="queries per repeat of the NumPy-port CPU baseline")
    ap.add_argument("--cpu-torch-queries", type=int, default=512, help="queries per repeat of the torch CPU baseline")
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-regimes", action="store_true", help="skip the extra regime measurements (2^24-row index; N > 1: BASELINE config 3 / 5 row-sharded)")
    ap.add_argument("--regime-rows", type=int, default=0,
                    help="N > 1: total rows of the row-sharded regime index (default 2^24 = BASELINE config 3 when --rows is the default, else 4 x --rows)")
    ap.add_argument("--index-dtype", default="bf16", choices=["bf16", "fp8_e4m3", "fp8_e4m3_docs"],
                    help="fp8_e4m3 = BASELINE config 5 with index and queries quantised to OCP e4m3 (fp8 MFMA); fp8_e4m3_docs = as BASELINE "
                         "words it: e4m3 documents, bf16 queries (bf16 MFMA after an exact up-conversion; meant for few queries per pass)")
    ap.add_argument("--no-pipeline", action="store_true", help="N > 1: run every step's all-gather + merge before the next scan")
    ap.add_argument("--pipeline-single", action="store_true", help="N = 1: split-tail pipelining of consecutive steps (ShardedMipsIndex.search_async)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N > 1 path on ONE GPU (all ranks on cuda:0, host-staged collective)")
    ap.add_argument("--launch-check", action="store_true",
                    help="form the N-rank process group, exchange one tensor, print what was formed and exit (no search, no GPU)")
    return ap.parse_args()


def self_launch(args) -> int:
    """--gpus N > 1 without WORLD_SIZE: start N ranks as a child torch.distributed.run of this file.  Nothing in
    this process has touched the GPU yet (only argparse ran), and nothing is exec'ed: the child is a subprocess and
    this process exits with its code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MIPS_BENCH_SELF_LAUNCHED"] = "1"
    print(f"[bench] --gpus {args.gpus}: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def cpu_baselines(index, q_dev, args):
    """Both CPU baselines on bounded samples of the SAME workload (values read back from HBM, up-cast to fp32),
    >= 3 repeats each, median reported:
      port   oracle.inner_product = the literal NumPy restatement of mips.py:552-560 (matmul + full argsort)
      torch  oracle.torch_topk    = torch.topk(q @ d.T, k) on all host cores, 256-query chunks over <= 2^20-row
                                    document blocks with a running top-k merge"""
    import numpy as np
    import torch

    from oracle import mips_oracle as orc
    from oracle import synth

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    x = synth.bf16_bits_to_f32(index.rows_raw()) if index.dtype == "bf16" else synth.e4m3_bits_to_f32(index.rows_raw())
    qa = q_dev.float().cpu().numpy()
    if index.dtype == "fp8_e4m3":
        qa = synth.round_to_e4m3(qa)  # the device quantises the queries the same way (fp8_e4m3_docs keeps them in bf16)
    reps = max(1, args.cpu_repeats)

    nq = min(args.cpu_queries, args.queries)
    orc.inner_product(qa[:2], x[:4096], k=args.k, normalize=False)  # warm-up
    times, res = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        res = orc.inner_product(qa[:nq], x, k=args.k, normalize=False)
        times.append(time.perf_counter() - t0)
    med = statistics.median(times)
    port = {
        "value": nq / med, "unit": "queries/s", "cores": int(threads), "kind": "port", "repeats": reps,
        "best": nq / min(times),
        "sample": f"{nq} of {args.queries} queries x full {x.shape[0]}x{x.shape[1]} index (fp32 up-cast of the stored "
                  f"values), NumPy matmul + full argsort = oracle.inner_product (mips.py:552-560), median of {reps} runs "
                  f"of {med:.2f} s; argsort is single-threaded, matmul uses {threads} BLAS threads; host has "
                  f"{os.cpu_count()} logical cores",
    }

    ncpu = os.cpu_count() or 1
    torch.set_num_threads(ncpu)
    nqt = min(args.cpu_torch_queries, args.queries)
    xt, qt = torch.from_numpy(x), torch.from_numpy(qa[:nqt])
    orc.torch_topk(qt[:8], xt[:4096], args.k)
    ttimes, tres = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        tres = orc.torch_topk(qt, xt, args.k)
        ttimes.append(time.perf_counter() - t0)
    tmed = statistics.median(ttimes)
    tor = {
        "value": nqt / tmed, "unit": "queries/s", "cores": int(torch.get_num_threads()), "kind": "port", "repeats": reps,
        "best": nqt / min(ttimes),
        "sample": f"{nqt} of {args.queries} queries x full {x.shape[0]}x{x.shape[1]} index, torch.topk(q @ d.T, {args.k}) in "
                  f"fp32 (retriever_lightning.py:304-305 idiom), 256-query chunks x <= 2^20-row blocks, running top-k "
                  f"merge, median of {reps} runs of {tmed:.2f} s, torch threads = {torch.get_num_threads()} of "
                  f"{ncpu} logical cores",
    }
    return port, tor, res, (tres[0].numpy(), tres[1].numpy())


def regime(ram, torch, rows, d, nq, k, dtype, device, iters):
    """One extra (index, Q) point measured in this same run: scan-kernel time from HIP events -> both roofline
    fractions.  Used for the 2^24 x 768 index (BASELINE config 3's index on one GPU): Q = 4096 is the MFMA-bound
    case at full size, Q = 8 the HBM-bound one -- the north star's literal 'fraction of the HBM-read roofline'."""
    ix = ram.MipsIndex(d, dtype=dtype, device=device)
    ix.reserve(rows)
    ix.add_synthetic(rows, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
    out = []
    esz = 1 if dtype != "bf16" else 2
    qsz = 2 if dtype in ("bf16", "fp8_e4m3_docs") else 1   # bytes per staged query element
    for q_n, it in zip(nq, iters):
        q = ram.synth_fill(q_n, d, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS, dtype="bf16", device=device)
        for _ in range(2):
            ix.search(q, k)
        torch.cuda.synchronize()
        ix.scan_timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(it):
            ix.search(q, k)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / it
        ix.check()
        st = ix.margin_stats()
        ms, cnt = ix.scan_timing()
        scan = ms / max(1, cnt) * 1e-3
        fl = 2.0 * q_n * rows * d
        by = rows * d * float(esz) + q_n * d * float(qsz) + q_n * k * 12.0
        peak = PEAK_FP8_TFLOPS if (esz == 1 and qsz == 1) else PEAK_BF16_TFLOPS   # (e4m3 rows x bf16 queries run on the bf16 MFMA)
        hbm_bound = (by / (PEAK_HBM_GBS * 1e9)) > (fl / (peak * 1e12))
        out.append({
            "workload": f"{rows}x{d} {dtype} index, Q={q_n}, k={k}", "queries_per_s": q_n / wall, "call_ms": wall * 1e3,
            "kernel": ix.last_kernel, "kernel_ms": scan * 1e3, "launches_timed": cnt, "flagged": st["flagged"], "unresolved": st["unresolved"],
            "bound": "hbm" if hbm_bound else "mfma",
            "mfma_achieved_tflops": fl / scan / 1e12, "mfma_frac": fl / scan / (peak * 1e12),
            "hbm_achieved_gbs": by / scan / 1e9, "hbm_frac": by / scan / (PEAK_HBM_GBS * 1e9),
        })
    del ix
    torch.cuda.empty_cache()
    return out


def regime_sharded(ram, torch, dist, rows, d, nq, k, dtype, local_rank, world, rank, steps, pipelined, coll_dev, dist_info):
    """N > 1: BASELINE config 3 (config 5 with dtype fp8_e4m3) in the same run -- a `rows` x d index row-sharded over the ranks
    (weak in nothing: the total is fixed, rank r keeps rows [r ceil(rows / N), ...)), Q queries replicated, every step = local
    fused scan + ONE all-gather + replicated merge, all inside the timed region (barrier + synchronize on both sides, MAX over
    ranks).  Reports the system rate and, per rank, shard size, scan-kernel time (HIP events) and MFMA fraction.  Collective:
    every rank calls it."""
    index = ram.ShardedMipsIndex(d, metric=ram.METRIC_IP, dtype=dtype, device=local_rank)
    index.add_synthetic_global(rows, ram.SEED_DOCS, ram.SYNTH_GAUSS)
    q = ram.synth_fill(nq, d, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS, dtype="bf16", device=local_rank)
    local_rows = index.hi - index.lo

    def run(count):
        out = None
        if pipelined:
            prev = None
            for _ in range(count):
                cur = index.search_async(q, k)
                if prev is not None:
                    out = prev.result()
                prev = cur
            out = prev.result()
        else:
            for _ in range(count):
                out = index.search(q, k)
        return out

    run(2)
    torch.cuda.synchronize()
    dist.barrier()
    index.local.scan_timing(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, i = run(steps)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    index.check()
    st = index.margin_stats()  # (collective: summed over the shards)
    ms, cnt = index.local.scan_timing()
    scan_ms = ms / max(1, cnt)
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    per = torch.zeros(world, 2, dtype=torch.float64, device=coll_dev)
    per[rank, 0], per[rank, 1] = float(local_rows), scan_ms
    dist.all_reduce(per)
    chk = torch.stack([i.sum().to(torch.float64), s.double().sum()]).to(t.device)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    esz = 1 if dtype != "bf16" else 2
    peak = PEAK_FP8_TFLOPS if esz == 1 else PEAK_BF16_TFLOPS
    per = per.cpu().tolist()
    gpus = [{"rank": r, "rows": int(per[r][0]), "scan_kernel_ms": per[r][1],
             "mfma_frac": (2.0 * nq * per[r][0] * d / (per[r][1] * 1e-3) / (peak * 1e12)) if per[r][1] > 0 else None,
             "hbm_frac": ((per[r][0] * d * esz + nq * d * esz + nq * k * 12.0) / (per[r][1] * 1e-3) / (PEAK_HBM_GBS * 1e9)) if per[r][1] > 0 else None}
            for r in range(world)]
    kernel = index.local.last_kernel
    del index
    torch.cuda.empty_cache()
    return {"workload": f"{rows}x{d} {dtype} index row-sharded x{world} (BASELINE config {'5' if esz == 1 else '3'}"
                        f"{'' if rows == 1 << 24 else ' at reduced size'}), Q={nq} replicated, k={k}, 1 all-gather + merge per step",
            "queries_per_s": nq * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "scaling": "strong",
            "pipelined_exchange": bool(pipelined), "kernel": kernel, "per_gpu": gpus, "ranks_agree": bool(torch.equal(lo, hi)),
            "margin": st, "distributed": dist_info}


def regime_certified(ram, torch, rows, d, nq, device, iters=5):
    """Two more points of the same run, same index size as the headline: paths that rest on the per-query certificate
    (device outputs, default margin mode = stream-ordered re-scan of flagged queries, nothing synchronises) --
    the fp32-exact index (the reference's own dtype) through the two-stage search, and k = 10 through optimistic pools."""
    out = []
    g = torch.Generator(device=f"cuda:{device}").manual_seed(0xD0C5)
    for dtype, k in (("f32", 5), ("bf16", 10)):
        ix = ram.MipsIndex(d, dtype=dtype, device=device)
        ix.reserve(rows)
        if dtype == "f32":  # rows that are NOT bf16-representable (the synthetic generator emits bf16 values)
            for r0 in range(0, rows, 1 << 18):
                ix.add(torch.randn(min(1 << 18, rows - r0), d, device=f"cuda:{device}", generator=g))
            q = torch.randn(nq, d, device=f"cuda:{device}", generator=g)
        else:
            ix.add_synthetic(rows, 0, ram.SEED_DOCS, ram.SYNTH_GAUSS)
            q = ram.synth_fill(nq, d, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS, dtype="bf16", device=device)
        for _ in range(2):
            ix.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            ix.search(q, k)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters
        ix.check()
        st = ix.margin_stats()
        out.append({"workload": f"{rows}x{d} {'fp32-exact' if dtype == 'f32' else 'bf16'} index, Q={nq}, k={k}, device outputs, certified on the stream",
                    "queries_per_s": nq / wall, "call_ms": wall * 1e3, "kernel": ix.last_kernel, "flagged": st["flagged"],
                    "rescanned": st["rescanned"], "unresolved": st["unresolved"]})
        del ix
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if args.launch_check:
        # the N-rank launch path by itself: process group up, one collective, report (no GPU, no search)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo" if args.backend == "gloo" or not torch.cuda.is_available() else "nccl")
            t = torch.tensor([rank + 1], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                torch.cuda.set_device(local_rank)
                t = t.cuda()
            dist.all_reduce(t)
            total = int(t.item())
        else:
            total = 1
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "world_size": dist.get_world_size() if world > 1 else 1,
                              "backend": dist.get_backend() if world > 1 else None, "rank_sum": total,
                              "self_launched": os.environ.get("MIPS_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    import retrieval_augmented_mds_amd as ram  # the timed path touches the product only; oracle/ is imported
    # in cpu_baselines() alone

    if args.backend == "gloo":
        local_rank = 0  # rehearsal: every rank shares the one visible GPU
    torch.cuda.set_device(local_rank)
    dist_info = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
        rccl = None
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        except Exception:
            pass
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": rccl,
                     "self_launched": os.environ.get("MIPS_BENCH_SELF_LAUNCHED") == "1"}

    n, d, nq, k = args.rows, args.dim, args.queries, args.k
    index = ram.ShardedMipsIndex(d, metric=ram.METRIC_IP, dtype=args.index_dtype, device=local_rank)
    f8 = args.index_dtype != "bf16"
    esz = 1 if f8 else 2
    t_build = time.perf_counter()
    index.add_synthetic_global(n, ram.SEED_DOCS, ram.SYNTH_GAUSS)
    q_dev = ram.synth_fill(nq, d, 0, ram.SEED_QUERIES, ram.SYNTH_GAUSS, dtype="bf16", device=local_rank)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    local_rows = index.hi - index.lo

    def barrier():
        if world > 1:
            dist.barrier()

    coll_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"
    rows_per_gpu = [local_rows]
    if world > 1:  # every rank's shard size, as the process group reports it
        t = torch.zeros(world, dtype=torch.int64, device=coll_dev)
        t[rank] = local_rows
        dist.all_reduce(t)
        rows_per_gpu = [int(v) for v in t.cpu().tolist()]
        if sum(rows_per_gpu) != n:
            raise SystemExit(f"shards cover {sum(rows_per_gpu)} of {n} rows")

    # N > 1 over RCCL: consecutive steps are independent query batches, so the exchange step of batch t (the ONE
    # all-gather + merge, on a side stream) overlaps the shard scan of batch t + 1 (ShardedMipsIndex.search_async).
    # Every one of the K steps still completes inside the timed region.
    # (At N = 1 the same call would put the ~45 us tail of step t beside the scan of step t + 1; measured: no gain --
    # 4.794 vs 4.786 ms per step -- because a scan workgroup owns all 512 registers of every SIMD of its CU, so the
    # tail's waves only get on the machine when a scan workgroup leaves.  --pipeline-single turns it on anyway.)
    pipelined = ((world > 1 and args.backend == "nccl") or (world == 1 and args.pipeline_single)) and not args.no_pipeline

    def run_steps(count):
        out = None
        if pipelined:  # (read at call time: the fallback below may switch it off)
            prev = None
            for _ in range(count):
                cur = index.search_async(q_dev, k)
                if prev is not None:
                    out = prev.result()
                prev = cur
            if prev is not None:
                out = prev.result()
        else:
            for _ in range(count):
                out = index.search(q_dev, k)
        return out

    try:
        run_steps(max(1, args.warmup))
        torch.cuda.synchronize()
    except Exception as e:  # the overlapped exchange is an optimisation: never let it cost the measurement
        if not pipelined:
            raise
        print(f"[bench] pipelined exchange failed on rank {rank} ({e!r}); using the synchronous path", file=sys.stderr, flush=True)
        pipelined = False
    if world > 1:  # all ranks must take the same path
        flag = torch.tensor([1 if pipelined else 0], device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if pipelined and int(flag.item()) == 0:
            pipelined = False
        if not pipelined:
            run_steps(max(1, args.warmup))
    torch.cuda.synchronize()
    barrier()
    index.local.scan_timing(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, i = run_steps(args.steps)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    index.check()  # a scan kernel that gave up on its barrier poisons its results: never print a number over that
    if bool((i == ram.IDX_POISON).any()):
        raise SystemExit("the timed searches returned poisoned rows (scan kernel timeout on some rank)")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must hold the same global answer
        chk = torch.stack([i.sum().to(torch.float64), s.double().sum()]).to(t.device)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit("ranks disagree on the merged top-k")
    scan_ms_sum, scan_launches = index.local.scan_timing()
    scan_ms = scan_ms_sum / max(1, scan_launches)
    # margin check of the last timed search, summed over the shards (read now: the run is over).  A query left `unresolved`
    # kept an uncertified first result: the headline number is then not the number of an exact search, and is not printed.
    margin = index.margin_stats(synchronize=True)
    if margin["unresolved"] > 0:
        raise SystemExit(f"the timed searches left {margin['unresolved']} of {nq} queries unresolved (flagged {margin['flagged']}): "
                         "not an exact search -- no value is reported")

    ms_per_step = elapsed / args.steps * 1e3
    value = nq * args.steps / elapsed

    # a second, untimed-for-`value` pass with one event pair per step: the distribution of step times
    # (median and min).  Un-pipelined, so each sample is one complete step.
    step_ms = []
    if world == 1 or not pipelined:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in evs:
            a.record()
            index.search(q_dev, k)
            b.record()
        torch.cuda.synchronize()
        step_ms = [a.elapsed_time(b) for a, b in evs]
    step_stats = {"median": statistics.median(step_ms), "min": min(step_ms), "max": max(step_ms), "samples": len(step_ms),
                  "how": "one HIP event pair per step, separate pass after the timed region"} if step_ms else None

    # algorithmic work of ONE scan launch on this rank: flops = 2 Q N_local d,
    # bytes = N_local d s (index, read once) + Q d s (queries) + Q k 12 (results)
    flops = 2.0 * nq * local_rows * d
    bytes_ = local_rows * d * float(esz) + nq * d * float(esz) + nq * k * 12.0
    ach_tflops = flops / (scan_ms * 1e-3) / 1e12 if scan_ms > 0 else 0.0
    ach_gbs = bytes_ / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    # HBM-side traffic of one scan launch: PMC counters cannot be read from inside this process, so the
    # figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/<round>/),
    # corrected as MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE counts wide reads at half: x2), and
    # is only reported when that profile was taken on the same workload AND the same kernel instance; otherwise null.
    kernel_name = index.local.last_kernel
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
            t = json.load(f)
        same = (t["rows_per_gpu"], t["dim"], t["queries"], t["k"], t.get("dtype", "bf16")) == (local_rows, d, nq, k, args.index_dtype)
        if same and t.get("kernel", kernel_name) == kernel_name:
            traffic = (2.0 * t["fetch_size_kb"] + t["write_size_kb"]) * 1024.0
            traffic_src = t["source"]
    except Exception:
        pass
    # what a bare MFMA loop of the kernel's own operand mapping sustains on random data on this part
    # (tools/mfma_ceiling.hip; committed measurement): the roof this kernel can actually approach
    ceiling = None
    try:
        with open(os.path.join(ROOT, "profiles", "latest_ceiling.json")) as f:
            c = json.load(f)
        if not f8:
            ceiling = c
    except Exception:
        pass
    peak = PEAK_FP8_TFLOPS if args.index_dtype == "fp8_e4m3" else PEAK_BF16_TFLOPS   # (e4m3 rows x bf16 queries run on the bf16 MFMA)
    roofline = {
        "bound": "mfma", "achieved": ach_tflops, "peak": peak, "unit": "TFLOP/s",
        "frac": ach_tflops / peak, "traffic": traffic, "traffic_source": traffic_src,
        "kernel": kernel_name, "kernel_ms": scan_ms, "launches_timed": scan_launches,
        "flops_per_launch": flops, "bytes_per_launch": bytes_,
        "hbm_achieved": ach_gbs, "hbm_peak": PEAK_HBM_GBS, "hbm_unit": "GB/s", "hbm_frac": ach_gbs / PEAK_HBM_GBS,
        "note": "Q=4096 flop per index byte >> ~310 flop/B ridge: MFMA-bound; hbm_* = literal HBM-read fraction "
                "(see `regimes` for the HBM-bound measurement)",
    }
    if ceiling is not None:
        roofline["ceiling_tflops"] = ceiling["bare_loop_tflops"]
        roofline["ceiling_frac"] = ceiling["bare_loop_tflops"] / peak
        roofline["frac_of_ceiling"] = ach_tflops / ceiling["bare_loop_tflops"]
        roofline["ceiling_source"] = ceiling["source"]

    cpu = cpu_t = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np

        cpu, cpu_t, (cs, ci), (ts, ti) = cpu_baselines(index.local, q_dev, args)
        gi, gs = i.cpu().numpy(), s.cpu().numpy()
        parity = {"indices_equal_cpu_port": bool(np.array_equal(gi[: ci.shape[0]], ci)),
                  "max_rel_score_diff": float(np.max(np.abs(gs[: cs.shape[0]] - cs) / np.abs(cs))),
                  "indices_equal_cpu_torch": bool(np.array_equal(gi[: ti.shape[0]], ti)),
                  "max_rel_score_diff_torch": float(np.max(np.abs(gs[: ts.shape[0]] - ts) / np.abs(ts)))}

    regimes = None
    if rank == 0 and world == 1 and not args.no_regimes and not f8 and (n, d) == (1 << 20, 768):
        free_b, _ = torch.cuda.mem_get_info()
        if free_b > 40e9:
            regimes = regime(ram, torch, 1 << 24, 768, [4096, 8], k, "bf16", local_rank, [5, 20])
            regimes += regime(ram, torch, 1 << 22, 1024, [4096, 8], k, "bf16", local_rank, [5, 20])   # row pitch 1024 (BASELINE config 4's rows)
            # BASELINE config 5 as it is worded (e4m3 documents, bf16 queries) where bytes bind: the literal HBM-read fraction
            regimes += regime(ram, torch, 1 << 24, 768, [8, 64], k, "fp8_e4m3_docs", local_rank, [20, 20])
            regimes += regime_certified(ram, torch, n, d, nq, local_rank)
    if world > 1 and not args.no_regimes:
        # the configuration the north star quotes scaling on (BASELINE config 3: 2^24 x 768 bf16, row-sharded) and config 5 (e4m3
        # rows) -- measured in this same N-rank run, after the headline, so that the driver's 1/2/4/8 curve carries them
        del index
        torch.cuda.empty_cache()
        rrows = args.regime_rows if args.regime_rows > 0 else ((1 << 24) if n == (1 << 20) else 4 * n)
        regimes = []
        for rdt in ("bf16", "fp8_e4m3"):
            need = -(-rrows // world) * d * (2 if rdt == "bf16" else 1) * 1.3
            ok = torch.tensor([1 if torch.cuda.mem_get_info()[0] > need else 0], device=coll_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()):
                regimes.append(regime_sharded(ram, torch, dist, rrows, d, nq, k, rdt, local_rank, world, rank, max(3, min(args.steps, 10)),
                                              pipelined, coll_dev, dist_info))

    if rank == 0:
        out = {
            "metric": f"MIPS queries/sec (exact top-{k}, {args.index_dtype} index)", "value": value, "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp8_e4m3": "fp8_e4m3", "fp8_e4m3_docs": "bf16 (e4m3 documents up-converted exactly)"}[args.index_dtype],
            "data": "synthetic",
            "config": {"workload": f"{n}x{d} {args.index_dtype} index (counter-based Gaussian, seed 0xD0C5), Q={nq} bf16 queries "
                                   f"resident in HBM, top-k={k}, exact; BASELINE config 2 at the defaults "
                                   f"(config 5 with --index-dtype fp8_e4m3)",
                       "index_rows": n, "dim": d, "queries": nq, "k": k,
                       "parallelism": (f"row-sharded x{world} + 1 all-gather" + (", tail + exchange of step t overlapped with the scan of step t+1" if pipelined else "")) if world > 1 else ("single GPU" + (", tail (select + exact re-score) of step t overlapped with the scan of step t+1" if pipelined else "")),
                       "rows_per_gpu": rows_per_gpu if world > 1 else local_rows},
            "distributed": dist_info,
            "step_ms": step_stats,
            "margin_check": {"flagged_queries_last_step": margin["flagged"], "settled_exactly": margin["rescanned"],
                             "unresolved": margin["unresolved"], "of": nq,
                             "mode": "certified on the stream: every search settles the queries it flags exactly (brute force on the canonical "
                                     "scores behind an MFMA pre-filter, one pass over the index per 16 flagged queries), without synchronising; a run that leaves a "
                                     "query unresolved prints no value",
                             "bound": "exact k-th score within d*2^-23*|q|*max|x| of the best MFMA score outside the candidate pool"},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_torch": cpu_t, "parity_vs_cpu_sample": parity,
            "regimes": regimes,
            "index_build_s": t_build,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
