#!/usr/bin/env python3
"""bench.py -- the headline SpMM measurement (BASELINE.json).

Workload: data/large_25605 (SuiteSparse n4c6-b13, 6300 x 25605, nnz 88 200) in CSR times a
synthetic dense B (25605 x K, K = 128 fp32, seeded; see mispmm/synth.py).  One "step" = one
full SpMM C = A @ B through the C ABI (mispmm_csr_f32), inputs resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N = 1 by default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N = 1: the K timed steps are captured once into a hipGraph on the bench stream and replayed
(launch-bound otherwise: one step is a few microseconds), bracketed by torch.cuda.synchronize();
`roofline.achieved` divides the algorithmic bytes by the mean per-launch time measured with HIP
events on that same stream.
N > 1: rows of A are split into N contiguous nnz-balanced ranges (one rank per GPU), B is
broadcast once from rank 0 over RCCL before the timed region, every rank multiplies its slab
each step, and the C row slabs are all-gathered with RCCL in buckets (--bucket steps per
collective) on a second stream overlapped with the following steps.  `value` is end to end
(kernels + gathers, max over ranks); `kernel_only` reports the same run's compute-stream time.
Strong scaling: the total work is fixed as N grows.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--matrix", default="n4c6-b13")
    p.add_argument("--k-cols", type=int, default=128, help="columns of the dense operand (BASELINE 'K')")
    p.add_argument("--kernel", type=int, default=0, help="CSR kernel id (0 = library default)")
    p.add_argument("--acc", default="reference", choices=["reference", "fast"])
    p.add_argument("--launch", default="graph", choices=["graph", "eager"])
    p.add_argument("--bucket", type=int, default=16, help="N>1: steps per C-slab all-gather")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline time budget")
    return p.parse_args()


def cpu_baseline(csr, b, budget_s, gpu_result, acc):
    """The CPU leg: the oracle (a port of the reference's sequential spmmCSRCpu) timed on this
    host with 1 thread -- and, since its output is at hand, used as the checker of the GPU result
    that was just timed (bit-exact in REFERENCE mode, 1e-5 of sum|a||b| in FAST mode)."""
    from oracle import oracle as orc
    orc.build()
    ref = orc.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)         # warm + checker
    if acc == "reference":
        parity = "bit-exact" if np.array_equal(gpu_result, ref) else "MISMATCH"
    else:
        scale = orc.spmm_csr(csr.row_ptrs, csr.col_idxs, np.abs(csr.data), np.abs(b)).astype(np.float64)
        ok = np.all(np.abs(gpu_result.astype(np.float64) - ref) <= 1e-5 * scale + 1e-37)
        parity = "within 1e-5" if ok else "MISMATCH"
    if parity == "MISMATCH":
        raise SystemExit("bench: GPU result does not match the oracle -- refusing to report a number")
    times, t_end = [], time.perf_counter() + budget_s
    while time.perf_counter() < t_end and len(times) < 5000:
        t0 = time.perf_counter()
        orc.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        times.append(time.perf_counter() - t0)
    best = min(times)
    # the same engine with its row loop split over the host cores this process may use (the reference is
    # single-threaded; this is the "host cores" column of SURVEY.md section 8(d)), a few seconds of it
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    mt = orc.spmm_csr_mt(csr.row_ptrs, csr.col_idxs, csr.data, b, threads)
    if not np.array_equal(mt, ref):
        raise SystemExit("bench: threaded CPU engine differs from the sequential one")
    mt_times, t_end = [], time.perf_counter() + min(3.0, budget_s)
    while time.perf_counter() < t_end and len(mt_times) < 5000:
        t0 = time.perf_counter()
        orc.spmm_csr_mt(csr.row_ptrs, csr.col_idxs, csr.data, b, threads)
        mt_times.append(time.perf_counter() - t0)
    mt_best = min(mt_times)
    return {"value": round(2.0 * csr.nnz * b.shape[1] / best / 1e9, 3), "unit": "GFLOP/s", "cores": 1,
            "all_cores": {"value": round(2.0 * csr.nnz * b.shape[1] / mt_best / 1e9, 3), "unit": "GFLOP/s",
                          "cores": threads, "ms_per_step": round(mt_best * 1e3, 4),
                          "note": "same engine, row loop split with OpenMP, bit-identical result"},
            "kind": "port", "ms_per_step": round(best * 1e3, 4), "gpu_parity": parity,
            "sample": f"the full workload ({csr.num_rows}x{csr.num_cols} nnz {csr.nnz} x K={b.shape[1]}), "
                      f"best of {len(times)} runs in {budget_s:.0f} s, oracle/spmm_oracle.c -O2, "
                      f"host has {os.cpu_count()} logical cores"}


def event_pair(l):
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    from mispmm import capi
    capi.check(l.mispmm_event_create(ctypes.byref(a)))
    capi.check(l.mispmm_event_create(ctypes.byref(b)))
    return a, b


def run_single(args):
    import torch
    from mispmm import capi, datasets, ops, synth
    l = capi.lib()
    torch.cuda.set_device(0)
    csr = datasets.load_csr(args.matrix)
    n = args.k_cols
    b_host = synth.dense_b(csr.num_cols, n)
    a = ops.DeviceCSR.from_host(csr)
    b = torch.from_numpy(b_host).cuda()
    c = torch.empty((csr.num_rows, n), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    sp = ctypes.c_void_p(stream.cuda_stream)

    def step():
        ops.spmm_csr(a, b, out=c, kernel=args.kernel, acc=args.acc, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    graphs = []
    if args.launch == "graph":
        chunk = min(args.steps, 1000)
        plan = [chunk] * (args.steps // chunk) + ([args.steps % chunk] if args.steps % chunk else [])
        cache = {}
        for size in plan:
            if size not in cache:
                capi.check(l.mispmm_graph_begin(sp))
                for _ in range(size):
                    step()
                g = ctypes.c_void_p()
                capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
                cache[size] = g
            graphs.append(cache[size])
        for g in set(g.value for g in graphs):   # one untimed replay per graph (upload)
            capi.check(l.mispmm_graph_launch(ctypes.c_void_p(g), sp))
        torch.cuda.synchronize()

    ev0, ev1 = event_pair(l)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.check(l.mispmm_event_record(ev0, sp))
    if graphs:
        for g in graphs:
            capi.check(l.mispmm_graph_launch(g, sp))
    else:
        for _ in range(args.steps):
            step()
    capi.check(l.mispmm_event_record(ev1, sp))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = ctypes.c_float()
    capi.check(l.mispmm_event_elapsed_ms(ev0, ev1, ctypes.byref(ms)))

    # the other accumulate mode, same launches, for the record (never `value`)
    other = "fast" if args.acc == "reference" else "reference"
    n_other = min(args.steps, 500)
    capi.check(l.mispmm_graph_begin(sp))
    for _ in range(n_other):
        ops.spmm_csr(a, b, out=c, kernel=args.kernel, acc=other, stream=stream)
    g_other = ctypes.c_void_p()
    capi.check(l.mispmm_graph_end(sp, ctypes.byref(g_other)))
    capi.check(l.mispmm_graph_launch(g_other, sp))
    torch.cuda.synchronize()
    capi.check(l.mispmm_event_record(ev0, sp))
    capi.check(l.mispmm_graph_launch(g_other, sp))
    capi.check(l.mispmm_event_record(ev1, sp))
    torch.cuda.synchronize()
    ms_other = ctypes.c_float()
    capi.check(l.mispmm_event_elapsed_ms(ev0, ev1, ctypes.byref(ms_other)))
    other_us = ms_other.value * 1e3 / n_other
    step()                      # leave the timed mode's result in C for the parity check
    torch.cuda.synchronize()

    # cold single shot (SURVEY.md 8(d) asks for it next to the steady-state figure): 1 GiB is written first so that
    # neither the L2s nor the 256 MiB Infinity Cache hold A, B or C; median of 5; HIP events around ONE eager launch
    # (an empty event pair costs a few microseconds itself, reported beside it)
    cold, empty = [], []
    flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(5):
            flush.fill_(1.0)
            capi.check(l.mispmm_event_record(ev0, sp))
            step()
            capi.check(l.mispmm_event_record(ev1, sp))
            stream.synchronize()
            capi.check(l.mispmm_event_elapsed_ms(ev0, ev1, ctypes.byref(ms_other)))
            cold.append(ms_other.value * 1e3)
            capi.check(l.mispmm_event_record(ev0, sp))
            capi.check(l.mispmm_event_record(ev1, sp))
            stream.synchronize()
            capi.check(l.mispmm_event_elapsed_ms(ev0, ev1, ctypes.byref(ms_other)))
            empty.append(ms_other.value * 1e3)
    del flush
    cold_us, empty_us = sorted(cold)[2], sorted(empty)[2]

    flops = datasets.spmm_flops(csr.nnz, n)
    abytes = datasets.csr_algorithmic_bytes(csr, n)
    traffic = None     # PMC counters cannot be read from inside this process: committed rocprofv3 figure
    try:
        with open(os.path.join(ROOT, "profiles", "r1", "traffic.json")) as f:
            entry = json.load(f).get(f"{args.matrix}/{n}")
        if entry and args.kernel in (0, 5):
            traffic = entry["total_bytes"]
    except OSError:
        pass
    launch_s = ms.value * 1e-3 / args.steps
    achieved = abytes / launch_s / 1e9
    info = capi.device_info(0)
    out = {
        "metric": "SpMM GFLOP/s, large_25605 (n4c6-b13) CSR x dense K=%d fp32" % n,
        "value": round(flops * args.steps / wall / 1e9, 2), "unit": "GFLOP/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 6),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "SuiteSparse n4c6-b13 (reference data/large_25605) x seeded synthetic B",
        "config": {"workload": f"{args.matrix} CSR {csr.num_rows}x{csr.num_cols} nnz {csr.nnz} x dense K={n} fp32",
                   "kernel": args.kernel, "acc_mode": args.acc, "launch": args.launch, "device": info["name"],
                   "uniform_row_hint": a.uniform_row_nnz if args.kernel in (0, 5) else 0},
        "achieved_hbm_GBps": round(abytes * args.steps / wall / 1e9, 1),
        "other_acc_mode": {"acc_mode": other, "launch_us": round(other_us, 3),
                           "roofline_frac": round(abytes / (other_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                           "note": "reference = the reference engine's fp64 accumulate (bit-exact); fast = fp32 fma chain (<= 1e-5)"},
        "cold_single_shot": {"launch_us": round(cold_us, 2), "empty_event_pair_us": round(empty_us, 2),
                             "note": "one eager launch after a 1 GiB cache flush, HIP events, median of 5 (not the metric)"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": abytes, "launch_us": round(launch_s * 1e6, 3),
                     "note": "launch_us = HIP-event time over the timed region / steps (includes inter-kernel gaps); "
                             "traffic = L2<->fabric bytes per launch from rocprofv3 PMC (profiles/r1/traffic.json)"},
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(csr, b_host, args.cpu_seconds, c.cpu().numpy(), args.acc)
    print(json.dumps(out))


def run_multi(args):
    import torch
    import torch.distributed as dist
    from mispmm import capi, datasets, dist as mdist, synth
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    capi.lib()
    csr = datasets.load_csr(args.matrix)
    n = args.k_cols
    job = mdist.ShardedCsrSpmm(csr, n, device=torch.device("cuda", local), kernel=args.kernel, acc=args.acc,
                               bucket=args.bucket)
    b_host = synth.dense_b(csr.num_cols, n) if rank == 0 else None
    job.broadcast_b(b_host)                      # one-time, outside the timed region
    job.run(args.warmup)
    job.finish()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.run(args.steps)
    job.finish()
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t0
    # the same steps with C left row-sharded (no collective): the kernel-only figure
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    job.run(args.steps, gather=False)
    job.finish(gather=False)
    torch.cuda.synchronize()
    dist.barrier()
    compute_s = time.perf_counter() - t1
    t = torch.tensor([wall, compute_s], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, compute_s = float(t[0]), float(t[1])
    # sharded + gathered C must equal the unsharded single-GPU product bit for bit (row independence)
    ok = True
    if rank == 0:
        from mispmm import ops
        whole = ops.spmm_csr(ops.DeviceCSR.from_host(csr, device=job.device), job.b, kernel=args.kernel, acc=args.acc)
        torch.cuda.synchronize()
        ok = bool(torch.equal(whole, job.gathered_c()))
    if rank == 0:
        if not ok:
            raise SystemExit("bench: gathered C differs from the unsharded product -- refusing to report a number")
        flops = datasets.spmm_flops(csr.nnz, n)
        abytes = datasets.csr_algorithmic_bytes(csr, n)
        out = {
            "metric": "SpMM GFLOP/s, large_25605 (n4c6-b13) CSR x dense K=%d fp32" % n,
            "value": round(flops * args.steps / wall / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 6),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "SuiteSparse n4c6-b13 (reference data/large_25605) x seeded synthetic B",
            "config": {"workload": f"{args.matrix} CSR {csr.num_rows}x{csr.num_cols} nnz {csr.nnz} x dense K={n} fp32",
                       "parallelism": f"row-sharded x{world}, B replicated, C slabs all-gathered (RCCL) every "
                                      f"{args.bucket} steps", "kernel": args.kernel, "acc_mode": args.acc,
                       "check": "gathered C == unsharded single-GPU C (bitwise)"},
            "achieved_hbm_GBps": round(abytes * args.steps / wall / 1e9, 1),
            "kernel_only": {"value": round(flops * args.steps / compute_s / 1e9, 2), "unit": "GFLOP/s",
                            "ms_per_step": round(compute_s * 1e3 / args.steps, 6),
                            "note": "the same steps re-run with C left row-sharded (no collective), max over ranks"},
            "roofline": {"bound": "hbm", "achieved": round(abytes * args.steps / compute_s / 1e9, 1),
                         "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": round(abytes * args.steps / compute_s / 1e9 / (HBM_PEAK_GBS * world), 4),
                         "traffic": None},
        }
        print(json.dumps(out))
    dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("MISPMM_FORCE_DIST") == "1"    # rehearse the RCCL path with a single rank
    if args.gpus > 1 or world > 1 or force:
        if world == 1 and not force:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        run_multi(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
