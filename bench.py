#!/usr/bin/env python3
"""bench.py -- the SpMM measurement of BASELINE.json, one JSON line per run.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config headline|2|3|4|5]

One "step" = one full SpMM C = A @ B through the C ABI of libmispmm.so with A, B and C resident in HBM.

  --config headline  large_25605 (n4c6-b13) CSR x dense K=128 fp32        (the BASELINE.json metric; default)
           2         medium_4096 (stand-in delaunay_n12) CSR x K=128 fp32
           3         large_25605 ELL x K=256 fp32
           4         large_20000 (ACTIVSg10K) BSR block 16 x K=128 bf16 (MFMA)
           5         large_25605 CSR x K=512 fp32  (row-sharded when --gpus N > 1)

N = 1.  W eager warm-up steps, then the K timed steps are captured once into a hipGraph on the bench stream.
A step lasts a few microseconds, so a single replay of a small K would time the launch latency of the graph and
the clock ramp instead of the kernel: the graph is therefore replayed untimed for >= 50 ms (precondition) and
then R times back to back inside one HIP-event pair, R chosen so that the timed region lasts >= 20 ms; that is
repeated for 5 rounds.  A graph of fewer than 1000 launches holds the K steps several times over (two consecutive
hipGraphLaunch calls leave the GPU idle for ~7 us: graph-API machinery, reported as `graph_of_exactly_steps_us`).  `ms_per_step`, `value` and `roofline` come from the MEDIAN round's event time divided by
K * R launches; min and the spread are reported beside it; `replays` = R.  K and W are used exactly as passed.
Where the gathered operand B sits in memory moves the time of one binary on one box by up to 6 % (profiles/r3/
placement_probe.log: fresh copies of B and C in one process), so the whole measurement above is made on `--placements`
(default 5) freshly allocated copies of B and C and the line reports the MEDIAN copy; `timing.placements_us` lists them all.

`hbm_streaming` (N = 1, every configuration): the steady-state loop above multiplies the same B into the same C, so its
working set (17-66 MB) never leaves the 256 MiB Infinity Cache -- `value` is that figure, as BASELINE.json defines the metric.
Beside it the SAME kernel is timed over S distinct (B, C) pairs in rotation, S = ceil(512 MiB / (bytes of B + C)) (headline:
32 pairs, 522 MB), captured into one hipGraph whose launch count is a multiple of S, so that every launch reads a B and
writes a C that 2 x 256 MiB of other operands have passed the caches since: B streams from HBM, C streams to it, A (0.7 MB,
the same matrix every step) stays cached.  It is also the honest form of the reference's timing contract -- one cold,
un-warmed launch per kernel (reference/src/engine/engine.cpp:41-44) -- without the event-pair overhead of a single shot.

`roofline.traffic` is measured in the same run: before this process touches the GPU it runs the configuration twice as a
child under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE: separate passes, eager launches); profiles/r3/traffic.json is the
fallback (--no-extras / --no-live-traffic skip the passes).

N > 1 (--config headline | 3 | 4 | 5: CSR by rows, ELL by rows, bf16 BSR-16 by block rows).  Started plainly
(`python bench.py --gpus N`), this process spawns its N ranks itself (children, before
it makes any GPU call); under torch.distributed.run it is one of the ranks.  Rows of A are cut into N contiguous
nnz-balanced ranges, B is broadcast once (outside the timed region), every rank multiplies its slab each step
and the C row slabs are exchanged in buckets: `allgather` (RCCL all_gather_into_tensor on a second stream) or
`peer` (the kernel's slab is copied straight into every peer's C over xGMI through IPC-mapped buffers).  `value`
is end to end (kernels + exchange, max over ranks); `kernel_only` re-runs the steps with C left row-sharded.
Strong scaling: the total work is fixed as N grows.

The CPU leg (`cpu_baseline`) is the oracle under oracle/ -- a port of the reference's sequential CPU engine --
timed on this host with 1 thread, and also the checker of the GPU result that was just timed: a mismatch
refuses to print a number.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, same guide
PRECONDITION_S = 0.05            # untimed replays before the timed region
TIMED_S = 0.02                   # minimum length of one timed round
ROUNDS = 5
WATCHDOG_EXIT = 3                # exit code of every rank when an exchange mode hung and the watchdog printed the partial line
MIN_GRAPH_NODES = 1000           # launches per captured graph (the K steps are captured ceil(1000 / K) times over)
STREAM_BYTES = 512 << 20        # (B, C) pairs in rotation for the HBM-streamed figure: twice the 256 MiB Infinity Cache
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r4", "traffic.json")
_REAL_STDOUT = None


def emit(obj):
    """The one JSON line of the contract, on the process's original stdout."""
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--config", default="headline", choices=["headline", "2", "3", "4", "5"])
    p.add_argument("--matrix", default=None, help="override the configuration's matrix (packed name under data/)")
    p.add_argument("--k-cols", type=int, default=None, help="override the columns of the dense operand (BASELINE 'K')")
    p.add_argument("--kernel", type=int, default=0, help="kernel id of the format's entry point (0 = library default; CSR 7 = the LDS-tile "
                                                         "kernel mispmm_csr_lds_tile_f32, an experiment)")
    p.add_argument("--acc", default="reference", choices=["reference", "fast"])
    p.add_argument("--launch", default="graph", choices=["graph", "eager"])
    p.add_argument("--bucket", type=int, default=0, help="N>1: steps per C-slab exchange and per bucket hipGraph (0 = 64)")
    p.add_argument("--exchange", default="allgather", choices=["allgather", "peer", "both"],
                   help="N>1: how C slabs travel: `allgather` (default) = RCCL all_gather_into_tensor, `peer` = direct stores into "
                        "IPC-mapped peer buffers (has only ever run between processes on one card), `both` measures the two and "
                        "reports the faster as `value`")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--mode-timeout", type=int, default=240,
                   help="N>1: seconds an exchange mode after the first measured one may take before the line is printed without it (0 = wait)")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the other accumulate mode, the batched launch and (unless --hbm-streaming on) the HBM-streamed figure")
    p.add_argument("--hbm-streaming", default="auto", choices=["auto", "on", "off"],
                   help="N=1: time the kernel over distinct (B, C) pairs in rotation as well (`hbm_streaming`); auto = unless --no-extras")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline time budget")
    p.add_argument("--no-live-traffic", action="store_true",
                   help="N=1: skip the live PMC measurement.  By default (and unless --no-extras) this configuration is first run "
                        "twice as a child under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: separate passes, eager launches) "
                        "before anything here touches the GPU, and roofline.traffic is the L2<->fabric bytes per launch measured "
                        "there; on any failure the committed figure of profiles/r3/traffic.json is printed instead (adds ~40 s)")
    p.add_argument("--operand-sets", type=int, default=1,
                   help="N=1: > 1 makes the MAIN timed loop rotate over this many distinct (B, C) pairs (0 = as many as 512 MiB "
                        "takes): the HBM-streamed form of the workload.  The default line keeps the resident loop as `value` and "
                        "reports the streamed one as `hbm_streaming`; this flag is for profiler children and experiments")
    p.add_argument("--batch", type=int, default=8, help="N>1: dense operands per launch of the `kernel_only_batched` leg (0 = skip it)")
    p.add_argument("--placements", type=int, default=5,
                   help="N=1: time the K steps on this many freshly allocated copies of B and C and report the MEDIAN copy "
                        "(every copy is listed under timing.placements); 1 = the operands as first allocated")
    return p.parse_args()


# ------------------------------------------------------------------------------------------ workloads
class Workload:
    """One BASELINE configuration on one GPU: operands resident on the device, `step()` enqueues one SpMM."""
    fmt = "csr"
    dtype = "f32"

    def step(self, stream, acc=None):
        raise NotImplementedError

    def host_result(self):
        raise NotImplementedError

    def fresh_operands(self):
        """New device copies of the dense operand and of C (the old ones stay alive, so the new ones sit elsewhere):
        the time of these gather kernels moves by up to 6 % with WHERE B sits (profiles/r3/placement_probe.log)."""
        import torch
        self._kept = getattr(self, "_kept", []) + [self.b, self.c]
        self.b = torch.from_numpy(self.b_host).cuda()
        self.c = torch.empty_like(self.c)

    # -- the HBM-streamed form: distinct (dense operand, C) pairs in rotation --------------------------------------
    def dense_operand(self):
        return self.b

    def pair_bytes(self):
        d = self.dense_operand()
        return d.numel() * d.element_size() + self.c.numel() * self.c.element_size()

    def make_pairs(self, n):
        """n (dense operand, C) pairs in buffers of their own (pair 0 = the operands of the resident loop); same values in
        every copy -- what differs is the ADDRESS, which is all the caches see."""
        import torch
        d = self.dense_operand()
        return [(d, self.c)] + [(d.clone(), torch.empty_like(self.c)) for _ in range(n - 1)]


class CsrWorkload(Workload):
    fmt = "csr"

    def __init__(self, args, matrix, n, label):
        import torch
        from mispmm import datasets, ops, synth
        self.args, self.n, self.label, self.matrix = args, n, label, matrix
        self.csr = datasets.load_csr(matrix)
        self.b_host = synth.dense_b(self.csr.num_cols, n)
        self.a = ops.DeviceCSR.from_host(self.csr)
        self.b = torch.from_numpy(self.b_host).cuda()
        self.c = torch.empty((self.csr.num_rows, n), dtype=torch.float32, device="cuda")
        self.flops = datasets.spmm_flops(self.csr.nnz, n)
        self.abytes = datasets.csr_algorithmic_bytes(self.csr, n)
        self.workload = (f"{matrix} CSR {self.csr.num_rows}x{self.csr.num_cols} nnz {self.csr.nnz} x dense K={n} fp32")
        self.extra_config = {"uniform_row_hint": self.a.uniform_row_nnz if args.kernel in (0, 5) else 0}
        self.has_fast = True
        # --kernel 7: the LDS-tile kernel (mispmm_csr_lds_tile_f32; an experiment of round 4)
        self.tiles = ops.DeviceCSRTiles.from_host(self.csr) if args.kernel == 7 else None
        if self.tiles is not None:
            self.extra_config.update({"lds_tiles": self.tiles.num_tiles, "b_slices_staged_over_nnz": round(self.tiles.num_listed / self.csr.nnz, 4)})

    def step(self, stream, acc=None, pair=None):
        from mispmm import ops
        b, c = pair or (self.b, self.c)
        if self.tiles is not None:
            ops.spmm_csr_tiles(self.tiles, b, out=c, acc=acc or self.args.acc, stream=stream)
            return
        ops.spmm_csr(self.a, b, out=c, kernel=self.args.kernel, acc=acc or self.args.acc, stream=stream)

    def host_result(self):
        return self.c.cpu().numpy()

    def oracle_call(self, orc):
        c = self.csr
        return lambda: orc.spmm_csr(c.row_ptrs, c.col_idxs, c.data, self.b_host)

    def oracle_threads_call(self, orc, threads):
        c = self.csr
        return lambda: orc.spmm_csr_mt(c.row_ptrs, c.col_idxs, c.data, self.b_host, threads)

    def check(self, orc, got, ref, acc):
        if acc == "reference":
            return "bit-exact" if np.array_equal(got, ref) else "MISMATCH"
        c = self.csr
        scale = orc.spmm_csr(c.row_ptrs, c.col_idxs, np.abs(c.data), np.abs(self.b_host)).astype(np.float64)
        return "within 1e-5" if np.all(np.abs(got.astype(np.float64) - ref) <= 1e-5 * scale + 1e-37) else "MISMATCH"


class EllWorkload(Workload):
    fmt = "ell"

    def __init__(self, args, matrix, n, label):
        import torch
        from mispmm import datasets, formats, ops, synth
        self.args, self.n, self.label, self.matrix = args, n, label, matrix
        self.csr = datasets.load_csr(matrix)
        self.ellc = formats.csr_to_ell_colmajor(self.csr)            # the reference's on-disk / in-class layout
        self.b_host = synth.dense_b(self.csr.num_cols, n)
        self.a = ops.DeviceELL.from_host(self.ellc)
        self.b = torch.from_numpy(self.b_host).cuda()
        self.c = torch.empty((self.csr.num_rows, n), dtype=torch.float32, device="cuda")
        self.flops = datasets.spmm_flops(self.csr.nnz, n)
        self.abytes = datasets.ell_algorithmic_bytes(self.csr.num_rows, self.a.width, self.csr.num_cols, n)
        self.workload = (f"{matrix} ELL {self.csr.num_rows}x{self.csr.num_cols} width {self.a.width} "
                         f"(nnz {self.csr.nnz}) x dense K={n} fp32")
        self.extra_config = {"ell_width": self.a.width}
        self.has_fast = True

    def step(self, stream, acc=None, pair=None):
        from mispmm import ops
        b, c = pair or (self.b, self.c)
        ops.spmm_ell(self.a, b, out=c, kernel=self.args.kernel, acc=acc or self.args.acc, stream=stream)

    def host_result(self):
        return self.c.cpu().numpy()

    def oracle_call(self, orc):
        e = self.ellc
        return lambda: orc.spmm_ell_colmajor(e.num_rows, e.row_idxs, e.data, self.b_host)

    def oracle_threads_call(self, orc, threads):
        return None

    def check(self, orc, got, ref, acc):
        if acc == "reference":
            return "bit-exact" if np.array_equal(got, ref) else "MISMATCH"
        c = self.csr
        scale = orc.spmm_csr(c.row_ptrs, c.col_idxs, np.abs(c.data), np.abs(self.b_host)).astype(np.float64)
        return "within 1e-5" if np.all(np.abs(got.astype(np.float64) - ref) <= 1e-5 * scale + 1e-37) else "MISMATCH"


class BsrBf16Workload(Workload):
    """BASELINE config 4.  The reference has no bf16: A and B are rounded to bf16 (RNE) before BOTH the oracle
    and the kernel; fp32 accumulate on v_mfma_f32_16x16x32_bf16; C fp32."""
    fmt = "bsr"
    dtype = "bf16"

    def __init__(self, args, matrix, n, label, block=16):
        import torch
        from mispmm import datasets, formats, ops, synth
        self.args, self.n, self.label, self.matrix, self.block = args, n, label, matrix, block
        self.csr = datasets.load_csr(matrix)
        self.bsr = formats.csr_to_bsr(self.csr, block)
        self.b_host = synth.dense_b(self.csr.num_cols, n)
        self.a = ops.DeviceBSR.from_host(self.bsr)
        # once-per-upload analysis: occupied columns per block row, (a) in 4 fixed step slots per block row for the
        # workgroup-per-block-row kernel (default), (b) as a plain step list for the wave-per-block-row kernel (--kernel 2)
        self.slots = ops.DeviceBSRCSlots.from_host(self.bsr)
        self.bsrc = ops.DeviceBSRC.from_host(self.bsr)
        self.blocks16 = ops.f32_to_bf16(self.a.data)
        self.b16 = ops.f32_to_bf16(torch.from_numpy(self.b_host).cuda())
        self.c = torch.empty((self.csr.num_rows, n), dtype=torch.float32, device="cuda")
        self.a16_host = synth.bf16_round(self.bsr.data.reshape(-1)).reshape(self.bsr.data.shape)
        self.b16_host = synth.bf16_round(self.b_host.reshape(-1)).reshape(self.b_host.shape)
        self.flops = datasets.spmm_flops(self.csr.nnz, n)                       # useful flops (non-zeros of A)
        self.executed_flops = 2.0 * self.bsr.num_blocks * block * block * n       # dense block products
        # algorithmic bytes = what the kernel that runs must move: ITS operand + B (bf16) + C (fp32).  The BSR-16 operand
        # (dense 16 x 16 bf16 blocks) is what only the dense-block kernel reads; it is kept as a labelled second figure.
        dense_b_c = self.csr.num_cols * n * 2 + self.csr.num_rows * n * 4
        self.bsr16_bytes = datasets.bsr_algorithmic_bytes(self.bsr, n, elem=2, out_elem=4)
        self.kernel_bytes = {"slots": self.slots.operand_bytes() + dense_b_c,
                             "steps": self.bsrc.num_steps * (1024 + 128) + (self.bsr.num_block_rows + 1) * 4 + dense_b_c,
                             "dense": self.bsr16_bytes}
        self.which = {0: "slots", 1: "dense", 2: "steps"}.get(args.kernel, "slots")
        self.abytes = self.kernel_bytes[self.which]
        self.workload = (f"{matrix} BSR block {block} {self.csr.num_rows}x{self.csr.num_cols} "
                         f"{self.bsr.num_blocks} blocks (nnz {self.csr.nnz}) x dense K={n} bf16, C fp32")
        self.kernel_names = {"slots": "column-compacted block rows, workgroup per block row, 4 step slots (mispmm_bsrc_slots_bf16)",
                             "steps": "column-compacted block rows, wave per block row (mispmm_bsrc_bf16)",
                             "dense": "one B panel per block (mispmm_bsr_bf16)"}
        self.extra_config = {"block_dim": block, "blocks": int(self.bsr.num_blocks), "mfma_k_steps": int(self.bsrc.num_steps),
                             "bsr_kernel": self.kernel_names[self.which]}
        self.has_fast = False

    def fresh_operands(self):
        import torch
        from mispmm import ops
        self._kept = getattr(self, "_kept", []) + [self.b16, self.c]
        self.b16 = ops.f32_to_bf16(torch.from_numpy(self.b_host).cuda())
        self.c = torch.empty_like(self.c)

    def dense_operand(self):
        return self.b16

    def step(self, stream, acc=None, which=None, pair=None):
        from mispmm import ops
        which = which or self.which
        b16, c = pair or (self.b16, self.c)
        if which == "dense":
            ops.spmm_bsr_bf16(self.a, self.blocks16, b16, out_bf16=False, out=c, stream=stream)
        elif which == "steps":
            ops.spmm_bsrc_bf16(self.bsrc, b16, out_bf16=False, out=c, stream=stream)
        else:
            ops.spmm_bsrc_slots_bf16(self.slots, b16, out_bf16=False, out=c, stream=stream)

    def host_result(self):
        return self.c.cpu().numpy()

    def oracle_call(self, orc):
        s = self.bsr
        return lambda: orc.spmm_bsr(s.num_rows, self.block, self.block, s.block_row_ptrs, s.block_col_idxs,
                                    self.a16_host, self.b16_host)

    def oracle_threads_call(self, orc, threads):
        return None

    def check(self, orc, got, ref, acc):
        from mispmm import synth
        c = self.csr
        scale = orc.spmm_csr(c.row_ptrs, c.col_idxs, np.abs(synth.bf16_round(c.data)),
                             np.abs(self.b16_host)).astype(np.float64)
        ok = np.all(np.abs(got.astype(np.float64) - ref) <= 2e-6 * scale + 1e-30)
        return "within 2e-6 of sum|a||b| (bf16-rounded inputs, fp32 accumulate)" if ok else "MISMATCH"


def make_workload(args):
    cfg = args.config
    if cfg == "headline":
        m, n = args.matrix or "n4c6-b13", args.k_cols or 128
        return CsrWorkload(args, m, n, "large_25605" if m == "n4c6-b13" else m)
    if cfg == "2":
        m, n = args.matrix or "delaunay_n12", args.k_cols or 128
        return CsrWorkload(args, m, n, "medium_4096 (stand-in delaunay_n12)" if m == "delaunay_n12" else m)
    if cfg == "3":
        m, n = args.matrix or "n4c6-b13", args.k_cols or 256
        return EllWorkload(args, m, n, "large_25605" if m == "n4c6-b13" else m)
    if cfg == "4":
        m, n = args.matrix or "ACTIVSg10K", args.k_cols or 128
        return BsrBf16Workload(args, m, n, "large_20000" if m == "ACTIVSg10K" else m)
    m, n = args.matrix or "n4c6-b13", args.k_cols or 512
    return CsrWorkload(args, m, n, "large_25605" if m == "n4c6-b13" else m)


def metric_label(w):
    kind = {"csr": "CSR", "ell": "ELL", "bsr": f"BSR block {getattr(w, 'block', 0)}"}[w.fmt]
    return f"SpMM GFLOP/s, {w.label} ({w.matrix}) {kind} x dense K={w.n} {'bf16' if w.dtype == 'bf16' else 'fp32'}"


# ------------------------------------------------------------------------------------------ CPU leg
def cpu_baseline(w, budget_s, gpu_result, acc):
    """The oracle (a port of the reference's sequential CPU engine for this format) timed on this host with 1
    thread on the full workload -- and, since its output is at hand, the checker of the GPU result."""
    from oracle import oracle as orc
    orc.build()
    call = w.oracle_call(orc)
    t0 = time.perf_counter()
    ref = call()                                                        # warm + checker
    first = time.perf_counter() - t0
    parity = w.check(orc, gpu_result, ref, acc)
    if parity == "MISMATCH":
        raise SystemExit("bench: GPU result does not match the oracle -- refusing to report a number")
    times, t_end = [first], time.perf_counter() + max(0.0, budget_s - first)
    while time.perf_counter() < t_end and len(times) < 5000:
        t0 = time.perf_counter()
        call()
        times.append(time.perf_counter() - t0)
    best = min(times)
    out = {"value": round(w.flops / best / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
           "ms_per_step": round(best * 1e3, 4), "gpu_parity": parity,
           "sample": f"the full workload ({w.workload}), best of {len(times)} runs in {budget_s:.0f} s, "
                     f"oracle/spmm_oracle.c -O2, host has {os.cpu_count()} logical cores"}
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    mt_call = w.oracle_threads_call(orc, threads)
    if mt_call is not None:
        # the same engine with its row loop split over the host cores (the reference is single-threaded;
        # this is the "host cores" column of SURVEY.md section 8(d))
        if not np.array_equal(mt_call(), ref):
            raise SystemExit("bench: threaded CPU engine differs from the sequential one")
        mt_times, t_end = [], time.perf_counter() + min(3.0, budget_s)
        while time.perf_counter() < t_end and len(mt_times) < 5000:
            t0 = time.perf_counter()
            mt_call()
            mt_times.append(time.perf_counter() - t0)
        mt_best = min(mt_times)
        out["all_cores"] = {"value": round(w.flops / mt_best / 1e9, 3), "unit": "GFLOP/s", "cores": threads,
                            "ms_per_step": round(mt_best * 1e3, 4),
                            "note": "same engine, row loop split with OpenMP, bit-identical result"}
    return out


# ------------------------------------------------------------------------------------------ timing
class Timer:
    def __init__(self, stream):
        from mispmm import capi
        self.capi, self.l = capi, capi.lib()
        self.stream = stream
        self.sp = ctypes.c_void_p(stream.cuda_stream)
        self.ev0, self.ev1 = ctypes.c_void_p(), ctypes.c_void_p()
        capi.check(self.l.mispmm_event_create(ctypes.byref(self.ev0)))
        capi.check(self.l.mispmm_event_create(ctypes.byref(self.ev1)))

    def capture(self, fn, steps):
        """`steps` calls of fn captured into hipGraphs of at most 1000 kernel nodes; returns the launch list."""
        chunk = min(steps, 1000)
        plan = [chunk] * (steps // chunk) + ([steps % chunk] if steps % chunk else [])
        cache, graphs = {}, []
        for size in plan:
            if size not in cache:
                self.capi.check(self.l.mispmm_graph_begin(self.sp))
                for _ in range(size):
                    fn()
                g = ctypes.c_void_p()
                self.capi.check(self.l.mispmm_graph_end(self.sp, ctypes.byref(g)))
                cache[size] = g
            graphs.append(cache[size])
        return graphs

    def _pass(self, graphs, fn, steps):
        if graphs:
            for g in graphs:
                self.capi.check(self.l.mispmm_graph_launch(g, self.sp))
        else:
            for _ in range(steps):
                fn()

    def event_ms(self, body):
        import torch
        ms = ctypes.c_float()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.capi.check(self.l.mispmm_event_record(self.ev0, self.sp))
        body()
        self.capi.check(self.l.mispmm_event_record(self.ev1, self.sp))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        self.capi.check(self.l.mispmm_event_elapsed_ms(self.ev0, self.ev1, ctypes.byref(ms)))
        return ms.value, wall * 1e3

    def measure(self, fn, steps, use_graph=True, timed_s=TIMED_S, precondition_s=PRECONDITION_S, rounds=ROUNDS,
                min_nodes=MIN_GRAPH_NODES):
        """Per-step statistics of `steps` captured calls of fn, replayed R times per round (see module docstring).
        A graph shorter than `min_nodes` launches is captured several times over into one graph: two consecutive
        hipGraphLaunch calls leave the GPU idle for about 7 us (measured: 20-step graphs 3.92 us per step, a
        2000-step graph 3.57), which is launch machinery of the graph API and not part of a step."""
        import torch
        requested = steps
        steps = steps * max(1, -(-min_nodes // steps)) if use_graph else steps
        graphs = self.capture(fn, steps) if use_graph else None
        self._pass(graphs, fn, steps)                      # graph upload / first touch, untimed
        torch.cuda.synchronize()
        ms1, _ = self.event_ms(lambda: self._pass(graphs, fn, steps))   # a first estimate of one pass
        est = max(ms1 * 1e-3, 1e-7)
        t_end = time.perf_counter() + precondition_s       # precondition: clocks and caches in steady state
        n_pre = 0
        while time.perf_counter() < t_end or n_pre < 2:
            for _ in range(max(1, int(0.005 / est))):
                self._pass(graphs, fn, steps)
            torch.cuda.synchronize()
            n_pre += 1
        replays = max(1, int(np.ceil(timed_s / est)))
        per_step_us, wall_ms = [], 0.0
        for _ in range(rounds):
            ms, wall = self.event_ms(lambda: [self._pass(graphs, fn, steps) for _ in range(replays)])
            per_step_us.append(ms * 1e3 / (steps * replays))
            wall_ms += wall
        per_step_us = np.array(per_step_us)
        if graphs:
            for g in set(g.value for g in graphs):
                self.capi.check(self.l.mispmm_graph_destroy(ctypes.c_void_p(g)))
        return {"median_us": float(np.median(per_step_us)), "min_us": float(per_step_us.min()),
                "max_us": float(per_step_us.max()), "replays": replays * (steps // requested), "rounds": rounds,
                "graph_nodes": steps if use_graph else 0,
                "wall_us": wall_ms * 1e3 / (steps * replays * rounds)}


def load_traffic(key, kernel_tag):
    """L2<->fabric bytes per launch from the committed rocprofv3 PMC passes -- only when the entry was taken on
    the kernel that just ran (PMC counters cannot be read from inside this process)."""
    try:
        with open(TRAFFIC_JSON) as f:
            entry = json.load(f).get(key)
    except (OSError, ValueError):
        return None, "no committed PMC entry"
    if not entry:
        return None, "no committed PMC entry for this workload"
    if entry.get("kernel_tag") != kernel_tag:
        return None, f"committed PMC entry is for kernel {entry.get('kernel_tag')!r}, this run used {kernel_tag!r}"
    return entry["total_bytes"], entry.get("source", TRAFFIC_JSON)


def live_traffic(args, operand_sets=1):
    """L2<->fabric bytes per launch of THIS configuration (operand_sets = 0: of its HBM-streamed form, the main loop rotating
    over (B, C) pairs), measured now: two children of this script under
    `rocprofv3 --pmc` (one counter group per pass, no trace domains), eager launches, no extras; counters corrected as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and reads half of wide coalesced reads on gfx950; WRITE_SIZE
    in KiB is exact); first quarter of the dispatches dropped.  Called before this process touches the GPU.  Returns
    (bytes, kernel_tag, note) or (None, None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, None, "rocprofv3 not on PATH"
    # sys.executable = the interpreter binary that runs this process (a `python3` found on PATH may be a shim script: one more
    # exec hop under the profiler, or another environment without torch)
    child = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--steps", str(args.steps), "--warmup", str(args.warmup),
             "--kernel", str(args.kernel), "--acc", args.acc, "--launch", "eager", "--no-extras", "--no-cpu-baseline", "--placements", "1",
             "--operand-sets", str(operand_sets)]
    if args.matrix:
        child += ["--matrix", args.matrix]
    if args.k_cols:
        child += ["--k-cols", str(args.k_cols)]
    means, tag = {}, None
    tmp = tempfile.mkdtemp(prefix="mispmm_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            proc = subprocess.Popen(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, text=True, cwd="/tmp", start_new_session=True,
                                    # MISPMM_AUTOTUNE=0: under the profiler (serialised, slower launches) the plan-vs-storage-order measurement
                                    # can come out differently from the parent's; the footprint rule gives the parent's choice on every
                                    # BASELINE configuration, and the kernel tags are compared anyway before the figure is used
                                    env=dict(os.environ, TMPDIR="/tmp", MISPMM_AUTOTUNE="0"))
            try:
                stdout, stderr = proc.communicate(timeout=150)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)          # exactly the session started above: the profiler and its child
                proc.communicate()
                return None, None, f"rocprofv3 --pmc {counter} timed out"
            if proc.returncode != 0:
                return None, None, f"rocprofv3 --pmc {counter} failed ({proc.returncode}): {stderr[-300:]}"
            try:
                tag = json.loads(stdout.strip().splitlines()[-1])["config"]["kernel_tag"]
            except Exception:  # noqa: BLE001
                return None, None, "the profiled child printed no line"
            per_kernel = {}
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] == counter:
                            per_kernel.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not per_kernel:
                return None, None, f"no {counter} samples in the rocprofv3 output"
            vals = max(per_kernel.values(), key=len)            # the workload's kernel = the one with the most dispatches
            vals = vals[len(vals) // 4:]
            means[counter] = sum(vals) / len(vals)
    except Exception as e:  # noqa: BLE001  (never let the optional measurement take the line down)
        return None, None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = int(means["FETCH_SIZE"] * 1024 * 2) + int(means["WRITE_SIZE"] * 1024)
    return total, tag, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE ({means['FETCH_SIZE']:.1f} KiB x 2 on gfx950) and --pmc WRITE_SIZE "
                        f"({means['WRITE_SIZE']:.1f} KiB) in separate passes of the same configuration (eager launches, first quarter dropped)")


def stream_loop(w, stream, nsets):
    """(fn, launches, nsets, pairs): fn enqueues the workload's next step on the next of `nsets` distinct (B, C) pairs
    (0 = as many as STREAM_BYTES takes, at least 4); `launches` = a whole number of rotations of about MIN_GRAPH_NODES steps,
    so that a graph of them can be replayed back to back without ever re-using a pair before all the others have passed."""
    import itertools
    import torch
    if nsets <= 0:
        nsets = max(4, -(-STREAM_BYTES // w.pair_bytes()))
    pairs = w.make_pairs(nsets)
    counter = itertools.count()

    def fn():
        w.step(stream, pair=pairs[next(counter) % nsets])

    for _ in range(nsets):                 # first touch of every pair, and the counter back at a multiple of nsets
        fn()
    torch.cuda.synchronize()
    return fn, nsets * max(1, -(-MIN_GRAPH_NODES // nsets)), nsets, pairs


def run_single(args):
    # children only: nothing in this process has touched the GPU yet
    want_live = not (args.no_live_traffic or args.no_extras or args.launch != "graph")
    live = live_traffic(args, args.operand_sets) if want_live else None
    live_stream = live_traffic(args, 0) if want_live and args.operand_sets == 1 and args.hbm_streaming != "off" else None
    import torch
    from mispmm import capi
    capi.lib()
    torch.cuda.set_device(0)
    w = make_workload(args)
    stream = torch.cuda.Stream()
    timer = Timer(stream)
    rotating = args.operand_sets != 1

    def step():
        w.step(stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    kernel_tag = capi.last_kernel()

    stats = []
    if rotating:
        # --operand-sets: the main loop itself is the HBM-streamed form (profiler children, experiments)
        fn, launches, nsets, pairs = stream_loop(w, stream, args.operand_sets)
        stats.append(timer.measure(fn, launches, use_graph=args.launch == "graph", min_nodes=1))
    else:
        # The same K steps on `--placements` fresh copies of B and C: where the gathered operand sits in memory moves the time
        # of one binary on one box by up to 6 % (profiles/r3/placement_probe.log), so a single allocation is a draw from that
        # range; the line reports the MEDIAN copy and lists them all.
        for i in range(max(1, args.placements)):
            if i:
                w.fresh_operands()
                step()
                torch.cuda.synchronize()
            stats.append(timer.measure(step, args.steps, use_graph=args.launch == "graph"))
    stat = sorted(stats, key=lambda s: s["median_us"])[(len(stats) - 1) // 2]
    step()                                # leave the timed mode's result in C for the parity check
    torch.cuda.synchronize()
    got = w.host_result()

    launch_us = stat["median_us"]
    achieved = w.abytes / (launch_us * 1e-6) / 1e9
    traffic, traffic_src = load_traffic(f"{args.config}:{w.matrix}/{w.n}/{args.acc}" + ("/nohint" if os.environ.get("MISPMM_NO_HINT") == "1" else ""),
                                        kernel_tag)
    if live is not None:
        if live[0] is not None and live[1] == kernel_tag:
            traffic, traffic_src = live[0], live[2]
        else:
            traffic_src = f"{traffic_src} (live measurement unavailable: {live[2] if live[0] is None else 'kernel tag differs'})"
    info = capi.device_info(0)
    out = {
        "metric": metric_label(w),
        "value": round(w.flops / (launch_us * 1e-6) / 1e9, 2), "unit": "GFLOP/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(launch_us * 1e-3, 6),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": w.dtype,
        "data": f"SuiteSparse {w.matrix} (reference data/{w.label.split(' ')[0]}) x seeded synthetic B",
        "config": {"workload": w.workload, "baseline_config": args.config, "kernel": args.kernel,
                   "kernel_tag": kernel_tag, "acc_mode": args.acc if w.has_fast else "fp32 accumulate (MFMA)",
                   "launch": args.launch, "device": info["name"], **w.extra_config},
        "timing": {"method": f"the {args.steps} steps captured into hipGraphs of {stat['graph_nodes']} launches, the steps "
                             f"replayed {stat['replays']}x per round inside one HIP-event pair, {stat['rounds']} rounds, after "
                             f">= {int(PRECONDITION_S * 1e3)} ms of untimed replays; value / ms_per_step / roofline use the median round",
                   "replays": stat["replays"], "rounds": stat["rounds"], "median_us": round(stat["median_us"], 4),
                   "min_us": round(stat["min_us"], 4), "max_us": round(stat["max_us"], 4),
                   "host_wall_us_per_step": round(stat["wall_us"], 4),
                   "placements_us": [round(x["median_us"], 4) for x in stats],
                   "placements_note": f"the timed steps were run on {len(stats)} freshly allocated copies of B and C (median round of each "
                                      "listed in allocation order); value / ms_per_step / roofline are those of the MEDIAN copy: "
                                      "operand placement alone moves these kernels by up to 6 %"},
        "achieved_hbm_GBps": round(achieved, 1),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes_per_launch": w.abytes, "launch_us": round(launch_us, 4),
                     "frac_at_min": round(w.abytes / (stat["min_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "launch_us = HIP-event time of the timed region / launches in it (inter-kernel gaps "
                             "included); traffic = L2<->fabric bytes per launch from rocprofv3 PMC: " + str(traffic_src),
                     "residency": "the steady-state loop multiplies the same A, B into the same C: the working set "
                                  f"({w.abytes / 1e6:.0f} MB) never leaves the chip -- every XCD's slice of B stays in its 4 MiB L2 from launch "
                                  "to launch where it fits (the L2 is kept across the kernels of a graph: tools/micro/l2_retention.hip), "
                                  "the rest in the 256 MiB Infinity Cache -- so `frac` is algorithmic bytes per second against the 8 TB/s "
                                  "HBM peak (the metric BASELINE.json defines), not measured HBM traffic, and `traffic` (PMC under the "
                                  "profiler: every dispatch starts with cold L2s) is an upper bound for this loop; `hbm_streaming` is the "
                                  "same kernel with B and C streamed from / to HBM (DESIGN.md section 5.6: L2 / Infinity Cache / HBM regimes)"},
    }
    out["roofline"]["traffic_source"] = ("live" if live is not None and live[0] is not None and live[1] == kernel_tag
                                         else "committed fallback" if traffic is not None else None)
    if rotating:
        out["config"]["operand_sets"] = nsets
        out["roofline"]["residency"] = (f"the timed loop rotates over {nsets} distinct (B, C) pairs ({nsets * w.pair_bytes() / 1e6:.0f} MB): "
                                        "B streams from HBM, C streams to it; only A stays cached")
        del pairs
    elif args.hbm_streaming == "on" or (args.hbm_streaming == "auto" and not args.no_extras):
        # The HBM-streamed figure (module docstring): the same kernel over S distinct (B, C) pairs in rotation, 2 x the Infinity
        # Cache of them, one graph of a whole number of rotations replayed back to back inside one HIP-event pair.
        fn, launches, nsets, pairs = stream_loop(w, stream, 0)
        st = timer.measure(fn, launches, use_graph=args.launch == "graph", min_nodes=1, rounds=3)
        s_traffic, s_src = None, "not measured (--no-live-traffic / --no-extras)"
        if live_stream is not None:
            s_traffic, s_src = (live_stream[0], live_stream[2]) if live_stream[0] is not None else (None, live_stream[2])
        out["hbm_streaming"] = {
            "launch_us": round(st["median_us"], 4), "min_us": round(st["min_us"], 4), "max_us": round(st["max_us"], 4),
            "achieved": round(w.abytes / (st["median_us"] * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(w.abytes / (st["median_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "value": round(w.flops / (st["median_us"] * 1e-6) / 1e9, 2), "value_unit": "GFLOP/s",
            "operand_sets": nsets, "bytes_in_rotation": int(nsets * w.pair_bytes()), "launches_per_graph": st["graph_nodes"],
            "replays": st["replays"], "algorithmic_bytes_per_launch": w.abytes, "kernel_tag": capi.last_kernel(),
            "traffic": s_traffic, "traffic_note": s_src,
            "note": f"the same kernel over {nsets} distinct (B, C) pairs in rotation ({nsets * w.pair_bytes() / 1e6:.0f} MB = 2 x the 256 MiB "
                    "Infinity Cache), a graph of a whole number of rotations replayed back to back: every launch reads its B from "
                    "HBM and streams its C to HBM, A stays cached.  Never `value` (BASELINE.json's metric is the resident loop). "
                    "It is also the cold-launch figure: N cold launches over N distinct operand sets inside ONE event pair, the "
                    "form of the reference's one-un-warmed-launch-per-kernel timing (engine.cpp:41-44) without event overhead"}
        del pairs, fn
        step()
        torch.cuda.synchronize()
    if w.fmt == "bsr":
        executed = w.executed_flops if w.which == "dense" else 2.0 * w.bsrc.num_steps * 16 * 32 * w.n
        ex = executed / (launch_us * 1e-6) / 1e12
        out["mfma"] = {"executed_TFLOPs": round(ex, 2), "dense_bf16_peak_frac": round(ex / BF16_MFMA_PEAK_TFLOPS, 4),
                       "note": "MFMA flops actually executed (K steps x 16 x 32 x N x 2); `value` counts the useful flops (2*nnz*K)"}
        out["roofline"]["bsr16_operand_bytes"] = w.bsr16_bytes
        out["roofline"]["frac_vs_bsr16_operand"] = round(w.bsr16_bytes / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        out["roofline"]["bytes_note"] = ("algorithmic_bytes_per_launch = the operand of the kernel that ran (column lists + bf16 "
                                         "tiles of the occupied columns) + B (bf16) + C (fp32); bsr16_operand_bytes = the same "
                                         "with the dense 16 x 16 bf16 blocks of the BSR file as A (SURVEY 8(d) config 4)")
    if not args.no_extras and w.fmt == "bsr":
        out["other_bsr_kernels"] = {}
        for other in ("slots", "steps", "dense"):
            if other == w.which:
                continue
            o = timer.measure(lambda: w.step(stream, which=other), min(args.steps, 500), rounds=3, precondition_s=0.01)
            out["other_bsr_kernels"][other] = {
                "kernel": w.kernel_names[other], "kernel_tag": capi.last_kernel(), "launch_us": round(o["median_us"], 4),
                "algorithmic_bytes_per_launch": w.kernel_bytes[other],
                "roofline_frac": round(w.kernel_bytes[other] / (o["median_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        w.step(stream)
        torch.cuda.synchronize()
    if not args.no_extras:
        if args.launch == "graph" and args.steps < MIN_GRAPH_NODES:
            g = timer.measure(step, args.steps, rounds=3, precondition_s=0.01, min_nodes=1)
            out["timing"]["graph_of_exactly_steps_us"] = round(g["median_us"], 4)
            out["timing"]["graph_of_exactly_steps_note"] = (
                f"the same steps as a {args.steps}-launch graph replayed back to back: the difference to median_us is the "
                "idle time between two hipGraphLaunch calls, spread over the steps of one graph")
        if w.has_fast:
            other = "fast" if args.acc == "reference" else "reference"
            o = timer.measure(lambda: w.step(stream, acc=other), min(args.steps, 500), rounds=3, precondition_s=0.01)
            out["other_acc_mode"] = {
                "acc_mode": other, "launch_us": round(o["median_us"], 4),
                "roofline_frac": round(w.abytes / (o["median_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "reference = the reference engine's accumulate (bit-exact); fast = fp32 fma chain (<= 1e-5)"}
        if w.fmt == "csr" and args.kernel in (0, 5):
            # the batched entry point (mispmm_csr_batch_f32): 8 different dense operands per launch, same A.  Never
            # `value`: the metric is one product per step; this shows what the ~1.2 us between two dependent launches
            # costs a product of this size
            from mispmm import ops, synth
            nb = 8
            bs = [w.b] + [torch.from_numpy(synth.dense_b(w.csr.num_cols, w.n, seed=1000 + i)).cuda() for i in range(1, nb)]
            cs = [torch.empty_like(w.c) for _ in range(nb)]
            bt = timer.measure(lambda: ops.spmm_csr_batch(w.a, bs, outs=cs, acc=args.acc, stream=stream), max(1, min(args.steps, 250)),
                               rounds=3, precondition_s=0.01)
            torch.cuda.synchronize()
            per = bt["median_us"] / nb
            same = bool(np.array_equal(cs[0].cpu().numpy(), got))
            out["batched"] = {"operands_per_launch": nb, "us_per_product": round(per, 4), "launch_us": round(bt["median_us"], 4),
                              "roofline_frac": round(w.abytes / (per * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                              "first_result_equals_single_launch": same, "kernel_tag": capi.last_kernel(),
                              "note": "mispmm_csr_batch_f32: one launch multiplies 8 different dense operands by the same A; algorithmic "
                                      "bytes per product against the 8 TB/s HBM peak (8 operands = the Infinity-Cache regime of DESIGN.md "
                                      "section 5.6, where a single launch takes longer than the L2-resident loop behind `value`)"}
            del bs, cs
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds, got, args.acc)
    emit(out)


# ------------------------------------------------------------------------------------------ N > 1
def run_multi(args):
    import torch
    import torch.distributed as dist
    from mispmm import capi, datasets, dist as mdist, synth
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:                      # a lone rank rehearsing the path: any free port
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    shared_gpu = os.environ.get("MISPMM_SHARE_GPU") == "1"        # rehearsal: several ranks on one card (gloo + IPC)
    dev_index = 0 if shared_gpu else local % max(1, ndev)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if shared_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    capi.lib()
    if args.config == "2":
        # medium_4096 is a 4 MB problem whose step is the launch boundary: sharding it measures nothing
        raise SystemExit("bench: --gpus N shards --config headline | 3 | 4 | 5; --config 2 is a single-GPU configuration")
    from mispmm import formats, ops
    fmt = {"3": "ell", "4": "bsr"}.get(args.config, "csr")
    cfg_matrix = args.matrix or ("ACTIVSg10K" if fmt == "bsr" else "n4c6-b13")
    n = args.k_cols or {"headline": 128, "3": 256, "4": 128, "5": 512}[args.config]
    csr = datasets.load_csr(cfg_matrix)
    bucket = args.bucket if args.bucket > 0 else 64
    modes = ["allgather", "peer"] if args.exchange == "both" else [args.exchange]
    b_host = synth.dense_b(csr.num_cols, n) if rank == 0 else None
    flops = datasets.spmm_flops(csr.nnz, n)
    results, whole, gathered_host = {}, None, None
    # what is sharded, how a rank's driver is built, what the unsharded single-GPU product is (the checker of the exchange),
    # and the algorithmic bytes of a shard -- per format: CSR by rows, ELL by rows, bf16 BSR-16 by block rows
    if fmt == "csr":
        operand, abytes = csr, datasets.csr_algorithmic_bytes(csr, n)
        shard_bounds = mdist.shard_bounds(csr.row_ptrs, world)
        kind, esize = "CSR", 4

        def make_job(mode, **kw):
            return mdist.ShardedCsrSpmm(csr, n, device=device, kernel=args.kernel, acc=args.acc, bucket=bucket, exchange=mode, **kw)

        def unsharded(b_dev):
            return ops.spmm_csr(ops.DeviceCSR.from_host(csr, device=device), b_dev, kernel=args.kernel, acc=args.acc)

        def shard_a_bytes(r0, r1, e0, e1):
            return (e1 - e0) * 8 + (r1 - r0 + 1) * 4
    elif fmt == "ell":
        operand = ops.colmajor_ell_to_rowmajor(formats.csr_to_ell_colmajor(csr))
        abytes = datasets.ell_algorithmic_bytes(csr.num_rows, operand.width, csr.num_cols, n)
        shard_bounds = mdist.ShardedEllSpmm._partition(operand, world)
        kind, esize = "ELL", 4

        def make_job(mode, **kw):
            return mdist.ShardedEllSpmm(operand, n, device=device, kernel=args.kernel, acc=args.acc, bucket=bucket, exchange=mode, **kw)

        def unsharded(b_dev):
            return ops.spmm_ell(ops.DeviceELL.from_host(operand, device=device), b_dev, kernel=args.kernel, acc=args.acc)

        def shard_a_bytes(r0, r1, e0, e1):
            return (r1 - r0) * operand.width * 8
    else:
        operand = formats.csr_to_bsr(csr, 16)
        shard_bounds = mdist.ShardedBsrcSlotsSpmm._partition(operand, world)
        kind, esize = "BSR block 16", 2
        abytes = None                                             # the bytes of the kernel's operand: summed over the shards below

        def make_job(mode, **kw):
            return mdist.ShardedBsrcSlotsSpmm(operand, n, device=device, bucket=bucket, exchange=mode, **kw)

        def unsharded(b_dev):
            return ops.spmm_bsrc_slots_bf16(ops.DeviceBSRCSlots.from_host(operand, device=device), b_dev, out_bf16=False)

        def shard_a_bytes(r0, r1, e0, e1):
            from mispmm.multi import bsr_block_row_slice
            local = ops.DeviceBSRCSlots.from_host(bsr_block_row_slice(operand, r0 // 16, r1 // 16), device=device)
            return local.operand_bytes()
    # who took part: every rank reports its device (PCI bus id) and the algorithmic bytes of ITS shard -- its slice of A,
    # the B rows its columns touch, its C slab ("multi-GPU per device ... node total = sum")
    r0, r1 = int(shard_bounds[rank]), int(shard_bounds[rank + 1])
    e0, e1 = int(csr.row_ptrs[min(r0, csr.num_rows)]), int(csr.row_ptrs[min(r1, csr.num_rows)])
    touched = int(np.unique(csr.col_idxs[e0:e1]).size)
    mine = {"rank": rank, "device": dev_index, "bus_id": capi.device_bus_id(dev_index), "host": os.uname().nodename,
            "rows": r1 - r0, "nnz": e1 - e0, "b_rows_touched": touched,
            "algorithmic_bytes": int(shard_a_bytes(r0, r1, e0, e1) + touched * n * esize + (r1 - r0) * n * 4)}
    if abytes is None:                                            # config 4 on one GPU: the unsharded operand of the same kernel
        abytes = int(ops.DeviceBSRCSlots.from_host(operand, device=device).operand_bytes() + csr.num_cols * n * 2 + csr.num_rows * n * 4)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    extras = {}                      # legs measured beside the exchange modes (kernel_only_batched)
    shard_kernel_tag = {}            # mispmm_last_kernel() of this rank's shard launches

    def compose(partial=False):
        """rank 0: the contract line from the exchange modes measured so far"""
        usable = {m: r for m, r in results.items() if "value" in r}
        if not usable:
            raise SystemExit(f"bench: no exchange mode could run: {results}")
        best = max(usable, key=lambda m: usable[m]["value"])
        r = usable[best]
        label = {"n4c6-b13": "large_25605", "ACTIVSg10K": "large_20000"}.get(cfg_matrix, cfg_matrix)
        node_bytes = int(sum(e["algorithmic_bytes"] for e in everyone))
        devices_seen = sorted({(e["host"], e["bus_id"]) for e in everyone})
        ko_s, e2e_s = r["kernel_only_ms_per_step"] * 1e-3, r["ms_per_step"] * 1e-3
        out = {
            "metric": f"SpMM GFLOP/s, {label} ({cfg_matrix}) {kind} x dense K={n} {'bf16' if fmt == 'bsr' else 'fp32'}",
            "value": r["value"], "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16" if fmt == "bsr" else "f32",
            "data": f"SuiteSparse {cfg_matrix} (reference data/{label}) x seeded synthetic B",
            "config": {"workload": f"{cfg_matrix} {kind} {csr.num_rows}x{csr.num_cols} nnz {csr.nnz} x dense K={n} {'bf16, C fp32' if fmt == 'bsr' else 'fp32'}",
                       "parallelism": f"{'block-row' if fmt == 'bsr' else 'row'}-sharded x{world}, B replicated, C slabs exchanged every {bucket} steps "
                                      f"({best}: " + ("RCCL all_gather_into_tensor" if best == "allgather" else
                                                      "direct copies into IPC-mapped peer buffers over xGMI") + ")",
                       "kernel": args.kernel, "kernel_tag": shard_kernel_tag.get("tag"),
                       "acc_mode": args.acc if fmt != "bsr" else "fp32 accumulate (MFMA)", "format": fmt,
                       "check": "exchanged C == unsharded single-GPU C (bitwise) on every rank"},
            "ranks_seen": {"world_size": world, "distinct_devices": len(devices_seen),
                           "devices": [f"{h}/{b}" for h, b in devices_seen],
                           "backend": "gloo (ranks share one card: MISPMM_SHARE_GPU rehearsal)" if shared_gpu else "nccl (RCCL)",
                           "per_rank": everyone},
            "achieved_hbm_GBps": round(abytes / e2e_s / 1e9, 1),
            "exchange_modes": results,
            "kernel_only": {"value": r["kernel_only_value"], "unit": "GFLOP/s", "ms_per_step": r["kernel_only_ms_per_step"],
                            "note": "the same steps re-run with C left row-sharded (no exchange) into ONE resident slab per rank -- "
                                    "the form of the N = 1 line's loop, which overwrites one C -- max over ranks"},
            "roofline": {"bound": "hbm", "achieved": round(node_bytes / ko_s / 1e9, 1),
                         "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": round(node_bytes / ko_s / 1e9 / (HBM_PEAK_GBS * world), 4),
                         "frac_end_to_end": round(node_bytes / e2e_s / 1e9 / (HBM_PEAK_GBS * world), 4),
                         "traffic": None, "algorithmic_bytes_per_launch": node_bytes,
                         "single_gpu_algorithmic_bytes": abytes,
                         "note": "algorithmic bytes = sum over ranks of (A slice + B rows its columns touch + C slab); "
                                 "frac = those bytes / kernel-only time (C left row-sharded) against the N-GPU aggregate "
                                 "HBM peak, frac_end_to_end = the same bytes / ms_per_step (exchange included)"},
        }
        out.update(extras)
        if partial:
            out["partial"] = "an exchange mode did not finish: the line holds what was measured before it"
        if not args.no_cpu_baseline:
            # the same CPU leg as at N = 1: the oracle, 1 thread, rank 0's host, and the checker of the exchanged C
            if "cpu" not in cache:
                if fmt == "csr":
                    shim = CsrWorkload.__new__(CsrWorkload)
                elif fmt == "ell":
                    shim = EllWorkload.__new__(EllWorkload)
                    shim.ellc = formats.csr_to_ell_colmajor(csr)
                else:
                    shim = BsrBf16Workload.__new__(BsrBf16Workload)
                    shim.bsr, shim.block = operand, 16
                    shim.a16_host = synth.bf16_round(operand.data.reshape(-1)).reshape(operand.data.shape)
                    shim.b16_host = synth.bf16_round(b_host.reshape(-1)).reshape(b_host.shape)
                shim.csr, shim.b_host, shim.flops, shim.n = csr, b_host, flops, n
                shim.workload = out["config"]["workload"]
                cache["cpu"] = cpu_baseline(shim, args.cpu_seconds, gathered_host, args.acc)
            out["cpu_baseline"] = cache["cpu"]
        return out

    cache = {}

    def persist():
        """rank 0, after every measured mode: keep the line composed so far where the parent (spawn_ranks) finds it should a
        LATER mode take this rank down (a fault, as opposed to a hang, kills the process before it can print anything)."""
        path = os.environ.get("MISPMM_BENCH_PARTIAL")
        if rank == 0 and path and any("value" in r for r in results.values()):
            try:
                with open(path + ".tmp", "w") as f:
                    json.dump(compose(partial=True), f)
                os.replace(path + ".tmp", path)
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"bench: could not persist the partial line: {e}\n")

    def run_mode(mode):
        nonlocal whole, gathered_host
        stall = os.environ.get("MISPMM_BENCH_STALL", "")   # test hooks (tests/test_gpu_multi.py): this mode never finishes / dies
        if stall == mode:
            time.sleep(3600)
        if stall == mode + ":die":
            os._exit(9)
        try:
            job = make_job(mode, debug_sentinel=os.environ.get("MISPMM_DIST_DEBUG") == "1")
        except Exception as e:  # noqa: BLE001  (the peer path needs IPC mapping between the ranks' devices)
            results[mode] = {"unavailable": f"{type(e).__name__}: {e}"}
            job = None
        # every rank must agree on whether the mode is usable
        flag = torch.tensor([1.0 if job is not None else 0.0], device="cpu" if shared_gpu else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag[0]) == 0.0:
            if job is not None:
                job.close()
                results[mode] = {"unavailable": "another rank could not set the exchange up"}
            return
        job.broadcast_b(b_host)                      # one-time, outside the timed region
        job.run(args.warmup)
        job.finish()
        shard_kernel_tag["tag"] = capi.last_kernel()

        def timed(total_steps, gather):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            job.run(total_steps, gather=gather)
            job.finish(gather=gather)
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if shared_gpu else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)      # every rank uses the same (slowest) figure
            return float(t[0])

        # as at N = 1: the K steps are repeated until the timed region lasts >= 20 ms (a single pass of a small K times
        # the launch latency of one bucket graph and one collective, not a step); `replays` says how often
        est = timed(args.steps, True)
        replays = max(1, int(np.ceil(TIMED_S / max(est, 1e-6))))
        unit = bucket // int(np.gcd(args.steps, bucket))       # replays that make K * R a whole number of buckets
        replays = -(-replays // unit) * unit
        wall = timed(args.steps * replays, True) / replays
        # the same steps with C left row-sharded (no exchange): the kernel-only figure.  One untimed bucket first: the graph of
        # the kernel-only steps is captured and instantiated on first use (a 500-launch graph: milliseconds of host time)
        job.run(job.bucket, gather=False)
        job.finish(gather=False)
        compute_s = timed(args.steps * replays, False) / replays
        # exchanged C must equal the unsharded single-GPU product bit for bit (row independence), on every rank
        job.run(1)
        job.finish()
        if whole is None:
            whole = unsharded(job.b)
            torch.cuda.synchronize()
        ok = torch.tensor([1.0 if torch.equal(whole, job.gathered_c()) else 0.0], device="cpu" if shared_gpu else device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok[0]) == 0.0:
            # no number for an exchange that delivered wrong bytes; the other modes are still measured and the line says so
            results[mode] = {"failed": "exchanged C differs from the unsharded single-GPU product on at least one rank"}
            job.close()
            return
        if rank == 0:
            gathered_host = job.gathered_c().cpu().numpy()
        results[mode] = {"replays": replays,
                         "value": round(flops * args.steps / wall / 1e9, 2), "ms_per_step": round(wall * 1e3 / args.steps, 6),
                         "kernel_only_value": round(flops * args.steps / compute_s / 1e9, 2),
                         "kernel_only_ms_per_step": round(compute_s * 1e3 / args.steps, 6)}
        if job.debug_sentinel:
            # MISPMM_DIST_DEBUG=1: every wait for a bucket checked that all peers' sentinels (written behind their slabs)
            # had already arrived -- the timing of such a run is not a measurement (host syncs per bucket)
            results[mode]["sentinel_checks"] = job.sentinel_checks
        job.close()

    def start_watchdog(mode):
        """A later exchange mode that does not finish (the peer exchange has only ever run between processes on one card)
        must not cost the line the earlier modes earned: after --mode-timeout seconds rank 0 prints the line with what was
        measured, marked `partial`, and every rank leaves with exit code WATCHDOG_EXIT -- a hang on the GPU is a defect, not
        a success, so the run is reported as failed although its line is there."""
        import threading

        def expire():
            results[mode] = {"unavailable": f"did not finish within {args.mode_timeout} s (watchdog); the line reports the modes before it"}
            if rank == 0:
                try:
                    emit(compose(partial=True))
                    sys.stdout.flush()
                finally:
                    os._exit(WATCHDOG_EXIT)
            time.sleep(30)                     # rank 0 prints first
            os._exit(WATCHDOG_EXIT)

        t = threading.Timer(args.mode_timeout, expire)
        t.daemon = True
        t.start()
        return t

    def run_batched():
        """kernel_only_batched: the kernel-only steps with `--batch` dense operands per launch (mispmm_csr_batch_f32 on every
        rank's shard, C left row-sharded in resident slabs) -- the launch boundary once per `batch` products."""
        job = mdist.ShardedCsrSpmm(csr, n, device=device, kernel=args.kernel, acc=args.acc, bucket=bucket, exchange="allgather", batch=args.batch)
        job.broadcast_b(b_host)
        job.run(job.bucket, gather=False)
        job.finish(gather=False)

        def timed(total_steps):
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            job.run(total_steps, gather=False)
            job.finish(gather=False)
            torch.cuda.synchronize()
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if shared_gpu else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0])

        per_pass = -(-args.steps // job.bucket) * job.bucket          # whole buckets (graph replays) only
        est = timed(per_pass)
        replays = max(1, int(np.ceil(TIMED_S / max(est, 1e-6))))
        t = timed(per_pass * replays) / (per_pass * replays)
        ok = True
        if whole is not None:                                         # every operand's slab == the rows of the unsharded product
            ok = all(torch.equal(job.local_slab(i), whole[r0:r1]) for i in range(job.batch))
        flag = torch.tensor([1.0 if ok else 0.0], device="cpu" if shared_gpu else device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        job.close()
        if float(flag[0]) == 0.0:
            extras["kernel_only_batched"] = {"failed": "a batched slab differs from the unsharded product"}
            return
        extras["kernel_only_batched"] = {"operands_per_launch": job.batch, "value": round(flops / t / 1e9, 2), "unit": "GFLOP/s",
                                         "ms_per_step": round(t * 1e3, 6), "replays": replays,
                                         "note": f"kernel-only steps, {job.batch} dense operands (replicas of B in buffers of their own) per launch "
                                                 "on every rank's shard, C left row-sharded; time per PRODUCT, max over ranks.  Never `value`"}

    for mode in modes:
        watchdog = start_watchdog(mode) if args.mode_timeout > 0 and any("value" in r for r in results.values()) else None
        try:
            run_mode(mode)
        finally:
            if watchdog is not None:
                watchdog.cancel()
        persist()
        dist.barrier()       # no rank enters the next mode (where a fault may take the job down) before rank 0 has persisted this one
    if args.batch > 1 and fmt == "csr" and any("value" in r for r in results.values()):
        try:
            run_batched()
        except Exception as e:  # noqa: BLE001  (an optional leg never costs the line)
            extras["kernel_only_batched"] = {"unavailable": f"{type(e).__name__}: {e}"}
    if rank == 0:
        emit(compose())
    dist.barrier()
    dist.destroy_process_group()


def spawn_ranks(args):
    """`python bench.py --gpus N` started plainly: start one child per rank (fresh processes; this parent never
    touches the GPU), relay rank 0's JSON line, exit non-zero if any rank failed.  Every child is watched: when one
    dies the others are terminated (a rank that lost a peer would sit in a collective until the RCCL watchdog fires);
    if the rendezvous port was taken (bind-then-close can collide), a fresh set of children is started on a new port."""
    import socket
    import tempfile

    def free_port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]

    partial_path = os.path.join(tempfile.gettempdir(), f"mispmm_bench_partial_{os.getpid()}.json")
    for attempt in range(3):
        port = free_port()
        procs, errs = [], []
        if os.path.exists(partial_path):
            os.remove(partial_path)
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", MISPMM_BENCH_CHILD="1", MISPMM_BENCH_PARTIAL=partial_path)
            err = tempfile.TemporaryFile()
            errs.append(err)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err))
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            for i, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = i
                    break
            time.sleep(0.05)
        if failed is None:
            failed = next((i for i, p in enumerate(procs) if p.returncode != 0), None)
        if failed is not None:
            for p in procs:                       # exactly the children started above, nothing else
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        out = procs[0].stdout.read() if procs[0].stdout else b""
        texts = []
        for err in errs:
            err.seek(0)
            texts.append(err.read().decode(errors="replace"))
            err.close()
        if failed is None:
            if os.path.exists(partial_path):
                os.remove(partial_path)
            sys.stderr.write(texts[0])
            os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, out)
            sys.exit(0)
        collided = any("EADDRINUSE" in t or "address already in use" in t.lower() for t in texts)
        sys.stderr.write(f"bench: rank {failed} exited with {procs[failed].returncode}; the other ranks were stopped\n")
        sys.stderr.write(texts[failed][-4000:])
        if not (collided and attempt < 2):
            # what was measured before the failure is still printed (marked `partial`) -- by rank 0 itself when its watchdog
            # fired, else from the line it persisted after its last finished mode -- and the exit code stays non-zero
            line = out if out.strip() else (open(partial_path, "rb").read() + b"\n" if os.path.exists(partial_path) else b"")
            if os.path.exists(partial_path):
                os.remove(partial_path)
            if line.strip():
                os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, line)
            sys.exit(procs[failed].returncode or 1)
        sys.stderr.write("bench: the rendezvous port was taken -- starting fresh ranks on another port\n")
    sys.exit(1)


def main():
    args = parse()
    # libraries print banners on stdout (RCCL its version block, gloo its connection summary): the contract is ONE JSON
    # line there, so everything but that line goes to stderr -- fd 1 is pointed at stderr and the line is written to the
    # saved descriptor at the end
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("MISPMM_FORCE_DIST") == "1"    # rehearse the distributed path with a single rank
    if args.gpus > 1 and world == 1 and not force:
        spawn_ranks(args)
    elif args.gpus > 1 or world > 1 or force:
        run_multi(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
