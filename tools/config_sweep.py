"""Time the BASELINE.json configurations that are parity-test cases rather than the bench line
(configs 2-5, single GPU), graph-replayed, HIP-event timed.  GPU box only.  One JSON line each."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, formats, ops, synth  # noqa: E402

ITERS, ROUNDS = 300, 5


def timed(fn, stream):
    l = capi.lib()
    sp = ctypes.c_void_p(stream.cuda_stream)
    fn()
    torch.cuda.synchronize()
    capi.check(l.mispmm_graph_begin(sp))
    for _ in range(ITERS):
        fn()
    g = ctypes.c_void_p()
    capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
    capi.check(l.mispmm_graph_launch(g, sp))
    torch.cuda.synchronize()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    capi.check(l.mispmm_event_create(ctypes.byref(e0)))
    capi.check(l.mispmm_event_create(ctypes.byref(e1)))
    ts, ms = [], ctypes.c_float()
    for _ in range(ROUNDS):
        capi.check(l.mispmm_event_record(e0, sp))
        capi.check(l.mispmm_graph_launch(g, sp))
        capi.check(l.mispmm_event_record(e1, sp))
        capi.check(l.mispmm_event_sync(e1))
        capi.check(l.mispmm_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        ts.append(ms.value * 1e3 / ITERS)
    capi.check(l.mispmm_graph_destroy(g))
    return float(np.median(ts))


def report(tag, us, flops, abytes, **extra):
    print(json.dumps({"config": tag, "us_per_spmm": round(us, 3), "GFLOPs": round(flops / us / 1e3, 1),
                      "algorithmic_GBps": round(abytes / us / 1e3, 1), "hbm_roofline_frac": round(abytes / us / 1e3 / 8000, 4),
                      **extra}), flush=True)


def main():
    s = torch.cuda.Stream()
    # config 2: medium_4096 (stand-in delaunay_n12) CSR x K=128
    csr = datasets.load_csr("delaunay_n12")
    a, b = ops.DeviceCSR.from_host(csr), torch.from_numpy(synth.dense_b(csr.num_cols, 128)).cuda()
    c = torch.empty((csr.num_rows, 128), device="cuda")
    for acc in ("reference", "fast"):
        us = timed(lambda: ops.spmm_csr(a, b, out=c, acc=acc, stream=s), s)
        report("2: medium_4096(delaunay_n12) CSR K=128 fp32", us, datasets.spmm_flops(csr.nnz, 128),
               datasets.csr_algorithmic_bytes(csr, 128), acc=acc)
    # config 3: large_25605 ELL x K=256
    csr = datasets.load_csr("n4c6-b13")
    ell = ops.DeviceELL.from_host(formats.csr_to_ell_colmajor(csr))
    b = torch.from_numpy(synth.dense_b(csr.num_cols, 256)).cuda()
    c = torch.empty((csr.num_rows, 256), device="cuda")
    for acc in ("reference", "fast"):
        us = timed(lambda: ops.spmm_ell(ell, b, out=c, acc=acc, stream=s), s)
        report("3: large_25605 ELL K=256 fp32", us, datasets.spmm_flops(csr.nnz, 256),
               datasets.ell_algorithmic_bytes(csr.num_rows, ell.width, csr.num_cols, 256), acc=acc, width=ell.width)
    # config 5 on one GPU: large_25605 CSR x K=512 ; plus the headline K=128 for reference
    a = ops.DeviceCSR.from_host(csr)
    for k in (128, 512):
        b = torch.from_numpy(synth.dense_b(csr.num_cols, k)).cuda()
        c = torch.empty((csr.num_rows, k), device="cuda")
        for acc in ("reference", "fast"):
            us = timed(lambda: ops.spmm_csr(a, b, out=c, acc=acc, stream=s), s)
            report(f"5/headline: large_25605 CSR K={k} fp32 (1 GPU)", us, datasets.spmm_flops(csr.nnz, k),
                   datasets.csr_algorithmic_bytes(csr, k), acc=acc)
    # config 4: large_20000 BSR-16 x K=128
    csr = datasets.load_csr("ACTIVSg10K")
    bsr = formats.csr_to_bsr(csr, 16)
    a = ops.DeviceBSR.from_host(bsr)
    bh = synth.dense_b(csr.num_cols, 128)
    b = torch.from_numpy(bh).cuda()
    c = torch.empty((csr.num_rows, 128), device="cuda")
    executed = 2.0 * bsr.num_blocks * 256 * 128
    useful = datasets.spmm_flops(csr.nnz, 128)
    for kernel, acc in ((1, "reference"), (1, "fast"), (2, "fast")):
        us = timed(lambda: ops.spmm_bsr(a, b, out=c, kernel=kernel, acc=acc, stream=s), s)
        report(f"4a: large_20000 BSR-16 K=128 fp32 kernel {kernel}", us, useful, datasets.bsr_algorithmic_bytes(bsr, 128),
               acc=acc, executed_TFLOPs=round(executed / us / 1e6, 2))
    nz = ops.bsr_nonzeros(bsr)
    for acc in ("reference", "fast"):
        us = timed(lambda: ops.spmm_bsr_nonzeros(nz, b, out=c, acc=acc, stream=s), s)
        report("4a: large_20000 BSR-16 K=128 fp32 kernel 3 (zero-skipping, non-zero list)", us, useful,
               nz.nnz * 8 + (csr.num_rows + 1) * 4 + csr.num_cols * 128 * 4 + csr.num_rows * 128 * 4, acc=acc)
    blocks16, b16 = ops.f32_to_bf16(a.data), ops.f32_to_bf16(b)
    bsrc = ops.DeviceBSRC.from_host(bsr)
    slots = ops.DeviceBSRCSlots.from_host(bsr)
    # bytes quoted per kernel: the operand that kernel reads (column lists + tiles of the occupied columns) + B (bf16) + C
    for out_bf16 in (False, True):
        c16 = torch.empty((csr.num_rows, 128), dtype=torch.int16 if out_bf16 else torch.float32, device="cuda")
        dense_bc = csr.num_cols * 128 * 2 + csr.num_rows * 128 * (2 if out_bf16 else 4)
        ex = 2.0 * bsrc.num_steps * 16 * 32 * 128
        us = timed(lambda: ops.spmm_bsrc_slots_bf16(slots, b16, out_bf16=out_bf16, out=c16, stream=s), s)
        report("4: large_20000 BSR-16 K=128 bf16 MFMA, compacted block rows in 4 step slots (workgroup per block row), C "
               + ("bf16" if out_bf16 else "fp32"), us, useful, slots.operand_bytes() + dense_bc, mfma_k_steps=slots.used_steps,
               executed_TFLOPs=round(ex / us / 1e6, 2))
        us = timed(lambda: ops.spmm_bsrc_bf16(bsrc, b16, out_bf16=out_bf16, out=c16, stream=s), s)
        report("4: large_20000 BSR-16 K=128 bf16 MFMA, compacted block rows (wave per block row), C " + ("bf16" if out_bf16 else "fp32"), us,
               useful, bsrc.num_steps * 1152 + (csr.num_rows // 16 + 1) * 4 + dense_bc, mfma_k_steps=bsrc.num_steps,
               executed_TFLOPs=round(ex / us / 1e6, 2))
    for out_bf16 in (False, True):
        c16 = torch.empty((csr.num_rows, 128), dtype=torch.int16 if out_bf16 else torch.float32, device="cuda")
        us = timed(lambda: ops.spmm_bsr_bf16(a, blocks16, b16, out_bf16=out_bf16, out=c16, stream=s), s)
        report("4: large_20000 BSR-16 K=128 bf16 MFMA, one B panel per block, C " + ("bf16" if out_bf16 else "fp32"), us, useful,
               datasets.bsr_algorithmic_bytes(bsr, 128, elem=2, out_elem=2 if out_bf16 else 4),
               executed_TFLOPs=round(executed / us / 1e6, 2), dense_bf16_mfma_peak_frac=round(executed / us / 1e6 / 2500, 4))


if __name__ == "__main__":
    main()
