"""Where does a wave of the bf16 BSR kernel (bsrc_slots_mfma_bf16, BASELINE config 4) spend its time?  Diagnostic build
of the library (-DMISPMM_STAMPS: s_memrealtime stamps, 100 MHz, into a side buffer), percentiles over the waves of the
LAST launch of a replayed graph.

  make -C cuda-optimization-for-spmm_amd stamps
  MISPMM_LIB=.../libmispmm_stamps.so python tools/stamp_bsr.py [--k-cols 128] [--launches 50]
GPU box only.  Stamps: 0 wave start, 1 column list arrived, 2 B rows + A tile arrived, 3 MFMAs done + partial tile in
LDS, 4 barrier passed, 5 stores issued, 6 stores drained."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, formats, ops, synth  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--matrix", default="ACTIVSg10K")
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--launches", type=int, default=50)
    a = p.parse_args()
    l = capi.lib()
    if not hasattr(l, "mispmm_debug_set_stamps_bsr"):
        raise SystemExit("this library was not built with -DMISPMM_STAMPS (set MISPMM_LIB)")
    l.mispmm_debug_set_stamps_bsr.argtypes = [ctypes.c_void_p]
    l.mispmm_debug_set_stamps_bsr.restype = ctypes.c_int
    csr = datasets.load_csr(a.matrix)
    bsr = formats.csr_to_bsr(csr, 16)
    slots = ops.DeviceBSRCSlots.from_host(bsr)
    b16 = ops.f32_to_bf16(torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols)).cuda())
    c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
    buf = torch.zeros((1 << 16, 8), dtype=torch.int64, device="cuda")
    capi.check(l.mispmm_debug_set_stamps_bsr(ctypes.c_void_p(buf.data_ptr())))
    stream = torch.cuda.Stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    ops.spmm_bsrc_slots_bf16(slots, b16, out=c, stream=stream)
    torch.cuda.synchronize()
    capi.check(l.mispmm_graph_begin(sp))
    for _ in range(a.launches):
        ops.spmm_bsrc_slots_bf16(slots, b16, out=c, stream=stream)
    g = ctypes.c_void_p()
    capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        capi.check(l.mispmm_graph_launch(g, sp))
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        for _ in range(20):
            capi.check(l.mispmm_graph_launch(g, sp))
        ev1.record(stream)
    torch.cuda.synchronize()
    print(f"{capi.last_kernel()}: pitch {ev0.elapsed_time(ev1) * 1e3 / (20 * a.launches):.2f} us per launch (stamp build, graph of {a.launches})")
    st = buf.cpu().numpy().astype(np.int64)
    st = st[st[:, 0] != 0]
    us = lambda x: x * 0.01                           # noqa: E731  100 MHz ticks
    t0 = st[:, 0].min()
    print(f"{a.matrix} x K={a.k_cols}: {len(st)} waves, first start -> last end {us(st[:, 6].max() - t0):.2f} us")
    used = st[:, 2] != st[:, 1]
    rows = [("wave start", st[:, 0] - t0), ("column-list hop", st[:, 1] - st[:, 0]), ("B rows + tile (used slots)", (st[:, 2] - st[:, 1])[used]),
            ("MFMA + LDS write", st[:, 3] - st[:, 2]), ("barrier wait", st[:, 4] - st[:, 3]), ("reduce + store issue", st[:, 5] - st[:, 4]),
            ("store drain", st[:, 6] - st[:, 5]), ("barrier passed at", st[:, 4] - t0), ("wave end", st[:, 6] - t0), ("wave lifetime", st[:, 6] - st[:, 0])]
    print("  [us]                        p10    p50    p90    max")
    for name, v in rows:
        q = np.percentile(us(v.astype(np.float64)), [10, 50, 90, 100])
        print(f"  {name:<26} {q[0]:6.2f} {q[1]:6.2f} {q[2]:6.2f} {q[3]:6.2f}")
    hw, xcc = st[:, 7] & 0xFFFFFFFF, (st[:, 7] >> 32) & 0xF
    cu_key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    keys, inv = np.unique(cu_key, return_inverse=True)
    waves = np.bincount(inv)
    print(f"  placement: {len(keys)} CUs used; waves per CU histogram {dict(zip(*np.unique(waves, return_counts=True)))}")


if __name__ == "__main__":
    main()
