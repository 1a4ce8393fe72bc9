"""Where does a wave of the headline launch spend its time?  Runs the PRODUCTION row-gather kernel from a diagnostic
build of the library (-DMISPMM_STAMPS: s_memrealtime stamps, 100 MHz, into a side buffer) and prints percentiles.

  make -C cuda-optimization-for-spmm_amd stamps        (builds libmispmm_stamps.so next to libmispmm.so)
  MISPMM_LIB=.../libmispmm_stamps.so python tools/stamp_headline.py [--acc reference|fast] [--k-cols 128]
GPU box only.  Stamps: 0 wave start, 1 (col, val) arrived, 2 last B row summed, 3 store issued, 4 store drained."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, ops, synth  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--matrix", default="n4c6-b13")
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--acc", default="reference")
    p.add_argument("--launches", type=int, default=50)
    p.add_argument("--graph", action="store_true", help="replay a graph of back-to-back launches (steady state)")
    a = p.parse_args()
    l = capi.lib()
    if not hasattr(l, "mispmm_debug_set_stamps"):
        raise SystemExit("this library was not built with -DMISPMM_STAMPS (set MISPMM_LIB)")
    l.mispmm_debug_set_stamps.argtypes = [ctypes.c_void_p]
    l.mispmm_debug_set_stamps.restype = ctypes.c_int
    csr = datasets.load_csr(a.matrix)
    da = ops.DeviceCSR.from_host(csr)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols)).cuda()
    c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
    slots = 1 << 16                                   # more waves than any launch here has
    buf = torch.zeros((slots, 8), dtype=torch.int64, device="cuda")
    capi.check(l.mispmm_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())))
    if a.graph:                                       # one hipGraph of `launches` kernels, replayed: the bench's steady state
        stream = torch.cuda.Stream()
        sp = ctypes.c_void_p(stream.cuda_stream)
        ops.spmm_csr(da, b, out=c, acc=a.acc, stream=stream)
        torch.cuda.synchronize()
        capi.check(l.mispmm_graph_begin(sp))
        for _ in range(a.launches):
            ops.spmm_csr(da, b, out=c, acc=a.acc, stream=stream)
        g = ctypes.c_void_p()
        capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
        for _ in range(3):
            capi.check(l.mispmm_graph_launch(g, sp))
        torch.cuda.synchronize()
    else:
        for _ in range(a.launches):                   # eager launches, microseconds apart; the last launch's stamps survive
            ops.spmm_csr(da, b, out=c, acc=a.acc)
        torch.cuda.synchronize()
    st = buf.cpu().numpy().astype(np.int64)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    us = lambda x: x * 0.01                           # noqa: E731  100 MHz ticks
    print(f"{a.matrix} x K={a.k_cols} acc={a.acc} {'graph replay' if a.graph else 'eager'}: {len(st)} waves, first start -> last end {us(st[:, 4].max() - t0):.2f} us")
    rows = [("wave start", st[:, 0] - t0), ("(col,val) hop", st[:, 1] - st[:, 0]), ("B gather + sums", st[:, 2] - st[:, 1]),
            ("store issue", st[:, 3] - st[:, 2]), ("store drain", st[:, 4] - st[:, 3]), ("wave end", st[:, 4] - t0),
            ("wave lifetime", st[:, 4] - st[:, 0])]
    print("  [us]              p10    p50    p90    max")
    for name, v in rows:
        q = np.percentile(us(v.astype(np.float64)), [10, 50, 90, 100])
        print(f"  {name:<16} {q[0]:6.2f} {q[1]:6.2f} {q[2]:6.2f} {q[3]:6.2f}")


if __name__ == "__main__":
    main()
