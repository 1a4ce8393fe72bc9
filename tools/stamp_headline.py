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
    p.add_argument("--sets", type=int, default=1, help="--graph: rotate over this many distinct (B, C) pairs (0 = 512 MiB of them): the "
                                                       "HBM-streamed form of the loop (bench.py `hbm_streaming`)")
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
    KEEP = 32                                         # kStampLaunches in row_gather.hpp: records of the last 32 launches
    slots = KEEP * (1 << 14)                          # more waves than any launch here has, per kept launch
    buf = torch.zeros((slots, 8), dtype=torch.int64, device="cuda")
    capi.check(l.mispmm_debug_set_stamps(ctypes.c_void_p(buf.data_ptr())))
    if a.graph:                                       # one hipGraph of `launches` kernels, replayed: the bench's steady state
        stream = torch.cuda.Stream()
        sp = ctypes.c_void_p(stream.cuda_stream)
        ops.spmm_csr(da, b, out=c, acc=a.acc, stream=stream)
        torch.cuda.synchronize()
        nsets = a.sets if a.sets > 0 else max(4, -(-(512 << 20) // ((csr.num_cols + csr.num_rows) * a.k_cols * 4)))
        pairs = [(b, c)] + [(b.clone(), torch.empty_like(c)) for _ in range(nsets - 1)]
        torch.cuda.synchronize()
        capi.check(l.mispmm_graph_begin(sp))
        for i in range(-(-a.launches // nsets) * nsets):
            ops.spmm_csr(da, pairs[i % nsets][0], out=pairs[i % nsets][1], acc=a.acc, stream=stream)
        g = ctypes.c_void_p()
        capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
        for _ in range(3):
            capi.check(l.mispmm_graph_launch(g, sp))
        torch.cuda.synchronize()
    else:
        for _ in range(a.launches):                   # eager launches, microseconds apart; the last launch's stamps survive
            ops.spmm_csr(da, b, out=c, acc=a.acc)
        torch.cuda.synchronize()
    allst = buf.cpu().numpy().astype(np.int64)
    allst = allst[allst[:, 0] != 0]
    us = lambda x: x * 0.01                           # noqa: E731  100 MHz ticks
    # the kept launches are contiguous record ranges of equal size; order them by their first start
    per = len(allst) // KEEP if len(allst) % KEEP == 0 else None
    if per:
        groups = sorted((allst[i * per:(i + 1) * per] for i in range(KEEP)), key=lambda g: g[:, 0].min())
        starts = np.array([g[:, 0].min() for g in groups])
        ends = np.array([g[:, 4].max() for g in groups])
        pitch, span, gap = us(np.diff(starts)), us(ends - starts), us(starts[1:] - ends[:-1])
        ok = pitch < 50                               # consecutive launches of one graph replay (not across replays)
        print(f"consecutive launches ({ok.sum()} pairs): pitch median {np.median(pitch[ok]):.2f} us | in-kernel span "
              f"(first wave start -> last wave end) median {np.median(span):.2f} us | idle gap between launches "
              f"(last end -> next first start) median {np.median(gap[ok]):.2f} us, min {gap[ok].min():.2f}, max {gap[ok].max():.2f}")
        st = groups[len(groups) // 2]
    else:
        st = allst
    t0 = st[:, 0].min()
    print(f"{a.matrix} x K={a.k_cols} acc={a.acc} {'graph replay' if a.graph else 'eager'}: {len(st)} waves, first start -> last end {us(st[:, 4].max() - t0):.2f} us")
    rows = [("wave start", st[:, 0] - t0), ("(col,val) hop", st[:, 1] - st[:, 0]), ("B gather + sums", st[:, 2] - st[:, 1]),
            ("store issue", st[:, 3] - st[:, 2]), ("store drain", st[:, 4] - st[:, 3]), ("wave end", st[:, 4] - t0),
            ("wave lifetime", st[:, 4] - st[:, 0])]
    print("  [us]              p10    p50    p90    max")
    for name, v in rows:
        q = np.percentile(us(v.astype(np.float64)), [10, 50, 90, 100])
        print(f"  {name:<16} {q[0]:6.2f} {q[1]:6.2f} {q[2]:6.2f} {q[3]:6.2f}")
    placement(st, us)


def placement(st, us):
    """Per-CU load: how many waves each CU got and when its last wave ended (HW_ID / XCC_ID stamps)."""
    hw, xcc = st[:, 5], st[:, 6] & 0xF
    cu_key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)
    t0 = st[:, 0].min()
    keys, inv = np.unique(cu_key, return_inverse=True)
    waves = np.bincount(inv)
    last_end = np.zeros(len(keys))
    first_start = np.full(len(keys), 1e18)
    for i in range(len(keys)):
        sel = inv == i
        last_end[i] = us(st[sel, 4].max() - t0)
        first_start[i] = us(st[sel, 0].min() - t0)
    print(f"  placement: {len(keys)} CUs used; waves per CU min/median/max = {waves.min()}/{int(np.median(waves))}/{waves.max()}; "
          f"histogram {dict(zip(*np.unique(waves, return_counts=True)))}")
    for w in np.unique(waves):
        sel = waves == w
        print(f"    CUs with {w:2d} waves: {sel.sum():3d}   last wave end p50 {np.median(last_end[sel]):.2f} us  max {last_end[sel].max():.2f} us")
    per_xcc = {int(x): int((xcc == x).sum()) for x in np.unique(xcc)}
    print(f"  waves per XCC: {per_xcc}")
    simd = (hw >> 4) & 3
    print(f"  waves per SIMD id: {dict(zip(*np.unique(simd, return_counts=True)))}")


if __name__ == "__main__":
    main()
