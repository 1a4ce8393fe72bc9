"""Where do the waves of the split body (csr_split.hpp; the long rows of the two-body launch) spend their time?  Diagnostic
build of the library (-DMISPMM_STAMPS: s_memrealtime stamps, 100 MHz, into a side buffer), the waves of the LAST launch of a
replayed graph.

  make -C cuda-optimization-for-spmm_amd stamps
  MISPMM_LIB=.../libmispmm_stamps.so python tools/stamp_split.py [--matrix GL7d25] [--k-cols 128] [--acc reference] [--kernel 0|6]
GPU box only.  Stamps: 0 start, 1 span read, 2 split pass done, 3 partial sums reduced, 4 hand-over barrier passed (waves of a
4-chunk row), 5 before the store (wave 0 of such a row: chunks added and tested; any wave: after an ordered pass), 6 end."""
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
    p.add_argument("--matrix", default="GL7d25")
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--acc", default="reference")
    p.add_argument("--kernel", type=int, default=0, help="0: the two-body launch (stamps of its split body); 6: the split kernel on every row")
    p.add_argument("--b-mode", default="uniform")
    p.add_argument("--launches", type=int, default=50)
    a = p.parse_args()
    l = capi.lib()
    if not hasattr(l, "mispmm_debug_set_stamps_split"):
        raise SystemExit("this library was not built with -DMISPMM_STAMPS (set MISPMM_LIB)")
    l.mispmm_debug_set_stamps_split.argtypes = [ctypes.c_void_p]
    l.mispmm_debug_set_stamps_split.restype = ctypes.c_int
    csr = datasets.load_csr(a.matrix)
    dev_a = ops.DeviceCSR.from_host(csr)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols, mode=a.b_mode)).cuda()
    c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
    buf = torch.zeros((1 << 17, 8), dtype=torch.int64, device="cuda")
    capi.check(l.mispmm_debug_set_stamps_split(ctypes.c_void_p(buf.data_ptr())))
    stream = torch.cuda.Stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    run = lambda: ops.spmm_csr(dev_a, b, out=c, kernel=a.kernel, acc=a.acc, stream=stream)  # noqa: E731
    run()
    torch.cuda.synchronize()
    capi.check(l.mispmm_graph_begin(sp))
    for _ in range(a.launches):
        run()
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
    print(f"{capi.last_kernel()}\n  {a.matrix} x K={a.k_cols} {a.acc}, B {a.b_mode}: pitch {ev0.elapsed_time(ev1) * 1e3 / (20 * a.launches):.2f} us per launch (stamp build)")
    st = buf.cpu().numpy().astype(np.int64)
    st = st[st[:, 0] != 0]
    if len(st) == 0:
        raise SystemExit("  no wave left stamps (did this launch run the split body?)")
    us = lambda x: np.asarray(x, dtype=np.float64) * 0.01        # noqa: E731  100 MHz ticks
    t0 = st[:, 0].min()
    length, shared, ordered = st[:, 7] & 0xFFFFFFFF, (st[:, 7] >> 32) & 1, (st[:, 7] >> 33) & 1
    print(f"  {len(st)} waves of the split body, first start -> last end {us(st[:, 6].max() - t0):.2f} us; {int(shared.sum())} waves hold a chunk of a "
          f"4-chunk row, {int(ordered.sum())} took the ordered pass")
    cols = [("start", st[:, 0] - t0), ("span read", st[:, 1] - st[:, 0]), ("split pass", st[:, 2] - st[:, 1]), ("reduce", st[:, 3] - st[:, 2]),
            ("barrier", st[:, 4] - st[:, 3]), ("chunks/ordered", st[:, 5] - st[:, 4]), ("store", st[:, 6] - st[:, 5]), ("end", st[:, 6] - t0)]
    for label, sel in (("waves of 4-chunk rows", shared == 1), ("other waves", shared == 0)):
        if not sel.any():
            continue
        print(f"  {label} ({int(sel.sum())}), us:      p10    p50    p90    max")
        for name, v in cols:
            q = np.percentile(us(v[sel]), [10, 50, 90, 100])
            print(f"    {name:<16} {q[0]:6.2f} {q[1]:6.2f} {q[2]:6.2f} {q[3]:6.2f}")
    print("  the 8 waves that end last:  entries  chunk  ordered |  start  span  pass  reduce barrier chunks store |  end")
    for i in np.argsort(-st[:, 6])[:8]:
        v = [us(c_[i]) for _, c_ in cols]
        print(f"    {int(length[i]):22d} {int(shared[i]):6d} {int(ordered[i]):8d} | " + " ".join(f"{x:6.2f}" for x in v[:7]) + f" | {v[7]:5.2f}")


if __name__ == "__main__":
    main()
