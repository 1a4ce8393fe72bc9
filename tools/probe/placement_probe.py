"""Does the headline's time depend on WHERE its operands sit?  One process, one kernel, several freshly allocated copies
of B and C (earlier copies are kept alive so every copy has other addresses), each timed through a 1000-launch graph.
  python tools/probe/placement_probe.py [--copies 8]
Prints us per SpMM with the device addresses (mod 2 MiB and mod 1 GiB) of B and C.  GPU box only."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, ops, synth  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--copies", type=int, default=8)
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--pad-mib", type=int, default=0, help="allocate this much between copies (changes the placement pattern)")
    p.add_argument("--vary", default="both", choices=["both", "b", "c"], help="which operand gets a fresh copy each time")
    a = p.parse_args()
    l = capi.lib()
    csr = datasets.load_csr("n4c6-b13")
    da = ops.DeviceCSR.from_host(csr)
    bh = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols))
    stream = torch.cuda.Stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    keep = []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(a.copies):
        if a.pad_mib:
            keep.append(torch.empty(a.pad_mib << 20, dtype=torch.uint8, device="cuda"))
        if i == 0 or a.vary in ("both", "b"):
            b = bh.cuda()
        if i == 0 or a.vary in ("both", "c"):
            c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
        keep += [b, c]
        ops.spmm_csr(da, b, out=c, stream=stream)
        torch.cuda.synchronize()
        capi.check(l.mispmm_graph_begin(sp))
        for _ in range(1000):
            ops.spmm_csr(da, b, out=c, stream=stream)
        g = ctypes.c_void_p()
        capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
        for _ in range(10):
            capi.check(l.mispmm_graph_launch(g, sp))
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            with torch.cuda.stream(stream):
                ev0.record(stream)
                for _ in range(8):
                    capi.check(l.mispmm_graph_launch(g, sp))
                ev1.record(stream)
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1) * 1e3 / 8000)
        capi.check(l.mispmm_graph_destroy(g))
        bp, cp = b.data_ptr(), c.data_ptr()
        print(f"copy {i}: {np.median(times):.3f} us (min {min(times):.3f} max {max(times):.3f})  B @ {bp:#x} (mod 2 MiB {bp % (2 << 20):#x}, mod 1 GiB {bp % (1 << 30):#x})  "
              f"C @ {cp:#x} (mod 2 MiB {cp % (2 << 20):#x})")


if __name__ == "__main__":
    main()
