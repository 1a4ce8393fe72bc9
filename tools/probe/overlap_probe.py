"""How much of a headline step is the DEPENDENCY between two launches?  The same kernel as (a) one chain of dependent
launches (what bench.py times: `value`), (b) 2 / 4 independent chains in one hipGraph, each writing its own C, so that
launches of different chains may overlap.  (b) is an anatomy of the launch boundary, NOT a figure of merit: a stream of
SpMMs is ordered.   python tools/probe/overlap_probe.py [--k-cols 128]     GPU box only."""
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
    p.add_argument("--matrix", default="n4c6-b13")
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--per-chain", type=int, default=250)
    a = p.parse_args()
    l = capi.lib()
    csr = datasets.load_csr(a.matrix)
    da = ops.DeviceCSR.from_host(csr, plan=False)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols)).cuda()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    print(f"# {a.matrix} x K={a.k_cols} REFERENCE, graphs of {a.per_chain} launches per chain, us per SpMM")
    for chains in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(chains)]
        cs = [torch.empty((csr.num_rows, a.k_cols), device="cuda") for _ in range(chains)]
        for s, c in zip(streams, cs):
            ops.spmm_csr(da, b, out=c, stream=s)
        torch.cuda.synchronize()
        main_s = streams[0]
        sp = ctypes.c_void_p(main_s.cuda_stream)
        capi.check(l.mispmm_graph_begin(sp))
        fork = torch.cuda.Event()
        fork.record(main_s)
        for s in streams[1:]:
            s.wait_event(fork)
        for _ in range(a.per_chain):
            for s, c in zip(streams, cs):
                ops.spmm_csr(da, b, out=c, stream=s)
        for s in streams[1:]:
            j = torch.cuda.Event()
            j.record(s)
            main_s.wait_event(j)
        g = ctypes.c_void_p()
        capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
        for _ in range(10):
            capi.check(l.mispmm_graph_launch(g, sp))
        torch.cuda.synchronize()
        times = []
        for _ in range(5):
            with torch.cuda.stream(main_s):
                ev0.record(main_s)
                for _ in range(20):
                    capi.check(l.mispmm_graph_launch(g, sp))
                ev1.record(main_s)
            torch.cuda.synchronize()
            times.append(ev0.elapsed_time(ev1) * 1e3 / (20 * a.per_chain * chains))
        capi.check(l.mispmm_graph_destroy(g))
        print(f"{chains} chain(s) of dependent launches: {np.median(times):.3f} us per SpMM (min {min(times):.3f} max {max(times):.3f})"
              + ("   <- bench.py's `value`" if chains == 1 else "   (launches of different chains overlap: not a figure of merit)"))


if __name__ == "__main__":
    main()
