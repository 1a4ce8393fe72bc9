"""Throughput of the headline SpMM when S independent steps are in flight on S streams
(each with its own C), captured as one fork/join hipGraph.  GPU box only."""
import argparse
import ctypes
import json
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
    p.add_argument("--iters", type=int, default=480)
    p.add_argument("--kernel", type=int, default=5)
    p.add_argument("--acc", default="reference")
    args = p.parse_args()
    l = capi.lib()
    csr = datasets.load_csr(args.matrix)
    a = ops.DeviceCSR.from_host(csr)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, args.k_cols)).cuda()
    abytes = datasets.csr_algorithmic_bytes(csr, args.k_cols)
    for nstreams in (1, 2, 3, 4, 6, 8):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        cs = [torch.empty((csr.num_rows, args.k_cols), device="cuda") for _ in range(nstreams)]
        main_s = streams[0]
        sp = ctypes.c_void_p(main_s.cuda_stream)
        torch.cuda.synchronize()
        capi.check(l.mispmm_graph_begin(sp))
        fork = torch.cuda.Event()
        fork.record(main_s)
        for s in streams[1:]:
            s.wait_event(fork)
        per = args.iters // nstreams
        for i in range(per):
            for s, c in zip(streams, cs):
                ops.spmm_csr(a, b, out=c, kernel=args.kernel, acc=args.acc, stream=s)
        for s in streams[1:]:
            e = torch.cuda.Event()
            e.record(s)
            main_s.wait_event(e)
        g = ctypes.c_void_p()
        capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
        capi.check(l.mispmm_graph_launch(g, sp))
        torch.cuda.synchronize()
        ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
        capi.check(l.mispmm_event_create(ctypes.byref(ev0)))
        capi.check(l.mispmm_event_create(ctypes.byref(ev1)))
        ts = []
        ms = ctypes.c_float()
        for _ in range(5):
            capi.check(l.mispmm_event_record(ev0, sp))
            capi.check(l.mispmm_graph_launch(g, sp))
            capi.check(l.mispmm_event_record(ev1, sp))
            capi.check(l.mispmm_event_sync(ev1))
            capi.check(l.mispmm_event_elapsed_ms(ev0, ev1, ctypes.byref(ms)))
            ts.append(ms.value * 1e3 / (per * nstreams))
        t = float(np.median(ts))
        print(json.dumps({"streams": nstreams, "us_per_spmm": round(t, 3), "GFLOPs": round(2 * csr.nnz * args.k_cols / t / 1e3, 1),
                          "algorithmic_GBps": round(abytes / t / 1e3, 1)}))
        capi.check(l.mispmm_graph_destroy(g))


if __name__ == "__main__":
    main()
