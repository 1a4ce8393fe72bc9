"""Time every CSR kernel variant on one workload in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  GPU box only.

  python tools/kernel_sweep.py [--matrix n4c6-b13] [--k-cols 128] [--iters 500] [--rounds 5]
Prints per (kernel, acc, launch mode): median / min microseconds per SpMM and the HBM roofline fraction.
"""
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
    p.add_argument("--iters", type=int, default=500)
    p.add_argument("--rounds", type=int, default=5)
    p.add_argument("--kernels", default="1,2,3,4")
    args = p.parse_args()
    l = capi.lib()
    csr = datasets.load_csr(args.matrix)
    a = ops.DeviceCSR.from_host(csr)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, args.k_cols)).cuda()
    c = torch.empty((csr.num_rows, args.k_cols), device="cuda")
    stream = torch.cuda.Stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    ev0, ev1 = ctypes.c_void_p(), ctypes.c_void_p()
    capi.check(l.mispmm_event_create(ctypes.byref(ev0)))
    capi.check(l.mispmm_event_create(ctypes.byref(ev1)))
    abytes = datasets.csr_algorithmic_bytes(csr, args.k_cols)
    variants = [(k, acc) for k in map(int, args.kernels.split(",")) for acc in ("reference", "fast")]
    graphs = {}
    for k, acc in variants:
        capi.check(l.mispmm_graph_begin(sp))
        for _ in range(args.iters):
            ops.spmm_csr(a, b, out=c, kernel=k, acc=acc, stream=stream)
        g = ctypes.c_void_p()
        capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
        graphs[(k, acc)] = g
        capi.check(l.mispmm_graph_launch(g, sp))
    torch.cuda.synchronize()
    times = {v: {"graph": [], "eager": []} for v in variants}
    ms = ctypes.c_float()
    for _ in range(args.rounds):
        for v in variants:
            capi.check(l.mispmm_event_record(ev0, sp))
            capi.check(l.mispmm_graph_launch(graphs[v], sp))
            capi.check(l.mispmm_event_record(ev1, sp))
            capi.check(l.mispmm_event_sync(ev1))
            capi.check(l.mispmm_event_elapsed_ms(ev0, ev1, ctypes.byref(ms)))
            times[v]["graph"].append(ms.value * 1e3 / args.iters)
            capi.check(l.mispmm_event_record(ev0, sp))
            for _ in range(args.iters):
                ops.spmm_csr(a, b, out=c, kernel=v[0], acc=v[1], stream=stream)
            capi.check(l.mispmm_event_record(ev1, sp))
            capi.check(l.mispmm_event_sync(ev1))
            capi.check(l.mispmm_event_elapsed_ms(ev0, ev1, ctypes.byref(ms)))
            times[v]["eager"].append(ms.value * 1e3 / args.iters)
    print(f"# {args.matrix} x K={args.k_cols}: algorithmic bytes {abytes}, 8 TB/s floor {abytes / 8e12 * 1e6:.2f} us")
    for v in variants:
        for mode in ("graph", "eager"):
            t = np.array(times[v][mode])
            print(json.dumps({"kernel": v[0], "acc": v[1], "launch": mode, "us_median": round(float(np.median(t)), 3),
                              "us_min": round(float(t.min()), 3),
                              "hbm_frac_at_median": round(abytes / (np.median(t) * 1e-6) / 8e12, 4)}))


if __name__ == "__main__":
    main()
