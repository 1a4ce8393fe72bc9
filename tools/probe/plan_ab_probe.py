"""Storage order vs clustered plan order, SAME process and SAME B / C buffers (so operand placement cancels out), alternating
1000-launch graphs.  python tools/probe/plan_ab_probe.py [--k-cols 128] [--parts 4,8,16,64]   GPU box only."""
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
    p.add_argument("--parts", default="4,8,16,64,256")
    p.add_argument("--acc", default="reference")
    a = p.parse_args()
    l = capi.lib()
    csr = datasets.load_csr(a.matrix)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols)).cuda()
    c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
    stream = torch.cuda.Stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    plain = ops.DeviceCSR.from_host(csr, plan=False)
    variants = [("storage order", lambda: ops.spmm_csr(plain, b, out=c, acc=a.acc, stream=stream))]
    for parts in [int(x) for x in a.parts.split(",")]:
        ops.PLAN_PARTS = parts
        d = ops.DeviceCSR.from_host(csr, plan=True)
        variants.append((f"plan order, {parts} clusters ({d.plan.natural_distinct} -> {d.plan.clustered_distinct} distinct columns)",
                         (lambda dd: (lambda: ops._csr_plan(dd, [b], [c], a.acc, stream)))(d)))
    graphs = []
    for name, fn in variants:
        fn()
        torch.cuda.synchronize()
        capi.check(l.mispmm_graph_begin(sp))
        for _ in range(1000):
            fn()
        g = ctypes.c_void_p()
        capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
        graphs.append(g)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    times = {name: [] for name, _ in variants}
    for g in graphs:
        for _ in range(5):
            capi.check(l.mispmm_graph_launch(g, sp))
    torch.cuda.synchronize()
    for _ in range(6):
        for (name, _), g in zip(variants, graphs):
            with torch.cuda.stream(stream):
                ev0.record(stream)
                for _ in range(6):
                    capi.check(l.mispmm_graph_launch(g, sp))
                ev1.record(stream)
            torch.cuda.synchronize()
            times[name].append(ev0.elapsed_time(ev1) * 1e3 / 6000)
    base = np.median(times["storage order"])
    print(f"# {a.matrix} x K={a.k_cols} {a.acc}, one process, one B / C, rounds interleaved")
    for name, _ in variants:
        t = np.array(times[name])
        print(f"{name:75s} {np.median(t):.3f} us (min {t.min():.3f} max {t.max():.3f})  {100 * (np.median(t) / base - 1):+.1f} %")


if __name__ == "__main__":
    main()
