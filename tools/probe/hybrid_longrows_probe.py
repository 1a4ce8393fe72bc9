"""Would a long-row matrix be served better by two bodies in one launch -- the row-gather body for its short rows, the
split body only for the long ones?  Feasibility numbers before building anything: the rows of GL7d25 up to T entries as
a matrix of their own through the row-gather kernel (tuning build, MISPMM_SPLIT=0), the rows above T as a matrix of
their own through the split kernel (span list), and the whole matrix as it runs today.  GPU box only.
  MISPMM_SPLIT=0 MISPMM_LIB=cuda-optimization-for-spmm_amd/libmispmm_tune.so python tools/probe/hybrid_longrows_probe.py"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, formats, ops, synth  # noqa: E402


def sub_rows(csr, keep):
    """the rows listed in `keep` as a matrix of their own (same columns)"""
    rp = csr.row_ptrs.astype(np.int64)
    lens = (rp[1:] - rp[:-1])[keep]
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in keep]) if len(keep) else np.zeros(0, np.int64)
    ptrs = np.zeros(len(keep) + 1, np.uint32)
    ptrs[1:] = np.cumsum(lens)
    return formats.CSR(len(keep), csr.num_cols, ptrs, csr.col_idxs[idx].astype(np.uint32), csr.data[idx].astype(np.float32))


def time_graph(fn, stream, launches=500, rounds=5):
    l = capi.lib()
    sp = ctypes.c_void_p(stream.cuda_stream)
    capi.check(l.mispmm_graph_begin(sp))
    for _ in range(launches):
        fn()
    g = ctypes.c_void_p()
    capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
    for _ in range(3):
        capi.check(l.mispmm_graph_launch(g, sp))
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = []
    for _ in range(rounds):
        with torch.cuda.stream(stream):
            ev0.record(stream)
            for _ in range(4):
                capi.check(l.mispmm_graph_launch(g, sp))
            ev1.record(stream)
        torch.cuda.synchronize()
        out.append(ev0.elapsed_time(ev1) * 1e3 / (4 * launches))
    return float(np.median(out))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--matrix", default="GL7d25")
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--acc", default="reference")
    p.add_argument("--b-mode", default="uniform", help="--ab: synth.dense_b mode: uniform, or exact (a grid of values on which no wave's exactness test fails)")
    p.add_argument("--threshold", type=int, default=0, help="--ab: rows of more entries than this go to the split body (0 = the library's 32)")
    p.add_argument("--force", action="store_true", help="only: the matrix's own kernel (no span list) against the two-body launch over a span list built anyway")
    p.add_argument("--ab", action="store_true", help="only: the two-body launch (kernel 0) against the split kernel on the whole list (kernel 6), rounds interleaved")
    a = p.parse_args()
    if a.force:
        # a matrix that gets no span list by itself: its own kernel against the two-body launch over a list built anyway
        csr = datasets.load_csr(a.matrix)
        lens = np.diff(csr.row_ptrs.astype(np.int64))
        b = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols)).cuda()
        c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
        stream = torch.cuda.Stream()
        plain, listed = ops.DeviceCSR.from_host(csr, spans=False), ops.DeviceCSR.from_host(csr, spans=True)
        res, tags = {"own kernel": [], "two-body launch": []}, {}
        for _ in range(3):
            for name, dev_a in (("own kernel", plain), ("two-body launch", listed)):
                res[name].append(time_graph(lambda: ops.spmm_csr(dev_a, b, out=c, acc=a.acc, stream=stream), stream, rounds=3))
                tags[name] = capi.last_kernel()
        base = np.median(res["own kernel"])
        print(f"# {a.matrix} x K={a.k_cols} {a.acc}: mean row {lens.mean():.1f}, longest {lens.max()}, rows of more than 32 entries {(lens > 32).sum()}")
        for name, t in res.items():
            print(f"{name:18s} {np.median(t):7.3f} us (min {min(t):.3f} max {max(t):.3f})  {100 * (np.median(t) / base - 1):+6.1f} %   {tags[name]}")
        return
    if a.ab:
        csr = datasets.load_csr(a.matrix)
        dev_a = ops.DeviceCSR.from_host(csr)
        if a.threshold:
            dev_a.long_spans = ops.spans_long_count(ops.csr_spans_by_length(csr.row_ptrs), a.threshold)
        b = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols, mode=a.b_mode)).cuda()
        c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
        stream = torch.cuda.Stream()
        res = {"two-body launch": [], "split kernel": []}
        tags = {}
        for _ in range(3):
            for name, kernel in (("two-body launch", 0), ("split kernel", 6)):
                res[name].append(time_graph(lambda: ops.spmm_csr(dev_a, b, out=c, kernel=kernel, acc=a.acc, stream=stream), stream, rounds=3))
                tags[name] = capi.last_kernel()
        base = np.median(res["split kernel"])
        print(f"# {a.matrix} x K={a.k_cols} {a.acc} B {a.b_mode} threshold {a.threshold or ops.HYBRID_ROW_LEN}: one process, one set of operands, rounds interleaved")
        for name, t in res.items():
            print(f"{name:18s} {np.median(t):7.3f} us (min {min(t):.3f} max {max(t):.3f})  {100 * (np.median(t) / base - 1):+6.1f} %   {tags[name]}")
        return
    csr = datasets.load_csr(a.matrix)
    lens = np.diff(csr.row_ptrs.astype(np.int64))
    b = torch.from_numpy(synth.dense_b(csr.num_cols, a.k_cols)).cuda()
    stream = torch.cuda.Stream()
    whole = ops.DeviceCSR.from_host(csr)
    c = torch.empty((csr.num_rows, a.k_cols), device="cuda")
    t = time_graph(lambda: ops.spmm_csr(whole, b, out=c, acc=a.acc, stream=stream), stream)
    print(f"# {a.matrix} x K={a.k_cols} {a.acc}: {csr.num_rows} rows, nnz {csr.nnz}, longest {lens.max()}")
    print(f"whole matrix as today                         {t:7.3f} us   {capi.last_kernel()}")
    gen = ops.DeviceCSR.from_host(csr, spans=False, plan=False)
    t = time_graph(lambda: ops.spmm_csr(gen, b, out=c, acc=a.acc, stream=stream, use_hint=False), stream)
    print(f"whole matrix, row-gather kernel               {t:7.3f} us   {capi.last_kernel()}")
    order = np.argsort(-lens, kind="stable")
    for T in (24, 32, 48):
        short = order[lens[order] <= T]
        long_ = order[lens[order] > T]
        for name, rows, spans in (("short rows, storage order", np.sort(short), False), ("short rows, longest first", short, False),
                                  ("long rows, split kernel", long_, True)):
            if len(rows) == 0:
                continue
            sub = sub_rows(csr, rows)
            dev = ops.DeviceCSR.from_host(sub, spans=spans, plan=False)
            cs = torch.empty((sub.num_rows, a.k_cols), device="cuda")
            if spans:
                t = time_graph(lambda: ops.spmm_csr(dev, b, out=cs, kernel=6, acc=a.acc, stream=stream), stream)
            else:
                t = time_graph(lambda: ops.spmm_csr(dev, b, out=cs, acc=a.acc, stream=stream, use_hint=False), stream)
            print(f"T={T:3d} {name:28s} rows {sub.num_rows:5d} nnz {sub.nnz:6d}  {t:7.3f} us   {capi.last_kernel()}")
        # both in ONE step: the long rows on the captured stream, the short rows on a second stream forked from it and
        # joined again (two parallel kernel nodes per step in the graph)
        if len(long_) and len(short):
            subs, subl = sub_rows(csr, np.sort(short)), sub_rows(csr, long_)
            ds, dl = ops.DeviceCSR.from_host(subs, spans=False, plan=False), ops.DeviceCSR.from_host(subl, spans=True, plan=False)
            c1, c2 = torch.empty((subs.num_rows, a.k_cols), device="cuda"), torch.empty((subl.num_rows, a.k_cols), device="cuda")
            side = torch.cuda.Stream()
            fork, join = torch.cuda.Event(), torch.cuda.Event()

            def both():
                fork.record(stream)
                side.wait_event(fork)
                ops.spmm_csr(dl, b, out=c2, kernel=6, acc=a.acc, stream=stream)
                ops.spmm_csr(ds, b, out=c1, acc=a.acc, stream=side, use_hint=False)
                join.record(side)
                stream.wait_event(join)
            t = time_graph(both, stream)
            print(f"T={T:3d} {'both, forked streams':28s} {'':24s}{t:7.3f} us")

            def serial():
                ops.spmm_csr(dl, b, out=c2, kernel=6, acc=a.acc, stream=stream)
                ops.spmm_csr(ds, b, out=c1, acc=a.acc, stream=stream, use_hint=False)
            t = time_graph(serial, stream)
            print(f"T={T:3d} {'both, one after the other':28s} {'':24s}{t:7.3f} us")


if __name__ == "__main__":
    main()
