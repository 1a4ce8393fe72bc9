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


def greedy_cluster(csr, parts):
    """Greedy growth of `parts` balanced row clusters maximising shared columns (experiment aid)."""
    import heapq
    import scipy.sparse as sp
    m, k = csr.num_rows, csr.num_cols
    indptr, ind = csr.row_ptrs.astype(np.int64), csr.col_idxs.astype(np.int64)
    at = sp.csr_matrix((np.ones(csr.nnz), ind, indptr), shape=(m, k)).T.tocsr()
    cap = -(-m // parts)
    unassigned = np.ones(m, bool)
    order, nxt = [], 0
    for _ in range(parts):
        colin, gain, heap, size = np.zeros(k, bool), np.zeros(m, np.int64), [], 0
        while nxt < m and not unassigned[nxt]:
            nxt += 1
        r = nxt if nxt < m else None
        while size < cap and r is not None:
            order.append(r); unassigned[r] = False; size += 1
            for c in ind[indptr[r]:indptr[r + 1]]:
                if not colin[c]:
                    colin[c] = True
                    for r2 in at.indices[at.indptr[c]:at.indptr[c + 1]]:
                        if unassigned[r2]:
                            gain[r2] += 1
                            heapq.heappush(heap, (-gain[r2], r2))
            r = None
            while heap:
                g, r2 = heapq.heappop(heap)
                if unassigned[r2] and -g == gain[r2]:
                    r = r2
                    break
            if r is None:
                while nxt < m and not unassigned[nxt]:
                    nxt += 1
                r = nxt if nxt < m else None
    return np.array(order)


def permute_rows(csr, order):
    from mispmm import formats
    lens = np.diff(csr.row_ptrs.astype(np.int64))[order]
    ptr = np.concatenate([[0], np.cumsum(lens)])
    idx = np.concatenate([np.arange(csr.row_ptrs[r], csr.row_ptrs[r + 1]) for r in order]).astype(np.int64)
    return formats.CSR(csr.num_rows, csr.num_cols, ptr.astype(np.uint32), csr.col_idxs[idx], csr.data[idx])


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--matrix", default="n4c6-b13")
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--iters", type=int, default=500)
    p.add_argument("--rounds", type=int, default=5)
    p.add_argument("--kernels", default="1,2,3,4")
    p.add_argument("--cluster", type=int, default=0, help="permute rows by greedy clustering into this many parts")
    args = p.parse_args()
    l = capi.lib()
    csr = datasets.load_csr(args.matrix)
    if args.cluster:
        csr = permute_rows(csr, greedy_cluster(csr, args.cluster))
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
