"""Does the ORDER in which a long-row matrix's rows are dispatched matter?  GL7d25 (rows of 2..422 entries) through the
default CSR path in natural order, longest-first and shortest-first (rows physically permuted: timing only)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, formats, ops, synth  # noqa: E402
import bench  # noqa: E402


def permute(csr, order):
    lens = np.diff(csr.row_ptrs.astype(np.int64))[order]
    ptr = np.concatenate([[0], np.cumsum(lens)])
    idx = np.concatenate([np.arange(csr.row_ptrs[r], csr.row_ptrs[r + 1]) for r in order]).astype(np.int64)
    return formats.CSR(csr.num_rows, csr.num_cols, ptr.astype(np.uint32), csr.col_idxs[idx], csr.data[idx])


def main():
    capi.lib()
    stream = torch.cuda.Stream()
    timer = bench.Timer(stream)
    for name in ("GL7d25", "g7jac010", "tols4000", "ACTIVSg10K"):
        csr = datasets.load_csr(name)
        n = 128
        b = torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda()
        c = torch.empty((csr.num_rows, n), device="cuda")
        lens = np.diff(csr.row_ptrs.astype(np.int64))
        for tag, order in (("natural", np.arange(csr.num_rows)), ("longest first", np.argsort(-lens, kind="stable")),
                           ("shortest first", np.argsort(lens, kind="stable"))):
            a = ops.DeviceCSR.from_host(permute(csr, order))
            for acc in ("reference", "fast"):
                st = timer.measure(lambda: ops.spmm_csr(a, b, out=c, acc=acc, stream=stream), 100, rounds=3, precondition_s=0.01)
                print(json.dumps({"matrix": name, "order": tag, "acc": acc, "us": round(st["median_us"], 3), "kernel": capi.last_kernel()}), flush=True)


if __name__ == "__main__":
    main()
