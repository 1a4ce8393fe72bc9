"""Where is the headline kernel's time spent: beyond the L2 or in the L1/L2 request path?
Same kernel, same instruction stream, same bytes through the vector L1s -- but the column indices of A are
folded into a window of `fold` B rows, so every B segment is L2 (fold small) or Infinity-Cache (fold large)
resident.  GPU box only.  Prints one JSON line per variant."""
import argparse
import ctypes
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


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--k-cols", type=int, default=128)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--folds", default="0,512,2048,8192")
    a = p.parse_args()
    capi.lib()
    csr = datasets.load_csr("n4c6-b13")
    n = a.k_cols
    stream = torch.cuda.Stream()
    timer = bench.Timer(stream)
    b = torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda()
    c = torch.empty((csr.num_rows, n), device="cuda")
    rng = np.random.default_rng(1)
    for fold in [int(x) for x in a.folds.split(",")]:
        for acc in ("reference", "fast"):
            if fold == 0:
                cols, tag = csr.col_idxs, "real columns"
            elif fold > 0:
                cols, tag = (csr.col_idxs % fold).astype(np.uint32), f"columns mod {fold} ({fold * n * 4 // 1024} KiB of B)"
            m = formats.CSR(csr.num_rows, csr.num_cols, csr.row_ptrs, cols, csr.data)
            da = ops.DeviceCSR.from_host(m)
            st = timer.measure(lambda: ops.spmm_csr(da, b, out=c, acc=acc, stream=stream), a.steps, rounds=3)
            print(json.dumps({"variant": tag, "acc": acc, "median_us": round(st["median_us"], 3), "min_us": round(st["min_us"], 3),
                              "kernel": capi.last_kernel()}), flush=True)
    # random columns over the whole of B: no L2 reuse at all
    cols = rng.integers(0, csr.num_cols, csr.nnz).astype(np.uint32)
    m = formats.CSR(csr.num_rows, csr.num_cols, csr.row_ptrs, cols, csr.data)
    da = ops.DeviceCSR.from_host(m)
    st = timer.measure(lambda: ops.spmm_csr(da, b, out=c, acc="reference", stream=stream), a.steps, rounds=3)
    print(json.dumps({"variant": "uniformly random columns", "acc": "reference", "median_us": round(st["median_us"], 3)}), flush=True)


if __name__ == "__main__":
    main()
