"""fp32-arithmetic rows (COO / ELL / BSR list) of a long-row matrix: the row-gather kernel it used to take, the split
kernel's shape in row order (mispmm_bsr_nonzeros_f32 = rows with boundaries), and longest first (mispmm_rows_split_f32)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, ops, synth  # noqa: E402
import bench  # noqa: E402


def main():
    l = capi.lib()
    stream = torch.cuda.Stream()
    timer = bench.Timer(stream)
    sp = ops._stream_ptr(stream)
    for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["GL7d25"]):
        csr = datasets.load_csr(name)
        a = ops.DeviceCSR.from_host(csr, spans=False)
        spans = torch.from_numpy(ops.csr_spans_by_length(csr.row_ptrs, 0xFFFFFFFF).reshape(-1).view(np.int32)).cuda()
        for n in (64, 128, 256):
            b = torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda()
            c = torch.empty((csr.num_rows, n), device="cuda")
            want = None
            for tag, call in (("row order", lambda: l.mispmm_bsr_nonzeros_f32(sp, csr.num_rows, csr.num_cols, csr.nnz, ops._p(a.row_ptrs), ops._p(a.col_idxs), ops._p(a.data), ops._p(b), n, n, ops._p(c), n, 0)),
                              ("longest first", lambda: l.mispmm_rows_split_f32(sp, csr.num_rows, csr.num_cols, csr.nnz, ops._p(a.col_idxs), ops._p(a.data), ops._p(spans), csr.num_rows, ops._p(b), n, n, ops._p(c), n, 0))):
                capi.check(call())
                st = timer.measure(call, 100, rounds=3, precondition_s=0.01)
                stream.synchronize()
                if want is None:
                    want = c.clone()
                print(json.dumps({"matrix": name, "n": n, "order": tag, "us": round(st["median_us"], 3), "same_bits": bool(torch.equal(c, want)),
                                  "tag": capi.last_kernel()}), flush=True)


if __name__ == "__main__":
    main()
