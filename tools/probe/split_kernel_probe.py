"""Kernel 6 (csr_split: one workgroup per row, the row dealt over its lane groups) against kernel 5 (lane-group row
gather, or the deep wave-per-row kernel with MISPMM_SPLIT=0 in the tuning build) on the matrices with long or uneven
rows, several dense widths; checks REFERENCE-mode bits against kernel 1 while it is at it.  Kernel 6 runs twice: in row
order (mispmm_csr_f32) and with the rows longest first (mispmm_csr_split_f32).
  usage: split_kernel_probe.py [matrix,matrix,..] [n,n,..]
  PROBE_B_MODE = uniform (default) | exact (2^-8 grid) | clamped (no product small enough to fail the re-association
                 test) | planted (exactly one element fails it) | wide (every wave fails it)
  PROBE_CUT_ROWS = L: every row longer than L becomes several rows of at most L entries (timing experiment)"""
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
    capi.lib()
    stream = torch.cuda.Stream()
    timer = bench.Timer(stream)
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["GL7d25", "g7jac010", "tols4000", "ACTIVSg10K", "n4c6-b13"]
    widths = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 128, 256, 512]
    for name in names:
        csr = datasets.load_csr(name)
        cut = int(os.environ.get("PROBE_CUT_ROWS", "0"))
        if cut:  # timing experiment: every row longer than `cut` becomes several rows of at most `cut` entries
            from mispmm import formats
            ptr = [0]
            for r in range(csr.num_rows):
                s0, e0 = int(csr.row_ptrs[r]), int(csr.row_ptrs[r + 1])
                while e0 - s0 > cut:
                    s0 += cut
                    ptr.append(s0)
                ptr.append(e0)
            csr = formats.CSR(len(ptr) - 1, csr.num_cols, np.asarray(ptr, np.uint32), csr.col_idxs, csr.data)
            print(json.dumps({"cut_rows": cut, "rows": csr.num_rows}), flush=True)
        a = ops.DeviceCSR.from_host(csr)
        for n in widths:
            mode = os.environ.get("PROBE_B_MODE", "uniform")
            if mode == "clamped":      # no product small enough to fail the re-association test
                bh = synth.dense_b(csr.num_cols, n)
                bh = np.where(np.abs(bh) < 2.0 ** -10, np.float32(2.0 ** -10), bh).astype(np.float32)
            elif mode == "planted":    # exactly one element fails it: a tiny value in the longest row's first column
                bh = synth.dense_b(csr.num_cols, n, mode="exact")
                lens = np.diff(csr.row_ptrs.astype(np.int64))
                r = int(lens.argmax())
                bh[csr.col_idxs[csr.row_ptrs[r] + 3], 0] = np.float32(2.0 ** -40)
            elif mode == "wide":       # every wave fails it: exponents spread over 2^60
                bh = synth.dense_b(csr.num_cols, n)
                bh = bh * np.exp2(np.random.default_rng(1).integers(-30, 31, size=bh.shape)).astype(np.float32)
            else:
                bh = synth.dense_b(csr.num_cols, n, mode=mode)
            b = torch.from_numpy(bh).cuda()
            c = torch.empty((csr.num_rows, n), device="cuda")
            want = ops.spmm_csr(a, b, kernel=1)
            alg = datasets.csr_algorithmic_bytes(csr, n)
            if a.spans is None:
                a = ops.DeviceCSR.from_host(csr, spans=True)
            for kernel, hint in ((5, False), (6, False), (6, True)):   # hint: kernel 6 takes the rows longest first
                for acc in ("reference", "fast"):
                    st = timer.measure(lambda: ops.spmm_csr(a, b, out=c, kernel=kernel, acc=acc, stream=stream, use_hint=hint), 100,
                                       rounds=3, precondition_s=0.01)
                    stream.synchronize()
                    rec = {"matrix": name, "n": n, "kernel": kernel, "acc": acc, "us": round(st["median_us"], 3),
                           "roofline": round(alg / (st["median_us"] * 1e-6) / 8e12, 3), "tag": capi.last_kernel()}
                    if acc == "reference":
                        rec["bit_exact"] = bool(torch.equal(c, want))
                    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
