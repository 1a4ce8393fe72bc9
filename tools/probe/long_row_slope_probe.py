"""Per-entry cost of ONE long row on an otherwise lightly loaded GPU: 2800 rows of 8 entries plus one row of L entries,
K = 21074 columns, N = 128.  T(L) slope = time per entry of the sequential chain.  Run with the tuning build and
MISPMM_LONGROWS=0/1 to compare the lane-group kernel with csr_wave_deep."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, formats, ops, synth  # noqa: E402
import bench  # noqa: E402


def main():
    capi.lib()
    rng = np.random.default_rng(3)
    m, k, n = 2800, 21074, int(os.environ.get("PROBE_N", "128"))
    stream = torch.cuda.Stream()
    timer = bench.Timer(stream)
    b = torch.from_numpy(synth.dense_b(k, n)).cuda()
    c = torch.empty((m, n), device="cuda")
    for L in (8, 64, 128, 256, 512, 1024, 2048):
        lens = np.full(m, 8, dtype=np.int64)
        lens[m // 2] = L
        ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        cols = np.concatenate([np.sort(rng.choice(k, size=x, replace=False)) for x in lens]).astype(np.uint32)
        vals = rng.uniform(-1, 1, int(ptr[-1])).astype(np.float32)
        a = ops.DeviceCSR.from_host(formats.CSR(m, k, ptr, cols, vals))
        for acc in ("reference", "fast"):
            st = timer.measure(lambda: ops.spmm_csr(a, b, out=c, acc=acc, stream=stream), 100, rounds=3, precondition_s=0.01)
            print(json.dumps({"L": L, "acc": acc, "us": round(st["median_us"], 3), "kernel": capi.last_kernel()}), flush=True)


if __name__ == "__main__":
    main()
