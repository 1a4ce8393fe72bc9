"""A handful of eager launches of kernel 6 in both accumulate modes, for `rocprofv3 --pmc ... -- python3 <this>`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import datasets, ops, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "GL7d25"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
csr = datasets.load_csr(name)
a = ops.DeviceCSR.from_host(csr)
b = torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda()
c = torch.empty((csr.num_rows, n), device="cuda")
for acc in ("reference", "fast"):
    for _ in range(20):
        ops.spmm_csr(a, b, out=c, kernel=6, acc=acc)
torch.cuda.synchronize()
