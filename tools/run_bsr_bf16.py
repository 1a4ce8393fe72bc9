"""Launch the BASELINE config-4 kernel (large_20000 BSR-16 x K=128 bf16) N times; for rocprofv3."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import datasets, formats, ops, synth
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 50
out_bf16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
csr = datasets.load_csr("ACTIVSg10K")
bsr = formats.csr_to_bsr(csr, 16)
a = ops.DeviceBSR.from_host(bsr)
b = torch.from_numpy(synth.dense_b(csr.num_cols, 128)).cuda()
blocks16, b16 = ops.f32_to_bf16(a.data), ops.f32_to_bf16(b)
c = torch.empty((csr.num_rows, 128), dtype=torch.int16 if out_bf16 else torch.float32, device="cuda")
for _ in range(n_launch):
    ops.spmm_bsr_bf16(a, blocks16, b16, out_bf16=out_bf16, out=c)
torch.cuda.synchronize()
print("launched", n_launch, "blocks", bsr.num_blocks)
