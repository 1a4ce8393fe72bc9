"""How much of the bf16 BSR-16 kernel's time is block-row length imbalance?  Times the real ACTIVSg10K BSR-16
and synthetic BSRs with the same shape and block count but (a) every block row 26/27 blocks, same columns
spread, (b) the real lengths with random columns.  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mispmm import datasets, formats, ops, synth  # noqa: E402
from config_sweep import timed  # noqa: E402


def synth_bsr(lens, ncb, rng):
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    cols = np.concatenate([np.sort(rng.choice(ncb, int(n), replace=False)) for n in lens]).astype(np.uint32)
    data = rng.uniform(-1, 1, (int(ptr[-1]), 16, 16)).astype(np.float32)
    return formats.BSR(len(lens) * 16, ncb * 16, int(data.size), 16, 16, ptr, cols, data)


def hot_bsr(lens, ncb, span, rng):
    """Same lengths, every block column drawn from [0, span): the B panels stay cache resident."""
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    cols = np.concatenate([np.arange(int(n)) % span for n in lens]).astype(np.uint32)
    data = rng.uniform(-1, 1, (int(ptr[-1]), 16, 16)).astype(np.float32)
    return formats.BSR(len(lens) * 16, ncb * 16, int(data.size), 16, 16, ptr, cols, data)


def main():
    s = torch.cuda.Stream()
    rng = np.random.default_rng(7)
    csr = datasets.load_csr("ACTIVSg10K")
    real = formats.csr_to_bsr(csr, 16)
    lens = np.diff(real.block_row_ptrs.astype(np.int64))
    nb, mb = int(lens.sum()), len(lens)
    even = np.full(mb, nb // mb)
    even[: nb - even.sum()] += 1
    cases = {"real ACTIVSg10K": real, "real lengths, random columns": synth_bsr(lens, mb, rng),
             "even lengths (26/27), random columns": synth_bsr(even, mb, rng),
             "sorted real lengths (longest first), random columns": synth_bsr(np.sort(lens)[::-1], mb, rng),
             "real lengths, 6 hot block columns (B panels L1 resident)": hot_bsr(lens, mb, 6, rng),
             "real lengths, 64 hot block columns (B panels L2 resident)": hot_bsr(lens, mb, 64, rng)}
    b = torch.from_numpy(synth.dense_b(csr.num_cols, 128)).cuda()
    b16 = ops.f32_to_bf16(b)
    c = torch.empty((csr.num_rows, 128), dtype=torch.float32, device="cuda")
    for tag, bsr in cases.items():
        a = ops.DeviceBSR.from_host(bsr)
        blocks16 = ops.f32_to_bf16(a.data)
        us = timed(lambda: ops.spmm_bsr_bf16(a, blocks16, b16, out_bf16=False, out=c, stream=s), s)
        print(f"{tag}: {us:.2f} us  (blocks {bsr.num_blocks}, longest block row {int(np.diff(bsr.block_row_ptrs.astype(np.int64)).max())})", flush=True)


if __name__ == "__main__":
    main()
