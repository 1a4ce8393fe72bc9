"""bf16 BSR-16 kernel time against the NUMBER of block rows launched (every s-th block row of ACTIVSg10K, same B):
a flat curve means the launch is bound by one workgroup's dependent chain, a linear one by throughput.  GPU box only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mispmm import datasets, formats, ops, synth  # noqa: E402
from config_sweep import timed  # noqa: E402


def take_block_rows(bsr, rows):
    p = bsr.block_row_ptrs.astype(np.int64)
    idx = np.concatenate([np.arange(p[r], p[r + 1]) for r in rows]) if len(rows) else np.zeros(0, np.int64)
    lens = np.array([p[r + 1] - p[r] for r in rows])
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    data = bsr.data[idx]
    return formats.BSR(len(rows) * 16, bsr.num_cols, int(data.size), 16, 16, ptr, bsr.block_col_idxs[idx], data)


def main():
    s = torch.cuda.Stream()
    csr = datasets.load_csr("ACTIVSg10K")
    full = formats.csr_to_bsr(csr, 16)
    lens = np.diff(full.block_row_ptrs.astype(np.int64))
    b16 = ops.f32_to_bf16(torch.from_numpy(synth.dense_b(csr.num_cols, 128)).cuda())
    order = np.argsort(-lens)
    cases = [("every 16th block row", list(range(0, 1250, 16))), ("every 8th", list(range(0, 1250, 8))),
             ("every 4th", list(range(0, 1250, 4))), ("every 2nd", list(range(0, 1250, 2))), ("all", list(range(1250))),
             ("the 8 longest block rows only", list(order[:8])), ("the 78 longest", list(order[:78])),
             ("the 78 shortest", list(order[-78:]))]
    for tag, rows in cases:
        sub = take_block_rows(full, rows)
        a = ops.DeviceBSR.from_host(sub)
        blocks16 = ops.f32_to_bf16(a.data)
        c = torch.empty((sub.num_rows, 128), dtype=torch.float32, device="cuda")
        us = timed(lambda: ops.spmm_bsr_bf16(a, blocks16, b16, out_bf16=False, out=c, stream=s), s)
        sl = np.diff(sub.block_row_ptrs.astype(np.int64))
        print(f"{tag}: {len(rows)} block rows, {sub.num_blocks} blocks, longest {int(sl.max())}: {us:.2f} us", flush=True)


if __name__ == "__main__":
    main()
