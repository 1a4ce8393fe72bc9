#!/usr/bin/env python3
"""Sparsity-sweep inputs (the reference's utils/python_utils/gen_sparse.py + test/sparsity.sh):
2048 x 2048 random A at density 0.1 .. 0.9 and a 2048 x 1024 dense B, values uniform in (-100, 100),
written as `sp_<d>_2048x2048/{matrix.csr, matrix.coo, dense.in}`.  Unlike the reference's
generator this one is SEEDED, so a directory can be regenerated bit for bit.

  python tools/gen_sparse.py <out_dir> [--rows 2048 --cols 2048 --k 1024 --densities 0.1,...,0.9 --seed 20241218]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import formats  # noqa: E402


def random_csr(rows, cols, density, rng, lo=-100.0, hi=100.0):
    """Exactly round(density * rows * cols) entries, positions uniform without replacement."""
    nnz = int(round(density * rows * cols))
    flat = np.sort(rng.choice(rows * cols, size=nnz, replace=False))
    r, c = flat // cols, flat % cols
    ptr = np.zeros(rows + 1, dtype=np.int64)
    np.add.at(ptr, r + 1, 1)
    vals = rng.uniform(lo, hi, size=nnz).astype(np.float32)
    return formats.CSR(rows, cols, np.cumsum(ptr).astype(np.uint32), c.astype(np.uint32), vals)


def write_dense_fast(path, b):
    with open(path, "w") as f:
        f.write(f"{b.shape[0]} {b.shape[1]}\n")
        np.savetxt(f, b, fmt="%.9g")


def generate(out_dir, rows=2048, cols=2048, k=1024, densities=(0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9),
             seed=20241218, coo=True):
    made = []
    for d in densities:
        rng = np.random.default_rng([seed, int(round(d * 1000))])
        path = os.path.join(out_dir, f"sp_{d}_{rows}x{cols}")
        os.makedirs(path, exist_ok=True)
        csr = random_csr(rows, cols, d, rng)
        formats.write_csr(os.path.join(path, "matrix.csr"), csr)
        if coo:
            formats.write_coo(os.path.join(path, "matrix.coo"), formats.csr_to_coo(csr))
        write_dense_fast(os.path.join(path, "dense.in"), rng.uniform(-100, 100, size=(cols, k)).astype(np.float32))
        made.append(path)
    return made


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("out_dir")
    p.add_argument("--rows", type=int, default=2048)
    p.add_argument("--cols", type=int, default=2048)
    p.add_argument("--k", type=int, default=1024)
    p.add_argument("--densities", default="0.1,0.2,0.3,0.4,0.5,0.6,0.7,0.8,0.9")
    p.add_argument("--seed", type=int, default=20241218)
    a = p.parse_args()
    for path in generate(a.out_dir, a.rows, a.cols, a.k, tuple(float(x) for x in a.densities.split(",")), a.seed):
        print(path)
