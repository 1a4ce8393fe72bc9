#!/usr/bin/env python3
"""Offline golden checker, the role of the reference's utils/python_utils/validate.py: walk a data
directory tree; in every directory that holds a sparse matrix (any of *.mtx, *.csr, *.coo) and a dense
operand (dense.in or dense.mtx), take the expected product from `result.expect` or compute it in
float64 and write it (10 decimals, as the reference does), then compare every `*.out` dump (what
`cuspmm --save` or DenseMatrix::save2File writes) against it with numpy.allclose.

  python tools/validate.py <directory> [--rtol 1e-5 --atol 1e-8]
Exit code 1 if any dump mismatches.  From-scratch: uses this repository's own readers (mispmm.formats).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import formats  # noqa: E402


def load_sparse(root, files):
    for f in sorted(files):
        p = os.path.join(root, f)
        if f.endswith(".mtx") and f != "dense.mtx":
            coo, _ = formats.read_mtx(p)
            return formats.coo_to_csr(coo, dtype=np.float64), p
        if f.endswith(".csr"):
            return formats.read_csr(p, dtype=np.float64), p
        if f.endswith(".coo"):
            return formats.coo_to_csr(formats.read_coo(p, dtype=np.float64), dtype=np.float64), p
    return None, None


def load_dense(root, files):
    if "dense.in" in files:
        return formats.read_dense(os.path.join(root, "dense.in"), dtype=np.float64).data
    if "dense.mtx" in files:
        coo, _ = formats.read_mtx(os.path.join(root, "dense.mtx"))
        return formats.coo_to_csr(coo, dtype=np.float64).to_dense()
    return None


def load_matrix_text(path):
    """A result dump: optional `rows cols [COL_MAJOR]` header line, then rows of numbers."""
    with open(path) as f:
        first = f.readline().split()
        rest = np.loadtxt(f, ndmin=2)
    header = len(first) in (2, 3) and all(t.isdigit() for t in first[:2]) and rest.size and rest.shape[1] != len(first)
    if header:
        rows, cols = int(first[0]), int(first[1])
        if len(first) == 3 and first[2] == "COL_MAJOR":
            return rest.reshape(cols, rows).T
        return rest.reshape(rows, cols)
    head = np.array([float(t) for t in first], dtype=np.float64).reshape(1, -1)
    return np.vstack([head, rest]) if rest.size else head


def process(root, files, rtol, atol):
    a, a_path = load_sparse(root, files)
    b = load_dense(root, files)
    if a is None or b is None:
        print(f"Skipping directory {root}: missing sparse matrix or dense operand.")
        return True
    expect_path = os.path.join(root, "result.expect")
    if os.path.exists(expect_path):
        expected = np.loadtxt(expect_path, ndmin=2)
        print(f"\tExpect file found: {expect_path}")
    else:
        expected = a.to_dense() @ b
        with open(expect_path, "w") as f:
            for row in expected:
                f.write(" ".join(f"{v:.10f}" for v in row) + "\n")
        print(f"\tCalculated expected result from {a_path} and saved it to {expect_path}")
    ok = True
    for f in sorted(files):
        if not f.endswith(".out"):
            continue
        got = load_matrix_text(os.path.join(root, f))
        if got.shape == expected.shape and np.allclose(got, expected, rtol=rtol, atol=atol):
            print(f"\tResult file {f} matches the expected result.")
        else:
            ok = False
            diff = np.abs(got - expected).max() if got.shape == expected.shape else float("nan")
            print(f"Result file {f} does NOT match the expected result (shape {got.shape} vs {expected.shape}, max |diff| {diff}).")
    return ok


def main():
    p = argparse.ArgumentParser()
    p.add_argument("directory")
    p.add_argument("--rtol", type=float, default=1e-5)
    p.add_argument("--atol", type=float, default=1e-8)
    a = p.parse_args()
    good = True
    for root, _, files in os.walk(a.directory):
        print(f"Processing directory: {root}")
        good &= process(root, files, a.rtol, a.atol)
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()
