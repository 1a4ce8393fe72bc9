"""Prototype input for tools/micro/cluster_lds.hip: rows of a uniform-row CSR clustered (greedy, shared columns) into
parts of R rows; per cluster the sorted distinct columns and, per entry, its index in that list.
  python tools/micro/cluster_plan.py /tmp/plan.bin [matrix] [R]
File: u32 M, K, width, R, nclusters, maxdist | per cluster: u32 ndist, u32 cols[maxdist] | u32 row_of[nclusters*R]
(0xFFFFFFFF = padding row) | u16 local[nclusters*R*width] | f32 vals[nclusters*R*width] (cluster-major, row, slot)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mispmm import datasets  # noqa: E402
from kernel_sweep import greedy_cluster  # noqa: E402


def main():
    out = sys.argv[1]
    name = sys.argv[2] if len(sys.argv) > 2 else "n4c6-b13"
    r = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    csr = datasets.load_csr(name)
    lens = np.diff(csr.row_ptrs.astype(np.int64))
    assert np.all(lens == lens[0]), "uniform rows only"
    w, m = int(lens[0]), csr.num_rows
    nclusters = -(-m // r)
    order = greedy_cluster(csr, nclusters) if r > 1 and os.environ.get("NO_CLUSTER") != "1" else np.arange(m)
    cols = csr.col_idxs.reshape(m, w)
    vals = csr.data.reshape(m, w)
    row_of = np.full(nclusters * r, 0xFFFFFFFF, np.uint32)
    row_of[:m] = order
    dist, local = [], np.zeros((nclusters * r, w), np.uint16)
    v = np.zeros((nclusters * r, w), np.float32)
    for c in range(nclusters):
        rows = order[c * r:(c + 1) * r]
        d, inv = np.unique(cols[rows], return_inverse=True)
        dist.append(d.astype(np.uint32))
        local[c * r:c * r + len(rows)] = inv.reshape(len(rows), w)
        v[c * r:c * r + len(rows)] = vals[rows]
    maxdist = max(len(d) for d in dist)
    total = sum(len(d) for d in dist)
    print(f"{name}: {m} rows x {w}, clusters of {r}: {nclusters}, distinct columns per cluster mean {total / nclusters:.0f} "
          f"max {maxdist}, reuse {m * w / total:.2f}x, LDS for 256-byte segments {maxdist * 256 / 1024:.0f} KB")
    with open(out, "wb") as f:
        np.array([m, csr.num_cols, w, r, nclusters, maxdist], np.uint32).tofile(f)
        for d in dist:
            np.array([len(d)], np.uint32).tofile(f)
            np.concatenate([d, np.zeros(maxdist - len(d), np.uint32)]).tofile(f)
        row_of.tofile(f)
        local.tofile(f)
        v.tofile(f)


if __name__ == "__main__":
    main()
