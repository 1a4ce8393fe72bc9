"""How many B rows could the rows of one workgroup share?  CPU only.
For every non-zero the row-gather kernel delivers one B-row slice through the CU's vector L1; rows of a workgroup that
hold the same column could fetch that slice once (an LDS-shared B tile) -- IF rows with common columns can be brought
into one workgroup.  This counts, for workgroups of w rows, the distinct columns per workgroup summed over the matrix
(= slices fetched with perfect sharing inside a workgroup) against nnz (= slices fetched today), for the storage order
and for a greedy grouping (seed row, then repeatedly the unplaced row with most columns in common with the group).
  python tools/probe/row_sharing_probe.py [matrix ...]"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import datasets  # noqa: E402


def distinct(rows, order, w):
    total = 0
    for i in range(0, len(order), w):
        s = set()
        for r in order[i:i + w]:
            s |= rows[r]
        total += len(s)
    return total


def greedy(rows, col_rows, w):
    m = len(rows)
    used, order = np.zeros(m, bool), []
    for seed in range(m):
        if used[seed]:
            continue
        group, cols, cnt = [seed], set(rows[seed]), collections.Counter()
        used[seed] = True
        for c in rows[seed]:
            for r in col_rows[c]:
                if not used[r]:
                    cnt[r] += 1
        while len(group) < w and cnt:
            r, _ = cnt.most_common(1)[0]
            del cnt[r]
            if used[r]:
                continue
            used[r] = True
            group.append(r)
            for c in rows[r] - cols:
                cols.add(c)
                for r2 in col_rows[c]:
                    if not used[r2]:
                        cnt[r2] += 1
        order += group
    return order


def main():
    for name in sys.argv[1:] or ["n4c6-b13", "ACTIVSg10K", "delaunay_n12"]:
        csr = datasets.load_csr(name)
        rp, ci = csr.row_ptrs.astype(np.int64), csr.col_idxs.astype(np.int64)
        rows = [set(ci[rp[r]:rp[r + 1]].tolist()) for r in range(csr.num_rows)]
        col_rows = collections.defaultdict(list)
        for r, s in enumerate(rows):
            for c in s:
                col_rows[c].append(r)
        deg = np.bincount(ci, minlength=csr.num_cols)
        print(f"# {name}: {csr.num_rows} x {csr.num_cols}, nnz {csr.nnz}, entries per occupied column mean {deg[deg > 0].mean():.2f}")
        print("  rows per workgroup   storage order        greedy groups")
        for w in (8, 16, 32, 64):
            nat, gr = distinct(rows, list(range(csr.num_rows)), w), distinct(rows, greedy(rows, col_rows, w), w)
            print(f"  {w:4d}                 {nat:7d} ({nat / csr.nnz:.2f})      {gr:7d} ({gr / csr.nnz:.2f})")


if __name__ == "__main__":
    main()
