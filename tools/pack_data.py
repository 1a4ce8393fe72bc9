"""Pack the SuiteSparse inputs the reference's data/ directory holds into small
compressed .npz files under data/ (run in the build container only; the GPU box
has no /root/reference).  These are DATA files (public SuiteSparse matrices the
reference's test scripts consume, test/csr.sh:3-14), not reference source.

Each .npz holds the matrix as CSR after symmetric expansion and duplicate
summation (what `scipy.io.mmread(...).tocsr()` gives the reference's converter,
utils/python_utils/convert_mtx.py:103): num_rows, num_cols, row_ptrs, col_idxs,
data (float64), field ('real'|'integer'|'pattern'), source (relative path).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import formats  # noqa: E402

REF_DATA = "/root/reference/data"
MATRICES = {
    # name -> relative .mtx path ; roles per BASELINE.json configs
    "n4c6-b13": "large_25605/n4c6-b13.mtx",        # headline / configs 3,5
    "ACTIVSg10K": "large_20000/ACTIVSg10K.mtx",    # config 4 (BSR-16)
    "delaunay_n12": "medium_4096/dense.mtx",       # config 2 stand-in (HFE18_96_in.mtx missing)
    "Hamrle1": "small_32x32/Hamrle1.mtx",          # config 1
    "n3c5-b6": "small_210/n3c5-b6.mtx",
    "sparse10x10": "small_10x10/sparse.mtx",       # symmetric integer: exercises mirror expansion
    "qh1484": "medium_1484/qh1484.mtx",
    "dw1024": "medium_2048/dw1024.mtx",
    "tols4000": "medium_4000/tols4000.mtx",
    "ch7-6-b5": "large_15120/ch7-6-b5.mtx",
    "GL7d25": "large_21074/GL7d25.mtx",
    "g7jac010": "medium_2880/g7jac010.mtx",
}


def main():
    out_dir = os.path.join(ROOT, "data")
    os.makedirs(out_dir, exist_ok=True)
    for name, rel in MATRICES.items():
        coo, field = formats.read_mtx(os.path.join(REF_DATA, rel))
        csr = formats.coo_to_csr(coo, dtype=np.float64)
        data = csr.data
        if field in ("integer", "pattern") and np.all(data == np.round(data)) and np.abs(data).max() < 127:
            data = data.astype(np.int8)      # +-1 matrices pack to almost nothing
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, num_rows=csr.num_rows, num_cols=csr.num_cols, row_ptrs=csr.row_ptrs,
                            col_idxs=csr.col_idxs, data=data, field=np.array(field), source=np.array(rel))
        print(f"{name}: {csr.num_rows}x{csr.num_cols} nnz={csr.nnz} field={field} -> "
              f"{os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
