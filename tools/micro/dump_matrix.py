import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import datasets
csr = datasets.load_csr(sys.argv[2] if len(sys.argv) > 2 else "n4c6-b13")
with open(sys.argv[1], "wb") as f:
    np.array([csr.num_rows, csr.num_cols, csr.nnz], np.uint32).tofile(f)
    csr.row_ptrs.astype(np.uint32).tofile(f); csr.col_idxs.astype(np.uint32).tofile(f); csr.data.astype(np.float32).tofile(f)
