"""Named workloads: the SuiteSparse matrices of the reference's data/ directory
(packed by tools/pack_data.py into <repo>/data/*.npz so they travel to the GPU
box) and the BASELINE.json configurations built on them."""
import os

import numpy as np

from . import formats

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "data")

# reference data directory -> packed matrix name (test/csr.sh:3-14)
DIR_TO_MATRIX = {
    "small_10x10": "sparse10x10", "small_32x32": "Hamrle1", "small_210": "n3c5-b6", "medium_1484": "qh1484", "medium_2048": "dw1024",
    "medium_2880": "g7jac010", "medium_4000": "tols4000",
    "medium_4096": "delaunay_n12",   # stand-in: HFE18_96_in.mtx is missing from the checkout
    "large_15120": "ch7-6-b5", "large_20000": "ACTIVSg10K", "large_21074": "GL7d25", "large_25605": "n4c6-b13",
}


def available():
    return sorted(f[:-4] for f in os.listdir(DATA_DIR) if f.endswith(".npz"))


def load_csr(name, dtype=np.float32):
    """name: a packed matrix name ('n4c6-b13') or a reference data dir ('large_25605')."""
    name = DIR_TO_MATRIX.get(name, name)
    path = os.path.join(DATA_DIR, name + ".npz")
    if not os.path.exists(path):
        raise FileNotFoundError(f"no packed matrix {name!r} under {DATA_DIR} (have: {available()})")
    z = np.load(path, allow_pickle=False)
    return formats.CSR(int(z["num_rows"]), int(z["num_cols"]), z["row_ptrs"].astype(np.uint32),
                       z["col_idxs"].astype(np.uint32), z["data"].astype(dtype))


def csr_algorithmic_bytes(csr, n_cols, elem=4):
    """A once + every B row once + C written once."""
    return csr.nnz * (4 + elem) + (csr.num_rows + 1) * 4 + csr.num_cols * n_cols * elem + csr.num_rows * n_cols * elem


def ell_algorithmic_bytes(num_rows, width, num_cols, n_cols, elem=4):
    return num_rows * width * (4 + elem) + num_cols * n_cols * elem + num_rows * n_cols * elem


def bsr_algorithmic_bytes(bsr, n_cols, elem=4, out_elem=4):
    return (bsr.num_blocks * bsr.block_row_size * bsr.block_col_size * elem + bsr.num_blocks * 4
            + (bsr.num_block_rows + 1) * 4 + bsr.num_cols * n_cols * elem + bsr.num_rows * n_cols * out_elem)


def spmm_flops(nnz, n_cols):
    return 2 * nnz * n_cols
