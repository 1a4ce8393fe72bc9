"""Generate tests/golden/ fixtures (run in the BUILD CONTAINER only).

Two kinds of fixture, both DATA (inputs and expected outputs), no reference code:

1. Files the reference's own repository commits as its known-answer data
   (data/small_10x10, data/small_32x32: .csr/.coo/dense.in/result.expect,
   SURVEY.md section 4) -- copied verbatim.
2. Vectors produced by RUNNING the reference's Python tooling, imported from
   /root/reference/utils/python_utils:
     convert_mtx.process_mtx      -> .csr/.coo/.bsr(1x1)/row-ELL/col-ELL/dense.in
     convert_matrix.save_bsr_matrix(block_size=(b,b)) -> .bsr with real blocks
     validate.calculate_result    -> expected A@B (scipy, float64) = result.expect
   For the large BASELINE configs the expected product is reduced to row sums,
   column sums and 2048 sampled elements (expected_large.npz) so the fixtures stay
   small; B is the seeded synthetic generator (mispmm.synth), A the .mtx file.

The GPU box has no /root/reference: tests only read the files written here.
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
sys.path.insert(0, os.path.join(REF, "utils", "python_utils"))

import convert_matrix  # noqa: E402  (reference)
import convert_mtx  # noqa: E402  (reference)
import validate  # noqa: E402  (reference)
from scipy.io import mmread  # noqa: E402

from mispmm import synth  # noqa: E402

COPIED = {
    "small_10x10": ["sparse.csr", "sparse.coo", "dense.in", "result.expect"],
    "small_32x32": ["Hamrle1.csr", "Hamrle1.coo", "dense.in", "result.expect"],
}
GENERATED_DIRS = ["small_210", "small_32x32", "small_10x10"]
LARGE = [
    # (tag, mtx path, K, synthetic mode)
    ("n4c6-b13_k128_uniform", "large_25605/n4c6-b13.mtx", 128, "uniform"),
    ("n4c6-b13_k128_exact", "large_25605/n4c6-b13.mtx", 128, "exact"),
    ("n4c6-b13_k256_uniform", "large_25605/n4c6-b13.mtx", 256, "uniform"),
    ("n4c6-b13_k512_uniform", "large_25605/n4c6-b13.mtx", 512, "uniform"),
    ("delaunay_n12_k128_uniform", "medium_4096/dense.mtx", 128, "uniform"),
    ("ACTIVSg10K_k128_uniform", "large_20000/ACTIVSg10K.mtx", 128, "uniform"),
    ("GL7d25_k64_uniform", "large_21074/GL7d25.mtx", 64, "uniform"),
]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    for d, files in COPIED.items():
        os.makedirs(os.path.join(HERE, d), exist_ok=True)
        for f in files:
            shutil.copyfile(os.path.join(REF, "data", d, f), os.path.join(HERE, d, f))

    # --- run the reference converter + validator on scratch copies -----------------
    for d in GENERATED_DIRS:
        with tempfile.TemporaryDirectory() as tmp:
            work = os.path.join(tmp, d)
            os.makedirs(work)
            for f in os.listdir(os.path.join(REF, "data", d)):
                if f.endswith(".mtx"):
                    shutil.copyfile(os.path.join(REF, "data", d, f), os.path.join(work, f))
            quiet(convert_mtx.process_mtx, work)
            quiet(validate.validate, work)          # writes result.expect (scipy A@B)
            out = os.path.join(HERE, d + "_generated")
            os.makedirs(out, exist_ok=True)
            for f in sorted(os.listdir(work)):
                if not f.endswith(".mtx"):
                    shutil.copyfile(os.path.join(work, f), os.path.join(out, f))
            # BSR with real blocks through the reference's other writer
            for f in os.listdir(work):
                if f.endswith(".mtx") and f != "dense.mtx":
                    m = mmread(os.path.join(work, f)).tocsr()
                    for b in (2, 4):
                        if m.shape[0] % b == 0 and m.shape[1] % b == 0:
                            quiet(convert_matrix.save_bsr_matrix, m.tobsr((b, b)),
                                  os.path.join(out, f"{os.path.splitext(f)[0]}_b{b}.bsr"), block_size=(b, b))

    # --- BSR-16 of ACTIVSg10K: header + index arrays + value checksum --------------
    m = mmread(os.path.join(REF, "data", "large_20000/ACTIVSg10K.mtx")).tocsr()
    with tempfile.TemporaryDirectory() as tmp:
        p = os.path.join(tmp, "a.bsr")
        quiet(convert_matrix.save_bsr_matrix, m.tobsr((16, 16)), p, block_size=(16, 16))
        with open(p) as f:
            header = [int(x) for x in f.readline().split()]
            ptrs = np.array(f.readline().split(), dtype=np.int64)
            idxs = np.array(f.readline().split(), dtype=np.int64)
            vals = np.array(f.read().split(), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "ACTIVSg10K_bsr16_index.npz"), header=np.array(header),
                        block_row_ptrs=ptrs.astype(np.uint32), block_col_idxs=idxs.astype(np.uint32),
                        value_sum=vals.sum(), value_abs_sum=np.abs(vals).sum(),
                        block_sums=vals.reshape(-1, 256).sum(axis=1))

    # --- large configs: reduced expected products ----------------------------------
    rng = np.random.default_rng(20241218)
    large = {}
    for tag, rel, k, mode in LARGE:
        a = mmread(os.path.join(REF, "data", rel)).tocoo()
        b = synth.dense_b(a.shape[1], k, mode=mode).astype(np.float64)
        c = np.asarray(validate.calculate_result(a, b))          # reference's expected-result formula
        absa = abs(a)
        scale = np.asarray(absa @ np.abs(b))                     # sum |a||b| per element, for tolerances
        rows = rng.integers(0, c.shape[0], 2048)
        cols = rng.integers(0, c.shape[1], 2048)
        large[tag + "/row_sums"] = c.sum(axis=1)
        large[tag + "/col_sums"] = c.sum(axis=0)
        large[tag + "/sample_rows"] = rows.astype(np.int32)
        large[tag + "/sample_cols"] = cols.astype(np.int32)
        large[tag + "/sample_vals"] = c[rows, cols]
        large[tag + "/sample_scale"] = scale[rows, cols]
        large[tag + "/row_scale"] = scale.sum(axis=1)
        large[tag + "/shape"] = np.array(c.shape)
    np.savez_compressed(os.path.join(HERE, "expected_large.npz"), **large)
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
