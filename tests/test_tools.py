"""The reference's integration sweeps, restated as tools/ (test/{csr,coo,bsr}.sh -> tools/sweep.py,
utils/python_utils/gen_sparse.py + test/sparsity.sh -> tools/gen_sparse.py + tools/sparsity_sweep.py).
CPU part: the generator is seeded, writes the reference's text formats, hits the requested density.
GPU part: both sweeps driven end to end on tiny inputs; every record of every HIP kernel must say correct = 1."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_gen_sparse_is_seeded_and_writes_reference_formats(tmp_path):
    import gen_sparse
    from mispmm import formats
    a = gen_sparse.generate(str(tmp_path / "a"), rows=64, cols=48, k=8, densities=(0.25,), seed=7)[0]
    b = gen_sparse.generate(str(tmp_path / "b"), rows=64, cols=48, k=8, densities=(0.25,), seed=7)[0]
    assert os.path.basename(a) == "sp_0.25_64x48"                       # directory naming of gen_sparse.py:63-84
    for f in ("matrix.csr", "matrix.coo", "dense.in"):
        assert open(os.path.join(a, f)).read() == open(os.path.join(b, f)).read(), f"{f} differs between two runs"
    csr = formats.read_csr(os.path.join(a, "matrix.csr"))
    assert (csr.num_rows, csr.num_cols, csr.nnz) == (64, 48, round(0.25 * 64 * 48))
    assert np.all(np.abs(csr.data) <= 100.0)
    coo = formats.read_coo(os.path.join(a, "matrix.coo"))
    assert np.array_equal(coo.col_idxs, csr.col_idxs) and np.all(np.diff(coo.row_idxs.astype(np.int64)) >= 0)
    dense = formats.read_dense(os.path.join(a, "dense.in")).data
    assert dense.shape == (48, 8)
    c = gen_sparse.generate(str(tmp_path / "c"), rows=64, cols=48, k=8, densities=(0.25,), seed=8)[0]
    assert open(os.path.join(a, "matrix.csr")).read() != open(os.path.join(c, "matrix.csr")).read()


@pytest.mark.gpu
def test_sweep_harness_over_data_directories(tmp_path):
    """tools/sweep.py = the reference's test/csr.sh + coo.sh + bsr.sh (+ ELL) loops, on two small directories."""
    out = tmp_path / "sweep"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep.py"), "--dirs", "small_210,medium_1484",
                        "--formats", "csr,coo,bsr,ell", "-k", "32", "--iters", "5", "--block", "4", "--out", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert re.search(r"\d+ records, 0 incorrect or failed", p.stdout), p.stdout[-2000:]
    rows = [json.loads(l) for l in open(out / "summary.jsonl")]
    assert {r["format"] for r in rows} == {"CSR", "COO", "BSR", "ELL"} and {r["dir"] for r in rows} == {"small_210", "medium_1484"}
    gpu = [r for r in rows if r["kernel"] not in ("0", "-1")]
    assert gpu and all("gflops" in r and r["gflops"] > 0 for r in gpu if r["correct"] == "1")
    for fmt in ("csr", "coo", "bsr", "ell"):                                  # `>> csr.json` of the shell scripts
        assert os.path.getsize(out / f"{fmt}.json") > 0


@pytest.mark.gpu
def test_sparsity_sweep_harness(tmp_path):
    """tools/sparsity_sweep.py = test/sparsity.sh on generated directories (small here: 256 x 256, two densities)."""
    out = tmp_path / "sparsity"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sparsity_sweep.py"), "--densities", "0.1,0.5", "--rows", "256",
                        "--cols", "256", "--k", "64", "--iters", "5", "--out", str(out)], capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("density")]
    assert len(lines) == 2 * (8 + 4)                  # CSR: kernels 0-6 and rocSPARSE; COO: 0, 1, 2 and rocSPARSE
    ours = [l for l in lines if " kernel -1 " not in l]
    assert all(" correct 1 " in l for l in ours), [l for l in ours if " correct 1 " not in l]
    assert os.path.getsize(out / "sparsity.json") > 0
