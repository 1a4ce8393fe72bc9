import sys, subprocess, numpy as np, tempfile, pathlib
sys.path.insert(0, "cuda-optimization-for-spmm_amd")
from mispmm import datasets, formats
d = pathlib.Path(tempfile.mkdtemp()) / "large_25605"; d.mkdir()
csr = datasets.load_csr("n4c6-b13", dtype=np.float64)
formats.write_csr(d / "n4c6-b13.csr", csr, integer=True)
formats.write_ell_colmajor(d / "n4c6-b13_rowind.ell", d / "n4c6-b13_values_colmajor.ell", formats.csr_to_ell_colmajor(csr, reference_width=True), integer=True)
for it in ("20", "200", "2000"):
    p = subprocess.run(["cuda-optimization-for-spmm_amd/cuspmm", "--csr", "--ell", "-k", "128", "--iters", it, "--no-vendor", "-d", str(d)], capture_output=True, text=True)
    print("iters", it); print(p.stdout[-3000:]); print(p.stderr[-500:])
