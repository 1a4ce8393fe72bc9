"""The general CSR entry point (no uniform-row hint) on the BASELINE matrices: us per product, REFERENCE mode.
MISPMM_LIB selects the library build, for A/B runs of kernel changes."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, datasets, ops, synth  # noqa: E402
import bench  # noqa: E402


def main():
    capi.lib()
    stream = torch.cuda.Stream()
    timer = bench.Timer(stream)
    for name, n in (("n4c6-b13", 128), ("n4c6-b13", 256), ("n4c6-b13", 512), ("delaunay_n12", 128), ("ACTIVSg10K", 128), ("g7jac010", 128),
                    ("tols4000", 128), ("ch7-6-b5", 128)):
        csr = datasets.load_csr(name)
        a = ops.DeviceCSR.from_host(csr)
        b = torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda()
        c = torch.empty((csr.num_rows, n), device="cuda")
        alg = datasets.csr_algorithmic_bytes(csr, n)
        st = timer.measure(lambda: ops.spmm_csr(a, b, out=c, stream=stream, use_hint=False), 200, rounds=5, precondition_s=0.02)
        print(json.dumps({"matrix": name, "n": n, "us": round(st["median_us"], 3), "min_us": round(st["min_us"], 3),
                          "roofline": round(alg / (st["median_us"] * 1e-6) / 8e12, 3), "tag": capi.last_kernel(),
                          "lib": os.path.basename(capi.LIB_PATH)}), flush=True)


if __name__ == "__main__":
    main()
