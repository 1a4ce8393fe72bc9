#!/usr/bin/env python3
"""The reference's integration sweeps (test/csr.sh, test/coo.sh, test/bsr.sh and an ELL equivalent):
loop the `cuspmm` CLI over the 12 data directories and append its records to <fmt>.json -- plus
GFLOP/s, achieved GB/s and HBM roofline fraction per kernel (--iters).  GPU box only.

The directories are materialised from the packed matrices (data/*.npz) in the reference's text
formats with this repository's own converter (mispmm.formats); the dense operand is the seeded
synthetic B with -k columns (the reference's dense.in files are densified SuiteSparse matrices of
up to 20000 x 20000 and are not carried).  data/medium_4096 uses the stand-in delaunay_n12.

  python tools/sweep.py [--formats csr,coo,bsr,ell] [-k 128] [--iters 200] [--out gpurun_out/sweep]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import datasets, formats  # noqa: E402

CLI = os.path.join(ROOT, "cuda-optimization-for-spmm_amd", "cuspmm")
DIRS = ["small_210", "small_32x32", "small_10x10", "medium_1484", "medium_2048", "medium_2880", "medium_4000",
        "medium_4096", "large_15120", "large_20000", "large_21074", "large_25605"]     # order of test/csr.sh


def materialise(d, root, block):
    name = datasets.DIR_TO_MATRIX[d]
    csr = datasets.load_csr(name, dtype=np.float64)
    out = os.path.join(root, d)
    os.makedirs(out, exist_ok=True)
    integer = bool(np.all(csr.data == np.round(csr.data)))
    formats.write_csr(os.path.join(out, name + ".csr"), csr, integer)
    formats.write_coo(os.path.join(out, name + ".coo"), formats.csr_to_coo(csr), integer)
    b = block if csr.num_rows % block == 0 and csr.num_cols % block == 0 else 1
    formats.write_bsr(os.path.join(out, name + ".bsr"), formats.csr_to_bsr(csr, b), integer)
    formats.write_ell_colmajor(os.path.join(out, name + "_rowind.ell"), os.path.join(out, name + "_values_colmajor.ell"),
                               formats.csr_to_ell_colmajor(csr), integer)
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--formats", default="csr,coo,bsr,ell")
    p.add_argument("-k", type=int, default=128)
    p.add_argument("--iters", type=int, default=200)
    p.add_argument("--block", type=int, default=16, help="BSR block size where the shape allows it (else 1)")
    p.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep"))
    p.add_argument("--dirs", default=",".join(DIRS))
    args = p.parse_args()
    os.makedirs(args.out, exist_ok=True)
    summary = []
    with tempfile.TemporaryDirectory() as tmp:
        for fmt in args.formats.split(","):
            log = os.path.join(args.out, fmt + ".json")
            open(log, "w").close()                       # `rm csr.json; touch csr.json`
            for d in args.dirs.split(","):
                path = materialise(d, tmp, args.block)
                r = subprocess.run([CLI, "--" + fmt, "-k", str(args.k), "--iters", str(args.iters), "-d", path],
                                   capture_output=True, text=True, timeout=900)
                with open(log, "a") as f:
                    f.write(r.stdout)
                if r.returncode != 0:
                    summary.append({"dir": d, "format": fmt, "error": r.stderr.strip()[-200:]})
                    continue
                for body in re.findall(r"\{(.*?)\},", r.stdout, flags=re.S):
                    rec = dict(re.findall(r'"([A-Za-z]+)":"([^"]*)"', body))
                    row = {"dir": d, "format": rec["format"], "kernel": rec["kernelType"], "correct": rec["correct"],
                           "kernel_ms": float(rec["cudaKernelTimeMs"])}
                    if "gflops" in rec:
                        row.update(us=float(rec["steadyKernelUs"]), gflops=float(rec["gflops"]),
                                   hbm_GBps=float(rec["hbmGBps"]), roofline=float(rec["rooflineFrac"]))
                    summary.append(row)
    with open(os.path.join(args.out, "summary.jsonl"), "w") as f:
        for row in summary:
            f.write(json.dumps(row) + "\n")
    bad = [r for r in summary if r.get("correct") == "0" and not (r["format"] == "BSR" and r["kernel"] == "2") or "error" in r]
    print(f"{len(summary)} records, {len(bad)} incorrect or failed")
    for r in bad:
        print("  ", r)
    best = {}
    for r in summary:
        if "gflops" in r and r["correct"] == "1":
            key = (r["dir"], r["format"])
            if key not in best or r["us"] < best[key]["us"]:    # fastest, not most flops: the dense-block BSR kernels count zeros
                best[key] = r
    for (d, fmt), r in sorted(best.items()):
        print(f"{d:14s} {fmt}  best kernel {r['kernel']:>2s}  {r['us']:9.2f} us  {r['gflops']:9.1f} GFLOP/s  "
              f"{r['hbm_GBps']:8.1f} GB/s  roofline {r['roofline']:.3f}")


if __name__ == "__main__":
    main()
