"""Run by tests/test_gpu_spmm.py::test_row_stream_* in a process of its own (MISPMM_LIB = the tuning build, MISPMM_STREAM=1
forces the persistent row-walking launch of row_stream.hpp wherever a shape has an instance): small and odd shapes against the
oracle, bit for bit.  Prints one line per case; exits non-zero on the first difference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-optimization-for-spmm_amd"))
from mispmm import capi, formats, ops, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def uniform_csr(m, k, w, seed):
    rng = np.random.default_rng(seed)
    cols = np.concatenate([np.sort(rng.choice(k, size=w, replace=False)) for _ in range(m)]).astype(np.uint32)
    return formats.CSR(m, k, (np.arange(m + 1) * w).astype(np.uint32), cols, rng.uniform(-2, 2, m * w).astype(np.float32))


def main():
    orc.build()
    capi.lib()
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    cases = 0
    # CSR with a constant row length: widths 9..16 (the body lengths 10 / 12 / 14 / 16 with and without dropped slots), row counts
    # that leave a ragged last workgroup / fewer rows than lane groups / several rows per lane group, N over the XCD tilings
    # (4x2, 2x4 / 1x8, 1x8) and the 8-lane groups of 32-column parts, both accumulate modes
    for m, k, w, n in ((6300, 3000, 14, 128), (6299, 3000, 13, 256), (50, 200, 9, 128), (20001, 900, 10, 64), (4099, 5000, 16, 512),
                       (70000, 400, 12, 128), (1234, 777, 15, 96), (9000, 30000, 11, 256)):
        csr = uniform_csr(m, k, w, m + w)
        b = synth.dense_b(k, n)
        ref = orc.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        a = ops.DeviceCSR.from_host(csr, plan=False)
        got = ops.spmm_csr(a, dev(b)).cpu().numpy()
        tag = capi.last_kernel()
        assert "row_stream" in tag, (m, k, w, n, tag)
        assert np.array_equal(got, ref), ("csr reference", m, k, w, n, tag)
        fast = ops.spmm_csr(a, dev(b), acc="fast").cpu().numpy().astype(np.float64)
        scale = orc.spmm_csr(csr.row_ptrs, csr.col_idxs, np.abs(csr.data), np.abs(b)).astype(np.float64)
        assert "row_stream" in capi.last_kernel() and np.all(np.abs(fast - ref) <= 1e-5 * scale + 1e-37), ("csr fast", m, k, w, n)
        # strided operands (ldb > N, ldc > N) and a sentinel in the gap columns of C
        if n in (128, 96):
            bw = torch.full((k, n + 8), 7.0, device="cuda")
            bw[:, :n] = dev(b)
            cw = torch.full((m, n + 4), -3.0, device="cuda")
            ops.spmm_csr(a, bw[:, :n], out=cw[:, :n])
            assert "row_stream" in capi.last_kernel()
            assert np.array_equal(cw[:, :n].cpu().numpy(), ref) and bool((cw[:, n:] == -3.0).all()), ("strided", m, k, w, n)
        # rows in a plan order (row map)
        if n == 128:
            planned = ops.DeviceCSR.from_host(csr, plan=True)
            out = torch.empty((m, n), device="cuda")
            assert ops._csr_plan(planned, [dev(b)], [out], "reference", None) and "row_stream" in capi.last_kernel() and "plan-order" in capi.last_kernel()
            assert np.array_equal(out.cpu().numpy(), ref), ("plan order", m, k, w, n)
        cases += 1
        print("csr", m, k, w, n, tag)
    # ELL: padding in the middle and at the end of rows, empty rows, the reference's fp32 sums
    for m, k, w, n in ((6300, 2000, 14, 256), (3001, 500, 10, 128), (12000, 800, 16, 64), (5000, 600, 12, 512)):
        rng = np.random.default_rng(m)
        cols = np.stack([np.sort(rng.choice(k, size=w, replace=False)) for _ in range(m)]).astype(np.uint32)
        vals = rng.uniform(-2, 2, (m, w)).astype(np.float32)
        pad = rng.random((m, w)) < 0.2
        pad[::7] = True                                             # whole rows of padding
        cols[pad], vals[pad] = 0xFFFFFFFF, 0.0
        ell = formats.ELLRowMajor(m, k, int((~pad).sum()), w, cols, vals)
        b = synth.dense_b(k, n)
        # oracle: the CSR of the occupied slots in slot order with the reference's ELL arithmetic (fp32 product, fp32 add) =
        # the COO oracle over the same entries
        rows = np.repeat(np.arange(m, dtype=np.uint32), w)[~pad.reshape(-1)]
        ref = orc.spmm_coo(m, rows, cols.reshape(-1)[~pad.reshape(-1)], vals.reshape(-1)[~pad.reshape(-1)], b)
        a = ops.DeviceELL.from_host(ell, compact=False)
        got = ops.spmm_ell(a, dev(b)).cpu().numpy()
        assert "row_stream" in capi.last_kernel(), capi.last_kernel()
        assert np.array_equal(got, ref), ("ell", m, k, w, n, capi.last_kernel())
        cases += 1
        print("ell", m, k, w, n, capi.last_kernel())
    # shapes without an instance keep the row-gather kernel (width > 16, width < 9, N not a multiple of 32 per column part)
    for m, k, w, n in ((500, 300, 17, 128), (500, 300, 8, 128), (500, 300, 14, 40)):
        csr = uniform_csr(m, k, w, 5)
        b = synth.dense_b(k, n)
        got = ops.spmm_csr(ops.DeviceCSR.from_host(csr, plan=False), dev(b)).cpu().numpy()
        assert "row_stream" not in capi.last_kernel()
        assert np.array_equal(got, orc.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b))
    print(f"row_stream cases ok: {cases}")


if __name__ == "__main__":
    main()
