"""Seeded random shapes through every format's HIP path, REFERENCE accumulate, bit-exact vs the oracle.
Covers what the curated cases cannot enumerate: odd M / K / N, ragged and empty rows, rows longer than a
chunk, padded leading dimensions, every CSR kernel id, square and rectangular BSR blocks."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from mispmm import formats, ops  # noqa: E402

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("MISPMM_FUZZ_SCALE", "1"))   # more seeds for an occasional deep run


def rand_csr(rng, m, k):
    style = rng.integers(0, 4)
    if style == 0:
        lens = rng.integers(0, min(k, 6) + 1, size=m)                      # short, some empty
    elif style == 1:
        lens = np.minimum(k, rng.geometric(0.08, size=m) - 1)              # long tail
    elif style == 2:
        lens = np.full(m, min(k, int(rng.integers(1, 40))))               # uniform (ELL-like)
    else:
        lens = np.where(rng.random(m) < 0.1, min(k, int(rng.integers(60, 400))), rng.integers(0, 4, size=m))
    lens = np.minimum(lens, k)
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    cols = np.concatenate([np.sort(rng.choice(k, size=int(n), replace=False)) for n in lens] + [np.zeros(0, np.int64)])
    vals = rng.standard_normal(int(ptr[-1])).astype(np.float32) * rng.choice([1e-3, 1.0, 1e3])
    return formats.CSR(m, k, ptr, cols.astype(np.uint32), vals)


def padded_device(x, rng):
    """x on the device inside a wider buffer (random leading dimension, sometimes a misaligned start)."""
    r, c = x.shape
    ld = c + int(rng.integers(0, 9))
    shift = int(rng.integers(0, 3))
    buf = torch.zeros(r * ld + 4, dtype=torch.float32, device="cuda")
    view = buf[shift:shift + r * ld].view(r, ld)[:, :c]
    view.copy_(torch.from_numpy(np.ascontiguousarray(x)).cuda())
    return view


def padded_out(r, c, rng):
    ld = c + int(rng.integers(0, 9))
    shift = int(rng.integers(0, 3))
    buf = torch.full((r * ld + 4,), 7.0, dtype=torch.float32, device="cuda")
    return buf[shift:shift + r * ld].view(r, ld)[:, :c]


@pytest.mark.parametrize("seed", range(24 * SCALE))
def test_fuzz_csr_coo_ell(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    m, k, n = int(rng.integers(1, 300)), int(rng.integers(1, 500)), int(rng.integers(1, 300))
    csr = rand_csr(rng, m, k)
    b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
    bd = padded_device(b, rng)
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    a = ops.DeviceCSR.from_host(csr)
    for kern in (0, 1, 2, 3, 4, 5):
        out = padded_out(m, n, rng)
        ops.spmm_csr(a, bd, out=out, kernel=kern)
        assert np.array_equal(out.cpu().numpy(), ref), f"seed {seed} CSR kernel {kern} M={m} K={k} N={n}"
    coo = formats.csr_to_coo(csr)
    ref32 = oracle.spmm_coo(m, coo.row_idxs, coo.col_idxs, coo.data, b)
    for ws in (True, False):
        assert np.array_equal(ops.spmm_coo(ops.DeviceCOO.from_host(coo), bd, workspace=ws).cpu().numpy(), ref32)
    ellc = formats.csr_to_ell_colmajor(csr)
    assert np.array_equal(oracle.spmm_ell_colmajor(m, ellc.row_idxs, ellc.data, b), ref32)
    out = padded_out(m, n, rng)
    ops.spmm_ell(ops.DeviceELL.from_host(ellc), bd, out=out)
    assert np.array_equal(out.cpu().numpy(), ref32), f"seed {seed} ELL"


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_fuzz_bsr(oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    br = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 32]))
    bc = br if rng.random() < 0.7 else int(rng.choice([1, 2, 3, 4, 7, 8]))
    mb, kb, n = int(rng.integers(1, 12)), int(rng.integers(1, 14)), int(rng.integers(1, 200))
    ptrs, idxs = [0], []
    for _ in range(mb):
        cnt = int(rng.integers(0, kb + 1))
        idxs += list(rng.permutation(kb)[:cnt])              # unsorted block columns, empty block rows
        ptrs.append(len(idxs))
    data = (rng.standard_normal((len(idxs), br, bc)) * (rng.random((len(idxs), br, bc)) < 0.5)).astype(np.float32)
    bsr = formats.BSR(mb * br, kb * bc, data.size, br, bc, np.array(ptrs, np.uint32), np.array(idxs, np.uint32), data)
    b = rng.uniform(-1, 1, size=(kb * bc, n)).astype(np.float32)
    ref = oracle.spmm_bsr(bsr.num_rows, br, bc, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, b)
    out = padded_out(bsr.num_rows, n, rng)
    ops.spmm_bsr(ops.DeviceBSR.from_host(bsr), padded_device(b, rng), out=out, kernel=1)
    assert np.array_equal(out.cpu().numpy(), ref), f"seed {seed} BSR {br}x{bc} Mb={mb} Kb={kb} N={n}"


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_fuzz_bsr_mfma(oracle, seed):
    """Random 16x16 / 32x32 block structures (empty block rows, odd block counts, unsorted block columns,
    N not a multiple of the 64 / 128-column super-tiles) through the fp32 and bf16 MFMA kernels."""
    from mispmm import synth
    rng = np.random.default_rng(3000 + seed)
    bd = int(rng.choice([16, 32]))
    mb, kb = int(rng.integers(1, 20)), int(rng.integers(1, 24))
    n = int(rng.choice([4, 8, 24, 64, 72, 128, 136, 200, 256]))
    ptrs, idxs = [0], []
    for _ in range(mb):
        cnt = int(rng.integers(0, min(kb, 9) + 1))
        idxs += list(rng.permutation(kb)[:cnt])
        ptrs.append(len(idxs))
    data = rng.uniform(-2, 2, size=(len(idxs), bd, bd)).astype(np.float32)
    bsr = formats.BSR(mb * bd, kb * bd, data.size, bd, bd, np.array(ptrs, np.uint32), np.array(idxs, np.uint32), data)
    b = rng.uniform(-1, 1, size=(kb * bd, n)).astype(np.float32)
    a = ops.DeviceBSR.from_host(bsr)
    bdev = torch.from_numpy(b).cuda()
    scale = np.abs(bsr.to_dense()).astype(np.float64) @ np.abs(b).astype(np.float64)
    if bd == 16:
        ref = oracle.spmm_bsr(bsr.num_rows, bd, bd, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, b)
        valu = ops.spmm_bsr(a, bdev, kernel=1, acc="fast").cpu().numpy()
        mfma = ops.spmm_bsr(a, bdev, kernel=2, acc="fast").cpu().numpy()
        # four waves take a block row's blocks round-robin: rows of at most one block are a single fma chain and
        # match the FAST VALU chain bit for bit, every row is within the FAST bound and deterministic
        one_block = np.repeat(np.diff(np.array(ptrs)) <= 1, bd)
        assert np.array_equal(mfma[one_block], valu[one_block]), f"seed {seed}: single-block rows must match the VALU chain"
        assert np.array_equal(mfma, ops.spmm_bsr(a, bdev, kernel=2, acc="fast").cpu().numpy())
        assert np.all(np.abs(mfma.astype(np.float64) - ref) <= 1e-5 * scale + 1e-30)
        assert np.all(np.abs(valu.astype(np.float64) - ref) <= 1e-5 * scale + 1e-30)
    a16 = synth.bf16_round(data.reshape(-1)).reshape(data.shape)
    b16 = synth.bf16_round(b.reshape(-1)).reshape(b.shape)
    ref16 = oracle.spmm_bsr(bsr.num_rows, bd, bd, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
    got = ops.spmm_bsr_bf16(a, ops.f32_to_bf16(a.data), ops.f32_to_bf16(bdev)).cpu().numpy()
    scale16 = np.abs(formats.BSR(bsr.num_rows, bsr.num_cols, bsr.nnz, bd, bd, bsr.block_row_ptrs, bsr.block_col_idxs,
                                 a16).to_dense()).astype(np.float64) @ np.abs(b16).astype(np.float64)
    assert np.all(np.abs(got.astype(np.float64) - ref16) <= 2e-6 * scale16 + 1e-30), f"seed {seed} bf16 {bd}x{bd} N={n}"
