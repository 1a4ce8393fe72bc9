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


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_fuzz_plan_order_and_general_entry_bet(oracle, seed):
    """Round 3's paths on random shapes: the clustered plan order (rows permuted, C rows scattered through rowMap; single,
    batched, strided operands, declined shapes falling back) and the general entry point's bet on uniform rows (taken
    whenever nnz divides by M, right or wrong) must return the oracle's bits."""
    rng = np.random.default_rng(5000 + seed)
    m, k = int(rng.integers(1, 700)), int(rng.integers(1, 900))
    n = int(rng.choice([4, 32, 40, 64, 96, 128, 160, 256]))
    csr = rand_csr(rng, m, k)
    if seed % 3 == 0 and csr.nnz % m:                         # make the entry count divide by M without making the rows uniform
        lens = np.diff(csr.row_ptrs.astype(np.int64))
        grow = (-csr.nnz) % m
        for r in rng.permutation(m):
            add = min(grow, k - int(lens[r]))
            lens[r] += add
            grow -= add
            if grow == 0:
                break
        if grow == 0:
            ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
            cols = np.concatenate([np.sort(rng.choice(k, size=int(x), replace=False)) for x in lens] + [np.zeros(0, np.int64)])
            csr = formats.CSR(m, k, ptr, cols.astype(np.uint32), rng.standard_normal(int(ptr[-1])).astype(np.float32))
    b = rng.uniform(-1, 1, size=(k, n)).astype(np.float32)
    bd = padded_device(b, rng) if seed % 2 else torch.from_numpy(b).cuda()
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    a = ops.DeviceCSR.from_host(csr, plan=True, spans=False)
    out = padded_out(m, n, rng)
    taken = ops._csr_plan(a, [bd], [out], "reference", None)
    if taken:
        assert np.array_equal(out.cpu().numpy(), ref), f"seed {seed} plan order M={m} K={k} N={n}"
        outs = [padded_out(m, n, rng) for _ in range(2)]
        if outs[0].stride(0) == outs[1].stride(0):
            assert ops._csr_plan(a, [bd, bd], outs, "reference", None)
            assert np.array_equal(outs[1].cpu().numpy(), ref), f"seed {seed} batched plan order"
    out = padded_out(m, n, rng)
    ops.spmm_csr(a, bd, out=out, use_hint=False)              # general entry point: bets when nnz % M == 0
    assert np.array_equal(out.cpu().numpy(), ref), f"seed {seed} general entry M={m} K={k} N={n} nnz={csr.nnz}"


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_fuzz_bsrc_slots(oracle, seed):
    """The workgroup-per-block-row bf16 kernel on random block rows: 0 .. 12 steps per row (extra steps behind the 4
    slots), 16 x bc blocks, widths that are and are not multiples of 128, fp32 and bf16 C, against the oracle on
    bf16-rounded operands and against the wave-per-block-row kernel."""
    from mispmm import synth
    rng = np.random.default_rng(7000 + seed)
    mb, bc = int(rng.integers(1, 40)), int(rng.choice([4, 8, 16, 32]))
    kb = int(rng.integers(1, 400 // bc + 2))
    n = int(rng.choice([8, 64, 128, 136, 256]))
    ptrs, idxs, blocks = [0], [], []
    for _ in range(mb):
        cnt = int(rng.choice([0, 1, 2, 3, 5, 8, kb])) if kb > 1 else int(rng.integers(0, 2))
        cnt = min(cnt, kb)
        idxs += list(np.sort(rng.choice(kb, size=cnt, replace=False)))
        for _ in range(cnt):
            blocks.append(np.where(rng.random((16, bc)) < rng.choice([0.05, 0.4, 1.0]), rng.uniform(-2, 2, (16, bc)), 0.0).astype(np.float32))
        ptrs.append(len(idxs))
    data = np.stack(blocks) if blocks else np.zeros((0, 16, bc), np.float32)
    bsr = formats.BSR(mb * 16, kb * bc, int(data.size), 16, bc, np.array(ptrs, np.uint32), np.array(idxs, np.uint32), data)
    b = rng.uniform(-1, 1, size=(kb * bc, n)).astype(np.float32)
    a16 = synth.bf16_round(data.reshape(-1)).reshape(data.shape)
    b16 = synth.bf16_round(b.reshape(-1)).reshape(b.shape)
    ref = oracle.spmm_bsr(mb * 16, 16, bc, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
    dense = np.zeros((mb * 16, kb * bc), np.float64)
    for r in range(mb):
        for q in range(ptrs[r], ptrs[r + 1]):
            dense[r * 16:(r + 1) * 16, idxs[q] * bc:(idxs[q] + 1) * bc] = np.abs(a16[q])
    scale = dense @ np.abs(b16).astype(np.float64)
    bd = ops.f32_to_bf16(torch.from_numpy(b).cuda())
    slots = ops.DeviceBSRCSlots.from_host(bsr)
    got = ops.spmm_bsrc_slots_bf16(slots, bd).cpu().numpy()
    assert np.all(np.abs(got.astype(np.float64) - ref) <= 2e-6 * scale + 1e-30), f"seed {seed} slots Mb={mb} bc={bc} Kb={kb} N={n}"
    old = ops.spmm_bsrc_bf16(ops.DeviceBSRC.from_host(bsr), bd).cpu().numpy()
    assert np.all(np.abs(got.astype(np.float64) - old) <= 4e-6 * scale + 1e-30)
    c16 = ops.bf16_to_f32(ops.spmm_bsrc_slots_bf16(slots, bd, out_bf16=True)).cpu().numpy()
    assert np.all(np.abs(c16 - ref) <= 2 ** -8 * np.abs(ref) + 2e-6 * scale + 1e-30)
