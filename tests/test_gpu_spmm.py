"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Run with `pytest -m gpu` on an MI355X.

Bars (BASELINE.json north_star, mispmm.h accumulate modes):
  acc="reference": BIT-EXACT vs the oracle (= the reference CPU engine's rounding
                   sequence) for CSR, COO, ELL and BSR.
  acc="fast"     : |c - oracle| <= 1e-5 * sum_k |a_k||b_k| per element (fp32 fma chain;
                   relative to the magnitude actually summed, so cancellation to ~0
                   does not make the bound meaningless).
  bf16 BSR       : inputs rounded to bf16 (RNE) before BOTH oracle and kernel, fp32
                   accumulate; |c - oracle| <= 2e-6 * sum|a||b| (accumulation order only).
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from mispmm import capi, datasets, formats, ops, synth  # noqa: E402

pytestmark = pytest.mark.gpu

FAST_RTOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "pytest -m gpu needs a GPU"
    assert os.path.exists(capi.LIB_PATH), "libmispmm.so must be built (no fallback path exists)"
    capi.lib()


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def abs_scale(csr, b):
    a = formats.CSR(csr.num_rows, csr.num_cols, csr.row_ptrs, csr.col_idxs, np.abs(csr.data))
    from oracle import oracle as orc
    return orc.spmm_csr(a.row_ptrs, a.col_idxs, a.data, np.abs(b)).astype(np.float64)


def assert_fast_close(c, ref, scale, rtol=FAST_RTOL):
    err = np.abs(c.astype(np.float64) - ref.astype(np.float64))
    bound = rtol * scale + 1e-37
    assert np.all(err <= bound), f"max err/bound {np.max(err / bound):.3g}"


def random_csr(m, k, row_lens, seed):
    rng = np.random.default_rng(seed)
    row_lens = np.asarray(row_lens, dtype=np.int64)
    ptr = np.concatenate([[0], np.cumsum(row_lens)]).astype(np.uint32)
    cols = np.concatenate([np.sort(rng.choice(k, size=n, replace=False)) for n in row_lens] + [np.zeros(0, np.int64)])
    vals = rng.uniform(-2, 2, size=int(ptr[-1])).astype(np.float32)
    return formats.CSR(m, k, ptr, cols.astype(np.uint32), vals)


def test_row_stream_forced_on_small_and_odd_shapes():
    """row_stream.hpp (the persistent row-walking launch) forced on through the tuning build (MISPMM_STREAM=1): CSR with a
    constant row length 9..16, ELL with padding, ragged workgroups, strided operands, plan order -- bit-exact against the
    oracle in REFERENCE mode, within 1e-5 of sum|a||b| in FAST mode (tests/_row_stream_cases.py, a process of its own)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tune = os.path.join(root, "cuda-optimization-for-spmm_amd", "libmispmm_tune.so")
    assert os.path.exists(tune), "run `make -C cuda-optimization-for-spmm_amd tune`"
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "_row_stream_cases.py")], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, MISPMM_LIB=tune, MISPMM_STREAM="1"))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-2500:])
    assert "row_stream cases ok: 12" in p.stdout


def test_production_library_never_takes_the_persistent_launch():
    """Measured and not adopted (profiles/r4/stream_ab.log): the production library does not carry the row_stream kernels and
    multiplies every shape with the one-row-per-lane-group launch."""
    csr = datasets.load_csr("n4c6-b13")
    a = ops.DeviceCSR.from_host(csr, plan=False)
    for n in (128, 256, 512):
        ops.spmm_csr(a, dev(synth.dense_b(csr.num_cols, n)))
        assert "row_gather" in capi.last_kernel(), (n, capi.last_kernel())


@pytest.mark.parametrize("name,n", [("n4c6-b13", 128), ("n4c6-b13", 512), ("n3c5-b6", 64)])
def test_csr_lds_tiles_match_oracle(oracle, name, n):
    """mispmm_csr_lds_tile_f32 (the north_star's kernel shape: the B rows of a group of rows staged once in LDS): tiles built
    by mispmm_csr_tiles_host (<= 16 rows, <= 128 distinct columns, every row in exactly one tile, an entry's slot names its
    column), product bit-exact in REFERENCE mode (a row keeps its entries in storage order) and within 1e-5 in FAST mode;
    strided operands keep their gap columns."""
    csr = datasets.load_csr(name)
    a = ops.DeviceCSRTiles.from_host(csr)
    trp, tcp = a.tile_row_ptrs.cpu().numpy().view(np.uint32), a.tile_col_ptrs.cpu().numpy().view(np.uint32)
    order, cols, slots = a.row_map.cpu().numpy().view(np.uint32), a.tile_cols.cpu().numpy().view(np.uint32), a.slots.cpu().numpy()
    assert sorted(order.tolist()) == list(range(csr.num_rows)) and trp[-1] == csr.num_rows and tcp[-1] == a.num_listed
    assert np.all(np.diff(trp.astype(np.int64)) <= 16) and np.all(np.diff(trp.astype(np.int64)) >= 1) and np.all(np.diff(tcp.astype(np.int64)) <= 128)
    w = a.row_nnz
    for t in (0, a.num_tiles // 2, a.num_tiles - 1):          # an entry's slot names its column in its tile's list
        for i in range(trp[t], trp[t + 1]):
            r = order[i]
            assert np.array_equal(cols[tcp[t]:tcp[t + 1]][slots[i * w:(i + 1) * w]], csr.col_idxs[csr.row_ptrs[r]:csr.row_ptrs[r + 1]])
        assert len(set(cols[tcp[t]:tcp[t + 1]].tolist())) == tcp[t + 1] - tcp[t]
    assert a.num_listed < csr.nnz                               # rows of a tile do share B rows
    b = synth.dense_b(csr.num_cols, n)
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    got = ops.spmm_csr_tiles(a, dev(b)).cpu().numpy()
    assert "csr_lds_tile" in capi.last_kernel()
    assert np.array_equal(got, ref)
    assert_fast_close(ops.spmm_csr_tiles(a, dev(b), acc="fast").cpu().numpy(), ref, abs_scale(csr, b))
    bw = torch.full((csr.num_cols, n + 8), 3.0, device="cuda")
    bw[:, :n] = dev(b)
    cw = torch.full((csr.num_rows, n + 4), -7.0, device="cuda")
    ops.spmm_csr_tiles(a, bw[:, :n], out=cw[:, :n])
    assert np.array_equal(cw[:, :n].cpu().numpy(), ref) and bool((cw[:, n:] == -7.0).all())
    for bad in (40, 96):                                        # column parts that are not whole 64-column groups are declined
        with pytest.raises(capi.MispmmError):
            ops.spmm_csr_tiles(a, dev(synth.dense_b(csr.num_cols, bad)))
    with pytest.raises(ValueError):
        ops.DeviceCSRTiles.from_host(datasets.load_csr("delaunay_n12"))


# --------------------------------------------------------------------------- CSR
CSR_CASES = [("Hamrle1", [1, 3, 32]), ("n3c5-b6", [8, 21]), ("qh1484", [64, 130]), ("delaunay_n12", [128]),
             ("GL7d25", [64, 100]), ("ACTIVSg10K", [128]), ("n4c6-b13", [128, 256, 512, 515])]


@pytest.mark.parametrize("kernel", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("name,ns", CSR_CASES)
def test_csr_matches_oracle(oracle, name, ns, kernel):
    csr = datasets.load_csr(name)
    a = ops.DeviceCSR.from_host(csr)
    for n in ns:
        b = synth.dense_b(csr.num_cols, n)
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        bd = dev(b)
        c = ops.spmm_csr(a, bd, kernel=kernel, acc="reference").cpu().numpy()
        assert np.array_equal(c, ref), f"{name} N={n} k{kernel}: reference mode must be bit-exact"
        cf = ops.spmm_csr(a, bd, kernel=kernel, acc="fast").cpu().numpy()
        assert_fast_close(cf, ref, abs_scale(csr, b))


@pytest.mark.parametrize("width", [1, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 23])
def test_uniform_rows_every_slot_count(oracle, width):
    """mispmm_csr_uniform_f32 and ELL pick a slot count (10 / 12 / 14 / 16) from the row length; N = 128 runs 16-lane
    groups, N = 256 8-lane groups, N = 100 / 20 the generic shapes.  Bit-exact against the oracle every time, and the
    ELL of the same rows with padding holes."""
    m, k = 333, 1200
    csr = random_csr(m, k, [width] * m, seed=100 + width)
    a = ops.DeviceCSR.from_host(csr)
    assert a.uniform_row_nnz == width
    rng = np.random.default_rng(width)
    for n in (128, 256, 100, 20):
        b = synth.dense_b(k, n)
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        assert np.array_equal(ops.spmm_csr(a, dev(b), acc="reference").cpu().numpy(), ref), (width, n)
        assert np.array_equal(ops.spmm_csr(a, dev(b), acc="reference", use_hint=False).cpu().numpy(), ref), (width, n)
        assert_fast_close(ops.spmm_csr(a, dev(b), acc="fast").cpu().numpy(), ref, abs_scale(csr, b))
        # ELL of that width with some entries turned into padding (anywhere in the row)
        cols = csr.col_idxs.reshape(m, width).copy()
        vals = csr.data.reshape(m, width).copy()
        hole = rng.random((m, width)) < 0.2
        cols[hole] = formats.ELL_PAD
        vals[hole] = 0.0
        ell = formats.ELLRowMajor(m, k, int((~hole).sum()), width, cols, vals)
        ref_ell = np.zeros((m, n), np.float32)
        for s_ in range(width):             # fp32 += in slot order, unfused, as spmmELLCpu sums a row
            live = ~hole[:, s_]
            ref_ell[live] = ref_ell[live] + vals[live, s_, None] * b[cols[live, s_]]
        c = ops.spmm_ell(ops.DeviceELL.from_host(ell), dev(b)).cpu().numpy()
        assert np.array_equal(c, ref_ell), (width, n)


@pytest.mark.parametrize("n", [1, 64, 130, 256, 512])
def test_csr_long_row_kernel(oracle, n):
    """Mean row length >= 24 sends kernel 0/5 to the split kernel (16-byte B rows, N < 384), else to the deep
    wave-per-row kernel (csr_wave_deep; the lane-group kernel from N = 384 on): rows of 0, 1, 7, 8, 9 entries, rows at and
    around the 512-entry LDS phase, a 1500-entry row; every vector width; bit-exact."""
    lens = [0, 1, 7, 8, 9, 511, 512, 513, 1500, 0, 64, 33, 40, 25, 31, 200]
    csr = random_csr(len(lens), 2000, lens, seed=11)
    assert csr.nnz // csr.num_rows >= 24
    a = ops.DeviceCSR.from_host(csr)
    b = synth.dense_b(csr.num_cols, n)
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    for kernel in (0, 5):
        c = ops.spmm_csr(a, dev(b), kernel=kernel, acc="reference").cpu().numpy()
        assert np.array_equal(c, ref)
        assert_fast_close(ops.spmm_csr(a, dev(b), kernel=kernel, acc="fast").cpu().numpy(), ref, abs_scale(csr, b))
    # a strided C / B (ldb, ldc > N) through the same kernel
    ld = n + 12
    bw = torch.zeros((csr.num_cols, ld), dtype=torch.float32, device="cuda")
    bw[:, :n] = dev(b)
    cw = torch.full((csr.num_rows, ld), -7.0, dtype=torch.float32, device="cuda")
    ops.spmm_csr(a, bw[:, :n], out=cw[:, :n], acc="reference")
    assert np.array_equal(cw[:, :n].cpu().numpy(), ref)
    assert torch.all(cw[:, n:] == -7.0)


@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4, 5, 6])
def test_csr_ragged_rows_and_edges(oracle, kernel):
    """Empty rows, rows of 1, 17, 64, 65, 200 and 2500 entries (k2 re-stages LDS past 1024
    pairs, k3/k4 loop 64-pair chunks), M not a multiple of any tile."""
    lens = [0, 1, 17, 64, 65, 200, 0, 2500, 3, 0, 16, 15, 33, 8, 4, 5, 9, 0, 0, 1300, 7]
    csr = random_csr(len(lens), 3000, lens, seed=7)
    a = ops.DeviceCSR.from_host(csr)
    for n in (1, 2, 5, 64, 72, 128, 260):
        b = synth.dense_b(csr.num_cols, n, seed=n)
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        c = ops.spmm_csr(a, dev(b), kernel=kernel, acc="reference").cpu().numpy()
        assert np.array_equal(c, ref), f"N={n}"


@pytest.mark.parametrize("kernel", [1, 2, 3, 4, 5, 6])
def test_csr_strided_and_unaligned_operands(oracle, kernel):
    csr = datasets.load_csr("qh1484")
    a = ops.DeviceCSR.from_host(csr)
    n = 96
    b = synth.dense_b(csr.num_cols, n)
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    for ldb, ldc, shift in ((n, n, 0), (n + 32, n + 64, 0), (n + 1, n + 3, 0), (n + 4, n + 4, 1), (n + 2, n + 2, 2)):
        big_b = torch.zeros(csr.num_cols * ldb + 8, dtype=torch.float32, device="cuda")
        bview = big_b[shift:shift + csr.num_cols * ldb].view(csr.num_cols, ldb)[:, :n]
        bview.copy_(dev(b))
        big_c = torch.full((csr.num_rows * ldc + 8,), -7.0, dtype=torch.float32, device="cuda")
        cview = big_c[shift:shift + csr.num_rows * ldc].view(csr.num_rows, ldc)[:, :n]
        ops.spmm_csr(a, bview, out=cview, kernel=kernel)
        assert np.array_equal(cview.cpu().numpy(), ref), (ldb, ldc, shift)
        full = big_c[shift:shift + csr.num_rows * ldc].view(csr.num_rows, ldc).cpu().numpy()
        assert np.all(full[:, n:] == -7.0), "columns past N must not be written"


@pytest.mark.parametrize("name,n", [("n4c6-b13", 128), ("ch7-6-b5", 64), ("n4c6-b13", 515), ("qh1484", 32)])
def test_csr_uniform_row_hint_equals_general_path(oracle, name, n):
    """mispmm_csr_uniform_f32 (row pointers never read) vs mispmm_csr_f32 vs the oracle: same bits.  The hint is
    only taken for matrices whose rows really all have the same length (qh1484 does not: hint 0)."""
    csr = datasets.load_csr(name)
    a = ops.DeviceCSR.from_host(csr)
    lens = np.diff(csr.row_ptrs.astype(np.int64))
    assert a.uniform_row_nnz == (int(lens[0]) if np.all(lens == lens[0]) else 0)
    assert (a.uniform_row_nnz > 0) == (name != "qh1484")
    b = synth.dense_b(csr.num_cols, n)
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    for acc in ("reference", "fast"):
        hinted = ops.spmm_csr(a, dev(b), acc=acc).cpu().numpy()
        general = ops.spmm_csr(a, dev(b), acc=acc, use_hint=False).cpu().numpy()
        assert np.array_equal(hinted, general)
        if acc == "reference":
            assert np.array_equal(hinted, ref)
    # a poisoned row pointer array proves the hinted call does not read it
    if a.uniform_row_nnz:
        poisoned = ops.DeviceCSR(a.num_rows, a.num_cols, a.nnz, torch.full_like(a.row_ptrs, 0x7FFFFFFF), a.col_idxs, a.data,
                                 a.uniform_row_nnz)
        assert np.array_equal(ops.spmm_csr(poisoned, dev(b)).cpu().numpy(), ref)


def test_csr_empty_and_overwrite():
    empty = formats.CSR(5, 9, np.zeros(6, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    c = torch.full((5, 16), 3.0, device="cuda")
    ops.spmm_csr(ops.DeviceCSR.from_host(empty), torch.ones(9, 16, device="cuda"), out=c)
    assert torch.all(c == 0), "beta = 0: C is overwritten even for an all-zero A"
    none = formats.CSR(0, 9, np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    assert ops.spmm_csr(ops.DeviceCSR.from_host(none), torch.ones(9, 4, device="cuda")).shape == (0, 4)


@pytest.mark.parametrize("n", [4, 60, 64, 128, 200, 256, 512])
def test_csr_split_row_kernel(oracle, n):
    """Kernel 6 (one wave per row x 32 columns, entries dealt over its 8 lane groups; csr_split.hpp) and kernel 0 on long
    rows: REFERENCE mode is bit-exact with the sequential engine whether a wave keeps the split sum (uniform B), or sums
    again in entry order because the re-association test fails -- products spread over 2^60, a NaN, an Inf, values near
    the fp32 limits -- and for rows of 0, 1, 7..9 entries, around the 512-entry LDS phase, 2500 entries."""
    lens = [0, 1, 7, 8, 9, 31, 32, 33, 511, 512, 513, 1025, 2500, 0, 64, 40, 25, 200, 422]
    csr = random_csr(len(lens), 3000, lens, seed=17)
    a = ops.DeviceCSR.from_host(csr, spans=False)      # the stateless entry point: rows in order
    b = synth.dense_b(csr.num_cols, n)
    rng = np.random.default_rng(5)
    wide = b * np.exp2(rng.integers(-30, 31, size=b.shape)).astype(np.float32)   # every long row fails the test
    sparse_b = np.where(rng.random(b.shape) < 0.5, np.float32(0), b)              # zero products are left out of it
    few = b.copy()                                                                # a few elements per row fail it
    few[rng.integers(0, b.shape[0], 40), rng.integers(0, n, 40)] *= np.float32(2.0 ** -40)
    special = b.copy()
    special[csr.col_idxs[40], 0] = np.inf
    special[csr.col_idxs[2000], n - 1] = np.nan
    special[csr.col_idxs[3000], n // 2] = np.float32(3e38)
    special[csr.col_idxs[3001], n // 2] = np.float32(-3e38)
    special[csr.col_idxs[100], 1] = np.float32(1e-44)                               # a denormal
    for name, bb in (("uniform", b), ("wide", wide), ("zeros", sparse_b), ("few", few), ("special", special)):
        with np.errstate(all="ignore"):
            ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, bb)
        for kernel in (6, 0):
            c = ops.spmm_csr(a, dev(bb), kernel=kernel, acc="reference").cpu().numpy()
            assert np.array_equal(c, ref, equal_nan=True), (name, kernel, n)
        # the same through the span list: rows of more than 128 (30) entries as 4 chunks on the waves of one workgroup
        for share in (0, 30):
            sp = ops.DeviceCSR.from_host(csr, spans=True, share_len=share)
            c = ops.spmm_csr(sp, dev(bb), kernel=6, acc="reference").cpu().numpy()
            assert np.array_equal(c, ref, equal_nan=True), (name, "spans", share, n)
            assert "longest-first" in capi.last_kernel()
    ops.spmm_csr(a, dev(b), kernel=0)
    assert ("csr_split" in capi.last_kernel()) == (n < 384)   # REFERENCE mode, kernel 0: the lane-group kernel from 384 on
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    fast = ops.spmm_csr(a, dev(b), kernel=6, acc="fast").cpu().numpy()
    assert_fast_close(fast, ref, abs_scale(csr, b))
    assert np.array_equal(fast, ops.spmm_csr(a, dev(b), kernel=6, acc="fast").cpu().numpy())   # fixed-order reduction
    # a strided C / B (ldb, ldc > N)
    ld = n + 12
    bw = torch.zeros((csr.num_cols, ld), dtype=torch.float32, device="cuda")
    bw[:, :n] = dev(b)
    cw = torch.full((csr.num_rows, ld), -7.0, dtype=torch.float32, device="cuda")
    ops.spmm_csr(a, bw[:, :n], out=cw[:, :n], kernel=6)
    assert np.array_equal(cw[:, :n].cpu().numpy(), ref)
    assert torch.all(cw[:, n:] == -7.0)


def test_csr_split_row_kernel_on_real_matrices(oracle):
    """GL7d25 (mean 29, longest 422 entries; integer coefficients), tols4000 (mean 2, longest 90) and the headline
    matrix through kernel 6, in row order (mispmm_csr_f32) and with the rows longest first (mispmm_csr_split_f32 + the
    spans built at upload): the order decides when a row runs, never what it returns."""
    for name in ("GL7d25", "tols4000", "n4c6-b13"):
        csr = datasets.load_csr(name)
        a = ops.DeviceCSR.from_host(csr, spans=True)
        auto = ops.DeviceCSR.from_host(csr)
        # built unasked for long rows (GL7d25: for the split kernel and the two-body launch) and for short rows with a few
        # long ones (tols4000, longest 90: for the two-body launch only)
        assert (auto.spans is not None, auto.spans_hybrid_only) == {"GL7d25": (True, False), "tols4000": (True, True), "n4c6-b13": (False, False)}[name]
        if name == "tols4000":
            b = synth.dense_b(csr.num_cols, 128)
            ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
            assert np.array_equal(ops.spmm_csr(auto, dev(b)).cpu().numpy(), ref) and "csr_hybrid" in capi.last_kernel()
            b = synth.dense_b(csr.num_cols, 96)              # no two-body launch for this width: the row-gather kernel, not the split kernel
            ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
            assert np.array_equal(ops.spmm_csr(auto, dev(b)).cpu().numpy(), ref) and "row_gather" in capi.last_kernel()
        for n in (32, 128, 256, 516):
            b = synth.dense_b(csr.num_cols, n)
            ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
            assert np.array_equal(ops.spmm_csr(a, dev(b), kernel=6, use_hint=False).cpu().numpy(), ref), (name, n)
            assert "csr_split" in capi.last_kernel() and "longest-first" not in capi.last_kernel()
            assert np.array_equal(ops.spmm_csr(a, dev(b), kernel=6).cpu().numpy(), ref), (name, n)
            assert "longest-first" in capi.last_kernel()
            fast = ops.spmm_csr(a, dev(b), kernel=6, acc="fast").cpu().numpy()
            assert_fast_close(fast, ref, abs_scale(csr, b))
        # rows that are not 16-byte vectors: the entry point declines, the general one takes over
        b = synth.dense_b(csr.num_cols, 30)
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        assert np.array_equal(ops.spmm_csr(a, dev(b), kernel=6).cpu().numpy(), ref)
        assert "csr_split" not in capi.last_kernel()
    # kernel 0 on a long-row matrix takes the same path unasked
    csr = datasets.load_csr("GL7d25")
    a = ops.DeviceCSR.from_host(csr)
    b = synth.dense_b(csr.num_cols, 128)
    assert np.array_equal(ops.spmm_csr(a, dev(b)).cpu().numpy(), oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b))
    assert "csr_hybrid" in capi.last_kernel()          # ... its long rows by the split body, the others by the row-gather body


def test_csr_two_body_launch_for_long_row_matrices(oracle):
    """mispmm_csr_hybrid_f32 (what kernel 0 takes on a matrix that carries a span list): the rows of more than 32 entries
    through the split kernel's body, the others through the row-gather body of the SAME launch, both reading the one span
    list.  REFERENCE mode returns the oracle's bits (and so the split kernel's), FAST stays within its bound; shapes
    without such a launch (more than 256 columns, column parts that are not one lane group wide) take the split kernel."""
    csr = datasets.load_csr("GL7d25")
    a = ops.DeviceCSR.from_host(csr)
    lens = np.diff(csr.row_ptrs.astype(np.int64))
    spans = a.spans.cpu().numpy().view(np.uint32).reshape(-1, 4)
    span_len = (spans[:, 2] - spans[:, 1]).astype(np.int64)
    assert a.long_spans % 4 == 0 and np.all(span_len[a.long_spans:] <= 32) and np.all(spans[a.long_spans:, 3] == 0)
    assert a.long_spans - 4 < int((spans[:, 3] == 1).sum()) + int((lens > 32).sum() - (lens > 128).sum()) <= a.long_spans
    for n, hybrid in ((8, False), (32, True), (64, True), (96, False), (128, True), (256, True), (516, False)):
        b = synth.dense_b(csr.num_cols, n)
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        got = ops.spmm_csr(a, dev(b))
        assert ("csr_hybrid" in capi.last_kernel()) == hybrid, (n, capi.last_kernel())
        assert np.array_equal(got.cpu().numpy(), ref), n
        assert torch.equal(got, ops.spmm_csr(a, dev(b), kernel=6))             # the split kernel on the whole list: same bits
        assert "csr_hybrid" not in capi.last_kernel()
        fast = ops.spmm_csr(a, dev(b), acc="fast").cpu().numpy()
        assert_fast_close(fast, ref, abs_scale(csr, b))
        assert np.array_equal(fast, ops.spmm_csr(a, dev(b), acc="fast").cpu().numpy())   # deterministic
    # values that make long rows fail the exactness test, an Inf and a NaN in B; a strided B and C whose gap survives
    n, ld = 128, 140
    rng = np.random.default_rng(3)
    b = synth.dense_b(csr.num_cols, n)
    b = (b * np.exp2(rng.integers(-30, 31, size=b.shape))).astype(np.float32)
    b[csr.col_idxs[5], 0] = np.inf
    b[csr.col_idxs[70000], 77] = np.nan
    with np.errstate(all="ignore"):
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    bw = torch.zeros((csr.num_cols, ld), dtype=torch.float32, device="cuda")
    bw[:, :n] = dev(b)
    cw = torch.full((csr.num_rows, ld), -7.0, dtype=torch.float32, device="cuda")
    ops.spmm_csr(a, bw[:, :n], out=cw[:, :n])
    assert "csr_hybrid" in capi.last_kernel()
    assert np.array_equal(cw[:, :n].cpu().numpy(), ref, equal_nan=True) and torch.all(cw[:, n:] == -7.0)
    # the entry point's own argument checks
    l = capi.lib()
    args = lambda n_long, n_spans=spans.shape[0]: (None, csr.num_rows, csr.num_cols, csr.nnz, a.col_idxs.data_ptr(), a.data.data_ptr(),  # noqa: E731
                                                   a.spans.data_ptr(), n_spans, n_long, bw.data_ptr(), n, ld, cw.data_ptr(), ld, 0)
    assert l.mispmm_csr_hybrid_f32(*args(a.long_spans + 2)) == capi.ERR_INVALID_ARG        # not a multiple of 4
    assert l.mispmm_csr_hybrid_f32(*args(4)) == capi.ERR_INVALID_ARG                       # cuts through the chunk groups
    assert l.mispmm_csr_hybrid_f32(*args(a.long_spans, spans.shape[0] + 1)) == capi.ERR_INVALID_ARG
    assert l.mispmm_csr_hybrid_f32(*args(spans.shape[0] - spans.shape[0] % 4)) in (capi.ERR_UNSUPPORTED, 0)


@pytest.mark.parametrize("seed", range(6))
def test_csr_split_row_kernel_random_shapes(oracle, seed):
    """Seeded sweep of kernel 6, in row order and longest first: row-length distributions (constant, geometric, a few
    giants among short rows, sorted either way, empty stretches), matrix and dense widths, value distributions of A and B
    (narrow, wide, sparse B, integers) -- REFERENCE mode must return the oracle's bits every time, FAST within its bound."""
    rng = np.random.default_rng(1000 + seed)
    hybrids = 0
    for case in range(8):
        m = int(rng.integers(1, 700))
        k = int(rng.integers(8, 3000))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            lens = np.full(m, int(rng.integers(0, min(k, 70))))
        elif kind == 1:
            lens = np.minimum(rng.geometric(1.0 / rng.integers(2, 60), m), k)
        elif kind == 2:
            lens = rng.integers(0, 6, m)
            lens[rng.integers(0, m, max(1, m // 40))] = rng.integers(min(k, 100), min(k, 1200) + 1, max(1, m // 40))
        elif kind == 3:
            lens = np.sort(np.minimum(rng.geometric(1.0 / 30, m), k))[:: int(rng.choice([-1, 1]))]
        else:
            lens = np.where(rng.random(m) < 0.5, 0, rng.integers(1, min(k, 300) + 1, m))
        lens = np.minimum(lens, k)
        csr = random_csr(m, k, lens, seed=int(rng.integers(1 << 30)))
        vals = csr.data
        a_kind = int(rng.integers(0, 3))
        if a_kind == 1:
            vals = (vals * np.exp2(rng.integers(-20, 21, vals.shape))).astype(np.float32)
        elif a_kind == 2:
            vals = rng.integers(-24, 25, vals.shape).astype(np.float32)
        csr = formats.CSR(m, k, csr.row_ptrs, csr.col_idxs, vals)
        n = int(rng.choice([4, 8, 28, 32, 36, 64, 100, 128, 132, 256, 260, 384, 512]))
        b = synth.dense_b(k, n, seed=int(rng.integers(1 << 20)))
        b_kind = int(rng.integers(0, 4))
        if b_kind == 1:
            b = (b * np.exp2(rng.integers(-30, 31, b.shape))).astype(np.float32)
        elif b_kind == 2:
            b = np.where(rng.random(b.shape) < 0.7, np.float32(0), b)
        elif b_kind == 3:
            b = synth.dense_b(k, n, mode="exact")
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        share = int(rng.choice([0, 8, 40, 300]))       # rows longer than this go to the 4 waves of a workgroup as chunks
        a = ops.DeviceCSR.from_host(csr, spans=True, share_len=share)
        what = (seed, case, m, k, kind, a_kind, n, b_kind, share)
        assert np.array_equal(ops.spmm_csr(a, dev(b), kernel=6, use_hint=False).cpu().numpy(), ref), what
        assert np.array_equal(ops.spmm_csr(a, dev(b), kernel=6).cpu().numpy(), ref), what
        assert "longest-first" in capi.last_kernel()
        # kernel 0: the two-body launch where the shape has one (the list's long rows split, its short rows by lane groups)
        assert np.array_equal(ops.spmm_csr(a, dev(b)).cpu().numpy(), ref), what
        hybrids += "csr_hybrid" in capi.last_kernel()      # (uniform rows, 384 columns or more, odd column parts go elsewhere)
        with np.errstate(all="ignore"):
            assert_fast_close(ops.spmm_csr(a, dev(b), kernel=6, acc="fast").cpu().numpy(), ref, abs_scale(csr, b))
            assert_fast_close(ops.spmm_csr(a, dev(b), acc="fast").cpu().numpy(), ref, abs_scale(csr, b))
    print(f"seed {seed}: {hybrids} of 8 cases took the two-body launch")


def test_csr_split_row_kernel_takes_the_ordered_sum_only_where_needed(oracle, tmp_path):
    """The measurement build counts the waves of kernel 6 that summed their row again in entry order: none on grid
    data, a handful on uniform data (products 2^19.. apart in rows of hundreds of entries), every wave with entries
    when B spreads over 2^60 -- and the results are the reference's bits in all three cases."""
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-optimization-for-spmm_amd")
    tune = os.path.join(pkg, "libmispmm_tune.so")
    assert os.path.exists(tune), "run `make -C cuda-optimization-for-spmm_amd tune`"
    code = (
        "import ctypes, sys, numpy as np, torch\n"
        f"sys.path.insert(0, {pkg!r})\n"
        "from mispmm import capi, datasets, ops, synth\n"
        "l = capi.lib()\n"
        "def resummed():\n"
        "    out = (ctypes.c_ulonglong * 2)()\n"
        "    capi.check(l.mispmm_debug_split_stats(out))\n"
        "    return int(out[0])\n"
        "csr = datasets.load_csr('GL7d25')\n"
        "a = ops.DeviceCSR.from_host(csr)\n"
        "uni = synth.dense_b(csr.num_cols, 128)\n"
        "wide = uni * np.exp2(np.random.default_rng(1).integers(-30, 31, size=uni.shape)).astype(np.float32)\n"
        "res = {}\n"
        "for tag, b in (('grid', synth.dense_b(csr.num_cols, 128, mode='exact')), ('uniform', uni), ('wide', wide)):\n"
        "    resummed()\n"
        "    res[tag] = ops.spmm_csr(a, torch.from_numpy(b).cuda(), kernel=6).cpu().numpy()\n"
        "    res[tag + '_resummed'] = np.array(resummed())\n"
        "# N % 32 != 0: the lanes past N of a wave's last 32-column part must drop their reads of padding entries too\n"
        "# (a read of B[0] there would feed its Inf into unused partial sums and make every such wave sum again)\n"
        "from mispmm import formats\n"
        "rng = np.random.default_rng(3)\n"
        "m, k, w, n = 200, 500, 41, 136\n"
        "cols = np.concatenate([np.sort(rng.choice(np.arange(1, k), size=w, replace=False)) for _ in range(m)]).astype(np.uint32)\n"
        "syn = formats.CSR(m, k, (np.arange(m + 1) * w).astype(np.uint32), cols, rng.choice([-1.0, 1.0], size=m * w).astype(np.float32))\n"
        "bb = synth.dense_b(k, n, mode='exact')\n"
        "bb[0, 0:4] = np.inf\n"
        "resummed()\n"
        "res['inf_row0'] = ops.spmm_csr(ops.DeviceCSR.from_host(syn, spans=False), torch.from_numpy(bb).cuda(), kernel=6).cpu().numpy()\n"
        "res['inf_row0_resummed'] = np.array(resummed())\n"
        "res['inf_row0_ptr'], res['inf_row0_cols'], res['inf_row0_vals'], res['inf_row0_b'] = syn.row_ptrs, syn.col_idxs, syn.data, bb\n"
        "np.savez(sys.argv[1], tag=np.array(capi.last_kernel()), **res)\n")
    path = str(tmp_path / "split.npz")
    p = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, MISPMM_LIB=tune), capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = np.load(path)
    assert "csr_split" in str(res["tag"])
    csr = datasets.load_csr("GL7d25")
    uni = synth.dense_b(csr.num_cols, 128)
    wide = uni * np.exp2(np.random.default_rng(1).integers(-30, 31, size=uni.shape)).astype(np.float32)
    for tag, b in (("grid", synth.dense_b(csr.num_cols, 128, mode="exact")), ("uniform", uni), ("wide", wide)):
        assert np.array_equal(res[tag], oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)), tag
    waves = int((np.diff(csr.row_ptrs.astype(np.int64)) > 0).sum()) * 4
    assert int(res["grid_resummed"]) == 0
    assert 0 < int(res["uniform_resummed"]) <= 40
    assert int(res["wide_resummed"]) >= 0.9 * waves
    want = oracle.spmm_csr(res["inf_row0_ptr"], res["inf_row0_cols"], res["inf_row0_vals"], res["inf_row0_b"])
    assert np.all(np.isfinite(want)) and np.array_equal(res["inf_row0"], want)
    assert int(res["inf_row0_resummed"]) == 0


def test_csr_nonfinite_values_follow_the_reference(oracle):
    csr = random_csr(4, 40, [3, 14, 16, 20], seed=3)
    b = synth.dense_b(40, 64)
    b[csr.col_idxs[5], 3] = np.inf
    b[csr.col_idxs[20], 7] = np.nan
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    a = ops.DeviceCSR.from_host(csr)
    for k in (1, 2, 3, 4, 5, 6):
        c = ops.spmm_csr(a, dev(b), kernel=k).cpu().numpy()
        assert np.array_equal(c, ref, equal_nan=True)


def test_exact_grid_inputs_make_every_kernel_and_format_agree_bitwise(oracle):
    """Headline matrix (values +-1) times a 2^-8-grid B: all partial sums are exact in fp32, so
    every kernel, accumulate mode and format must return identical bits -- a full-size,
    order-independent check."""
    csr = datasets.load_csr("n4c6-b13")
    b = synth.dense_b(csr.num_cols, 128, mode="exact")
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    bd = dev(b)
    a = ops.DeviceCSR.from_host(csr)
    for k in (1, 2, 3, 4, 5, 6):
        for acc in ("reference", "fast"):
            assert np.array_equal(ops.spmm_csr(a, bd, kernel=k, acc=acc).cpu().numpy(), ref)
    ell = ops.DeviceELL.from_host(formats.csr_to_ell_colmajor(csr))
    coo = ops.DeviceCOO.from_host(formats.csr_to_coo(csr))
    for acc in ("reference", "fast"):
        assert np.array_equal(ops.spmm_ell(ell, bd, acc=acc).cpu().numpy(), ref)
        assert np.array_equal(ops.spmm_coo(coo, bd, acc=acc).cpu().numpy(), ref)


def test_golden_fixtures_through_the_gpu(golden_dir):
    """The reference's committed result.expect files and the reduced expectations of the large
    configs (tests/golden/make_golden.py), straight through the HIP path."""
    from test_oracle import LARGE, check_large
    for d, stem in (("small_10x10", "sparse"), ("small_32x32", "Hamrle1")):
        csr = formats.read_csr(os.path.join(golden_dir, d, stem + ".csr"))
        b = formats.read_dense(os.path.join(golden_dir, d, "dense.in")).data
        expect = np.loadtxt(os.path.join(golden_dir, d, "result.expect"), ndmin=2)
        c = ops.spmm_csr(ops.DeviceCSR.from_host(csr), dev(b)).cpu().numpy()
        scale = np.abs(csr.to_dense()).astype(np.float64) @ np.abs(b).astype(np.float64)
        assert np.all(np.abs(c - expect) <= 2e-6 * scale + 1e-9)
    z = np.load(os.path.join(golden_dir, "expected_large.npz"))
    for tag, name, k, mode in LARGE:
        csr = datasets.load_csr(name)
        b = dev(synth.dense_b(csr.num_cols, k, mode=mode))
        for acc in ("reference", "fast"):
            check_large(ops.spmm_csr(ops.DeviceCSR.from_host(csr), b, acc=acc).cpu().numpy(), z, tag)


def test_graph_replay_equals_eager(oracle):
    import ctypes
    csr = datasets.load_csr("n4c6-b13")
    a = ops.DeviceCSR.from_host(csr)
    b = dev(synth.dense_b(csr.num_cols, 128))
    eager = ops.spmm_csr(a, b, kernel=3).cpu().numpy()
    l = capi.lib()
    s = torch.cuda.Stream()
    c = torch.zeros((csr.num_rows, 128), device="cuda")
    torch.cuda.synchronize()
    sp = ctypes.c_void_p(s.cuda_stream)
    capi.check(l.mispmm_graph_begin(sp))
    for _ in range(5):
        ops.spmm_csr(a, b, out=c, kernel=3, stream=s)
    g = ctypes.c_void_p()
    capi.check(l.mispmm_graph_end(sp, ctypes.byref(g)))
    assert torch.all(c == 0), "capture must not execute"
    capi.check(l.mispmm_graph_launch(g, sp))
    capi.check(l.mispmm_stream_sync(sp))
    assert np.array_equal(c.cpu().numpy(), eager)
    capi.check(l.mispmm_graph_destroy(g))


@pytest.mark.parametrize("name,n,count", [("n4c6-b13", 128, 5), ("n4c6-b13", 128, 19), ("qh1484", 64, 3), ("delaunay_n12", 128, 16),
                                          ("GL7d25", 64, 2), ("qh1484", 130, 3)])
def test_csr_batched_launch_equals_single_launches(oracle, name, n, count):
    """mispmm_csr_batch_f32: several dense operands, one launch (per 16) -- bit for bit what one call per operand gives,
    for the uniform-row and the general CSR path, for more than 16 operands, and for shapes the batched kernel does not
    take (long rows, a width that is not a multiple of 4: issued one by one)."""
    csr = datasets.load_csr(name)
    a = ops.DeviceCSR.from_host(csr)
    bs = [dev(synth.dense_b(csr.num_cols, n, seed=100 + i)) for i in range(count)]
    for acc in ("reference", "fast"):
        outs = ops.spmm_csr_batch(a, bs, acc=acc)
        torch.cuda.synchronize()
        for i, b in enumerate(bs):
            single = ops.spmm_csr(a, b, acc=acc)
            assert torch.equal(outs[i], single), (name, acc, i)
        if acc == "reference":
            ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, bs[-1].cpu().numpy())
            assert np.array_equal(outs[-1].cpu().numpy(), ref)


# --------------------------------------------------------------------------- ELL / COO
@pytest.mark.parametrize("name,ns", [("Hamrle1", [3, 32]), ("n3c5-b6", [21]), ("delaunay_n12", [128, 130]),
                                     ("n4c6-b13", [256]), ("ACTIVSg10K", [64])])
def test_ell_matches_oracle(oracle, name, ns):
    csr = datasets.load_csr(name)
    for ref_width in (True, False):
        if ref_width and np.diff(csr.row_ptrs.astype(np.int64)).max() < np.bincount(csr.col_idxs).max():
            continue   # the reference-sized padding cannot hold this matrix's longest column
        ellc = formats.csr_to_ell_colmajor(csr, reference_width=ref_width)
        a = ops.DeviceELL.from_host(ellc)
        for n in ns:
            b = synth.dense_b(csr.num_cols, n)
            ref = oracle.spmm_ell_colmajor(ellc.num_rows, ellc.row_idxs, ellc.data, b)
            c = ops.spmm_ell(a, dev(b), acc="reference").cpu().numpy()
            assert np.array_equal(c, ref), f"{name} N={n}"
            assert_fast_close(ops.spmm_ell(a, dev(b), acc="fast").cpu().numpy(), ref, abs_scale(csr, b))


def test_ell_mostly_padding_multiplies_from_the_occupied_slots(oracle):
    """An ELL that is mostly padding (as wide as its longest row) carries the list of its occupied slots
    (mispmm_ell_compact_*): same slot order, same bits as the padded kernel and as spmmELLCpu -- GL7d25 (93 % padding,
    long rows: the split kernel's shape), ACTIVSg10K (81 %), a uniform ELL keeps the padded kernel."""
    for name, n, listed, split in (("GL7d25", 128, True, True), ("ACTIVSg10K", 128, True, False), ("n4c6-b13", 64, False, False),
                                   ("GL7d25", 30, True, False)):
        csr = datasets.load_csr(name)
        ell = formats.csr_to_ell_colmajor(csr)
        a = ops.DeviceELL.from_host(ell)
        assert (a.compact is not None) == listed
        b = synth.dense_b(csr.num_cols, n)
        ref = oracle.spmm_ell_colmajor(ell.num_rows, ell.row_idxs, ell.data, b)
        assert np.array_equal(ops.spmm_ell(a, dev(b)).cpu().numpy(), ref), name
        # long rows: the two-body launch (split shape for the rows of more than 32 slots, lane groups for the others)
        assert ("csr_hybrid<ref32" in capi.last_kernel()) == split, capi.last_kernel()
        padded = ops.DeviceELL.from_host(ell, compact=False)
        assert np.array_equal(ops.spmm_ell(padded, dev(b)).cpu().numpy(), ref), name
        assert_fast_close(ops.spmm_ell(a, dev(b), acc="fast").cpu().numpy(), ref, abs_scale(csr, b))


def test_ell_padding_anywhere_and_wide_rows(oracle):
    rng = np.random.default_rng(5)
    m, k, w = 37, 500, 150
    cols = np.full((m, w), formats.ELL_PAD, dtype=np.uint32)
    vals = np.zeros((m, w), dtype=np.float32)
    for r in range(m):
        n = int(rng.integers(0, w + 1))
        slots = np.sort(rng.choice(w, size=n, replace=False))          # padding holes in the middle
        cols[r, slots] = np.sort(rng.choice(k, size=n, replace=False))
        vals[r, slots] = rng.uniform(-1, 1, n)
    ell = formats.ELLRowMajor(m, k, int((cols != formats.ELL_PAD).sum()), w, cols, vals)
    b = synth.dense_b(k, 40)
    ref = np.zeros((m, 40), np.float32)
    for r in range(m):                      # fp32 += in slot order, unfused, as spmmELLCpu sums a row
        for s in range(w):
            if cols[r, s] != formats.ELL_PAD:
                ref[r] = ref[r] + vals[r, s] * b[cols[r, s]]
    c = ops.spmm_ell(ops.DeviceELL.from_host(ell), dev(b)).cpu().numpy()
    assert np.array_equal(c, ref)


def test_coo_prepared_row_bounds(oracle):
    """Kernel 2 = kernel 1 minus the per-call boundary pass: boundaries from mispmm_coo_row_bounds (or left behind by a
    kernel-1 call) give the same bits; the boundaries themselves equal the CSR row pointers."""
    for name, n in (("n4c6-b13", 128), ("qh1484", 64), ("GL7d25", 32)):
        csr = datasets.load_csr(name)
        coo = formats.csr_to_coo(csr)
        a = ops.DeviceCOO.from_host(coo)
        b = synth.dense_b(csr.num_cols, n)
        ref = oracle.spmm_coo(coo.num_rows, coo.row_idxs, coo.col_idxs, coo.data, b)
        ws = ops.coo_row_bounds(a)
        assert np.array_equal(ws.cpu().numpy().astype(np.uint32), csr.row_ptrs)
        for acc in ("reference", "fast"):
            c1 = ops.spmm_coo(a, dev(b), kernel=1, acc=acc).cpu().numpy()
            c2 = ops.spmm_coo(a, dev(b), kernel=2, acc=acc, workspace=ws).cpu().numpy()
            assert np.array_equal(c1, c2)
            if acc == "reference":
                assert np.array_equal(c2, ref)
    with pytest.raises(capi.MispmmError):
        ops.spmm_coo(a, dev(b), kernel=2, workspace=False)


@pytest.mark.parametrize("workspace", [True, False])
@pytest.mark.parametrize("name,ns", [("Hamrle1", [5, 32]), ("qh1484", [64]), ("n4c6-b13", [128]), ("GL7d25", [30])])
def test_coo_matches_oracle(oracle, name, ns, workspace):
    csr = datasets.load_csr(name)
    coo = formats.csr_to_coo(csr)
    a = ops.DeviceCOO.from_host(coo)
    for n in ns:
        b = synth.dense_b(csr.num_cols, n)
        ref = oracle.spmm_coo(coo.num_rows, coo.row_idxs, coo.col_idxs, coo.data, b)
        c = ops.spmm_coo(a, dev(b), workspace=workspace).cpu().numpy()
        assert np.array_equal(c, ref), f"{name} N={n}"
        assert_fast_close(ops.spmm_coo(a, dev(b), acc="fast", workspace=workspace).cpu().numpy(), ref, abs_scale(csr, b))


@pytest.mark.parametrize("n", [32, 128, 260])
def test_coo_and_bsr_list_long_rows_take_the_split_shape(oracle, n):
    """A COO (prepared row bounds) / BSR non-zero list with 24 entries per row or more and 16-byte B rows runs on the
    split kernel's shape -- wave per row x 32 columns, XCD column grid -- with its fp32 sums in entry order: the bits of
    spmmCOOCpu / spmmBSRCpu, for rows of 0..2500 entries, products over 2^60, sparse B, and on GL7d25."""
    lens = [0, 1, 7, 8, 9, 255, 256, 257, 513, 2500, 0, 64, 40, 25, 200, 422, 31, 33]
    csr = random_csr(len(lens), 3000, lens, seed=23)
    rng = np.random.default_rng(9)
    b = synth.dense_b(csr.num_cols, n)
    wide = b * np.exp2(rng.integers(-30, 31, size=b.shape)).astype(np.float32)
    sparse_b = np.where(rng.random(b.shape) < 0.5, np.float32(0), b)
    for mat in (csr, datasets.load_csr("GL7d25")):
        coo = formats.csr_to_coo(mat)
        a = ops.DeviceCOO.from_host(coo)
        for bb in ((b, wide, sparse_b) if mat is csr else (synth.dense_b(mat.num_cols, n),)):
            ref = oracle.spmm_coo(coo.num_rows, coo.row_idxs, coo.col_idxs, coo.data, bb)
            assert np.array_equal(ops.spmm_coo(a, dev(bb)).cpu().numpy(), ref)
            # spans built at upload; where the width has a two-body launch (32-column parts) the rows of at most 32 entries
            # go through the row-gather body of the same launch (mispmm_rows_hybrid_f32): same sums in the same order
            two_body = n in (32, 64, 128, 256)
            assert ("csr_hybrid<ref32" in capi.last_kernel()) if two_body else ("csr_split" in capi.last_kernel() and "ref32,longest-first" in capi.last_kernel())
            if two_body:
                os.environ["MISPMM_NO_HYBRID"] = "1"
                try:
                    assert np.array_equal(ops.spmm_coo(a, dev(bb)).cpu().numpy(), ref)
                    assert "csr_split" in capi.last_kernel() and "ref32,longest-first" in capi.last_kernel()
                finally:
                    del os.environ["MISPMM_NO_HYBRID"]
            ws = ops.coo_row_bounds(a)
            assert np.array_equal(ops.spmm_coo(ops.DeviceCOO(a.num_rows, a.num_cols, a.nnz, a.row_idxs, a.col_idxs, a.data), dev(bb),
                                               workspace=ws, kernel=2).cpu().numpy(), ref)                  # no spans: rows in order
            assert "csr_split" in capi.last_kernel() and "longest-first" not in capi.last_kernel()
            fast = ops.spmm_coo(a, dev(bb), acc="fast").cpu().numpy()
            assert_fast_close(fast, ref, abs_scale(mat, bb))
            assert np.array_equal(ops.spmm_coo(a, dev(bb), workspace=False).cpu().numpy(), ref)      # binary-search kernel
    gl = datasets.load_csr("GL7d25")
    bsr = formats.csr_to_bsr(gl, 2)
    nz = ops.bsr_nonzeros(bsr)
    bb = synth.dense_b(gl.num_cols, n)
    ref = oracle.spmm_bsr(bsr.num_rows, 2, 2, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, bb)
    assert np.array_equal(ops.spmm_bsr_nonzeros(nz, dev(bb)).cpu().numpy(), ref)
    assert ("csr_hybrid<ref32" in capi.last_kernel()) if n in (32, 64, 128, 256) else ("csr_split" in capi.last_kernel() and "longest-first" in capi.last_kernel())


def test_coo_with_empty_leading_and_trailing_rows(oracle):
    coo = formats.COO(9, 6, np.array([2, 2, 5, 5, 5, 6], np.uint32), np.array([0, 3, 1, 2, 5, 4], np.uint32),
                      np.array([1, -2, 3, 4, -5, 6], np.float32))
    b = synth.dense_b(6, 12)
    ref = oracle.spmm_coo(9, coo.row_idxs, coo.col_idxs, coo.data, b)
    for ws in (True, False):
        assert np.array_equal(ops.spmm_coo(ops.DeviceCOO.from_host(coo), dev(b), workspace=ws).cpu().numpy(), ref)


# --------------------------------------------------------------------------- BSR
@pytest.mark.parametrize("name,block,ns", [("Hamrle1", 1, [32]), ("Hamrle1", 4, [7, 32]), ("n3c5-b6", 2, [21]),
                                           ("qh1484", 4, [64]), ("ACTIVSg10K", 16, [128]), ("ACTIVSg10K", 32, [64]),
                                           ("dw1024", 16, [100])])
def test_bsr_valu_matches_oracle(oracle, name, block, ns):
    csr = datasets.load_csr(name)
    bsr = formats.csr_to_bsr(csr, block)
    a = ops.DeviceBSR.from_host(bsr)
    for n in ns:
        b = synth.dense_b(csr.num_cols, n)
        ref = oracle.spmm_bsr(bsr.num_rows, block, block, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, b)
        c = ops.spmm_bsr(a, dev(b), kernel=1, acc="reference").cpu().numpy()
        assert np.array_equal(c, ref), f"{name} b{block} N={n}"
        assert_fast_close(ops.spmm_bsr(a, dev(b), kernel=1, acc="fast").cpu().numpy(), ref, abs_scale(csr, b))


def test_bsr_rectangular_blocks_and_unsorted_block_columns(oracle):
    rng = np.random.default_rng(11)
    br, bc, mb, kb = 3, 5, 7, 9
    ptrs, idxs = [0], []
    for _ in range(mb):
        n = int(rng.integers(0, kb + 1))
        idxs += list(rng.permutation(kb)[:n])           # storage order is NOT ascending
        ptrs.append(len(idxs))
    data = rng.uniform(-1, 1, (len(idxs), br, bc)).astype(np.float32)
    bsr = formats.BSR(mb * br, kb * bc, data.size, br, bc, np.array(ptrs, np.uint32), np.array(idxs, np.uint32), data)
    b = synth.dense_b(kb * bc, 24)
    ref = oracle.spmm_bsr(bsr.num_rows, br, bc, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, b)
    assert np.array_equal(ops.spmm_bsr(ops.DeviceBSR.from_host(bsr), dev(b), kernel=1).cpu().numpy(), ref)


@pytest.mark.parametrize("name,n", [("ACTIVSg10K", 128), ("ACTIVSg10K", 72), ("dw1024", 64), ("qh1484", 256)])
def test_bsr_mfma_f32_fast_and_deterministic(oracle, name, n):
    """v_mfma_f32_16x16x4_f32 is an exact k-ordered fma chain; the kernel maps k to ascending block columns, its four
    waves take a block row's blocks round-robin and their partial tiles are added in fixed order: FAST numerics
    (within 1e-5 of the oracle), identical bits from run to run, and -- for block rows of at most 4 blocks, where
    every wave holds at most one block -- still comparable term by term with the FAST VALU kernel."""
    csr = datasets.load_csr(name)
    pad = (-csr.num_rows) % 16, (-csr.num_cols) % 16
    if any(pad):
        csr = formats.CSR(csr.num_rows + pad[0], csr.num_cols + pad[1],
                          np.concatenate([csr.row_ptrs, np.full(pad[0], csr.row_ptrs[-1], np.uint32)]), csr.col_idxs, csr.data)
    bsr = formats.csr_to_bsr(csr, 16)
    a = ops.DeviceBSR.from_host(bsr)
    b = synth.dense_b(csr.num_cols, n)
    valu = ops.spmm_bsr(a, dev(b), kernel=1, acc="fast").cpu().numpy()
    mfma = ops.spmm_bsr(a, dev(b), kernel=2, acc="fast").cpu().numpy()
    assert np.array_equal(mfma, ops.spmm_bsr(a, dev(b), kernel=2, acc="fast").cpu().numpy())
    ref = oracle.spmm_bsr(bsr.num_rows, 16, 16, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, b)
    assert_fast_close(mfma, ref, abs_scale(csr, b))
    assert_fast_close(valu, ref, abs_scale(csr, b))
    one_block = np.repeat(np.diff(bsr.block_row_ptrs.astype(np.int64)) <= 1, 16)   # a single chain on one wave
    assert np.array_equal(mfma[one_block], valu[one_block])
    assert np.array_equal(ops.spmm_bsr(a, dev(b), kernel=0, acc="fast").cpu().numpy(), mfma)   # auto picks MFMA


@pytest.mark.parametrize("block", [16, 32])
@pytest.mark.parametrize("n,out_bf16", [(128, False), (128, True), (72, False), (4, False), (256, True)])
def test_bsr_bf16_mfma(oracle, n, out_bf16, block):
    """BASELINE config 4 (large_20000 BSR-16 x K=128 bf16), and 32 x 32 blocks.  Oracle = the reference's
    BSR CPU semantics on bf16-rounded A and B (the reference itself has no bf16)."""
    csr = datasets.load_csr("ACTIVSg10K")
    bsr = formats.csr_to_bsr(csr, block)
    a = ops.DeviceBSR.from_host(bsr)
    b = synth.dense_b(csr.num_cols, n)
    a16 = synth.bf16_round(bsr.data.reshape(-1)).reshape(bsr.data.shape)
    b16 = synth.bf16_round(b.reshape(-1)).reshape(b.shape)
    blocks_bits = ops.f32_to_bf16(a.data)
    b_bits = ops.f32_to_bf16(dev(b))
    assert np.array_equal(ops.bf16_to_f32(blocks_bits).cpu().numpy().reshape(a16.shape), a16), "device RNE == host RNE"
    ref = oracle.spmm_bsr(bsr.num_rows, block, block, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
    c = ops.spmm_bsr_bf16(a, blocks_bits, b_bits, out_bf16=out_bf16)
    got = (ops.bf16_to_f32(c) if out_bf16 else c).cpu().numpy()
    scale = abs_scale(formats.CSR(csr.num_rows, csr.num_cols, csr.row_ptrs, csr.col_idxs,
                                  synth.bf16_round(csr.data)), b16)
    if out_bf16:
        assert np.all(np.abs(got - ref) <= 2 ** -8 * np.abs(ref) + 2e-6 * scale + 1e-30)
    else:
        assert np.all(np.abs(got.astype(np.float64) - ref) <= 2e-6 * scale + 1e-30)


@pytest.mark.parametrize("name,block,n", [("ACTIVSg10K", 16, 128), ("qh1484", 4, 64), ("Hamrle1", 2, 32), ("dw1024", 32, 40)])
def test_bsr_zero_skipping_path(oracle, name, block, n):
    """mispmm_bsr_nonzeros_*: the block entries that are not zero, multiplied in the reference's order of addition --
    bit for bit spmmBSRCpu for finite operands (a skipped 0 * b only ever adds +-0)."""
    csr = datasets.load_csr(name)
    bsr = formats.csr_to_bsr(csr, block)
    nz = ops.bsr_nonzeros(bsr)
    assert nz.nnz == int(np.count_nonzero(bsr.data))
    b = synth.dense_b(csr.num_cols, n)
    ref = oracle.spmm_bsr(bsr.num_rows, block, block, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, b)
    assert np.array_equal(ops.spmm_bsr_nonzeros(nz, dev(b)).cpu().numpy(), ref)
    assert np.array_equal(ops.spmm_bsr(ops.DeviceBSR.from_host(bsr), dev(b), kernel=1).cpu().numpy(), ref)
    assert_fast_close(ops.spmm_bsr_nonzeros(nz, dev(b), acc="fast").cpu().numpy(), ref, abs_scale(csr, b))


def test_bsr_zero_skipping_differs_only_where_a_zero_meets_a_nonfinite(oracle):
    """The documented difference: 0 * Inf = NaN in the reference's dense block arithmetic (and in kernel 1); the
    zero-skipping path never forms that product.  Everywhere else the two agree bit for bit."""
    csr = datasets.load_csr("Hamrle1")
    bsr = formats.csr_to_bsr(csr, 4)
    b = synth.dense_b(csr.num_cols, 16)
    b[5, 3] = np.inf
    ref = oracle.spmm_bsr(bsr.num_rows, 4, 4, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data, b)
    dense = ops.spmm_bsr(ops.DeviceBSR.from_host(bsr), dev(b), kernel=1).cpu().numpy()
    assert np.array_equal(dense, ref, equal_nan=True)
    skip = ops.spmm_bsr_nonzeros(ops.bsr_nonzeros(bsr), dev(b)).cpu().numpy()
    differs = ~((skip == ref) | (np.isnan(skip) & np.isnan(ref)))
    assert differs.any() and np.all(np.isnan(ref[differs])) and np.all(differs[:, [c for c in range(16) if c != 3]] == False)  # noqa: E712
    touched = np.abs(csr.to_dense()[:, 5]) > 0            # rows with a true non-zero in column 5 get the Inf either way
    assert np.all(np.isinf(skip[touched, 3]) | np.isnan(skip[touched, 3]))


@pytest.mark.parametrize("name,n,taken", [("n4c6-b13", 512, True), ("n4c6-b13", 256, True), ("n4c6-b13", 128, True),
                                          ("delaunay_n12", 256, True), ("ACTIVSg10K", 288, True), ("tols4000", 64, True),
                                          ("ACTIVSg10K", 264, False), ("tols4000", 40, False), ("qh1484", 7, False)])
def test_csr_plan_order_is_bit_exact(oracle, name, n, taken):
    """mispmm_csr_plan_f32: the rows in a clustered order (mispmm_csr_cluster_rows_host), array row i written to row
    rowMap[i] of C.  Every row keeps its entries in storage order, so the result equals the oracle bit for bit, for
    uniform rows (no row pointers), ragged rows, rows with no entry, a strided C, single and batched launches.  Shapes
    the row-mapped kernel does not take (column parts that are not whole 32-column groups, narrow vectors) are declined
    and the public entry point multiplies from the unpermuted arrays."""
    csr = datasets.load_csr(name)
    b = synth.dense_b(csr.num_cols, n)
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    a = ops.DeviceCSR.from_host(csr, plan=True, spans=False)
    assert a.plan is not None and a.plan.clustered_distinct > 0
    bd = dev(b)
    cw = torch.full((csr.num_rows, n + 8), -7.0, device="cuda")
    assert ops._csr_plan(a, [bd], [cw[:, :n]], "reference", None) == taken
    if taken:
        assert "plan-order" in capi.last_kernel()
        got = cw.cpu().numpy()
        assert np.array_equal(got[:, :n], ref) and np.all(got[:, n:] == -7.0)
        outs = [torch.empty((csr.num_rows, n), device="cuda") for _ in range(3)]
        assert ops._csr_plan(a, [bd, bd * 2.0, bd], outs, "reference", None)
        assert np.array_equal(outs[0].cpu().numpy(), ref) and np.array_equal(outs[2].cpu().numpy(), ref)
        assert np.array_equal(outs[1].cpu().numpy(), ops.spmm_csr(a, bd * 2.0, use_hint=False).cpu().numpy())
        fast = torch.empty((csr.num_rows, n), device="cuda")
        assert ops._csr_plan(a, [bd], [fast], "fast", None)
        assert_fast_close(fast.cpu().numpy(), ref, abs_scale(csr, b))
    else:
        assert np.all(cw.cpu().numpy() == -7.0)                      # declined before anything was launched
    # the public entry point MEASURES plan order against storage order on the first product of a width, remembers the answer,
    # takes what it measured -- and gives the same bits either way.  The decision is a pure function of the two timings.
    assert np.array_equal(ops.spmm_csr(a, bd).cpu().numpy(), ref)
    tag = capi.last_kernel()
    use, times = a.tuned[(n, "reference")]
    assert ("plan-order" in tag) == use and use == (ops.autotune_pick(times) == 1)
    assert use or not taken or times[1] >= times[0] * 0.98           # the plan is dropped only when it is not 2 % faster
    if not taken:
        assert not use and times[1] == float("inf")                  # a shape the plan launch declines can never be chosen
    ops.spmm_csr(a, bd)
    assert capi.last_kernel() == tag and len(a.tuned) == 1           # measured once
    # the footprint rule stays as the prior where nothing can be measured (a stream being captured, MISPMM_AUTOTUNE=0)
    assert ops.plan_pays(25605, 512) and not ops.plan_pays(25605, 256) and not ops.plan_pays(4096, 512)


def test_autotune_decision_is_a_pure_function_of_the_timings():
    """mispmm_autotune_pick: candidate 0 (storage order) is kept unless another is at least min_gain faster; a candidate that
    could not run (inf), a zero or a NaN never wins; ties and noise inside the margin keep the default -- the same timings give
    the same choice, whatever produced them."""
    pick = ops.autotune_pick
    assert pick([12.6, 11.1]) == 1 and pick([3.47, 3.69]) == 0 and pick([13.60, 12.99]) == 1
    assert pick([10.0, 9.85]) == 0 and pick([10.0, 9.79]) == 1                 # 2 % margin
    assert pick([10.0, float("inf")]) == 0 and pick([float("inf"), 5.0]) == 1 and pick([10.0, float("nan")]) == 0
    assert pick([10.0, 0.0]) == 0 and pick([10.0, 9.0, 8.0]) == 2 and pick([10.0, 9.9, 9.85]) == 0
    assert pick([10.0, 9.0], min_gain=0.2) == 0
    assert all(pick([12.6, 11.1]) == 1 for _ in range(5))


def test_autotune_takes_the_plan_the_footprint_rule_misses():
    """ACTIVSg10K x K=256: 2.6 MB of B per XCD fit the L2, so the footprint rule says "storage order" -- measured, the clustered
    plan order is faster (12.6 -> 11.1 us, DESIGN.md section 9.4 of round 3).  The autotune measures it and takes the plan; the
    headline (n4c6-b13 x K=128: plan +6 %) keeps the storage order.  Timings are printed into the assertion message."""
    csr = datasets.load_csr("ACTIVSg10K")
    a = ops.DeviceCSR.from_host(csr, spans=False)
    assert a.plan is not None and not ops.plan_pays(csr.num_cols, 256)
    ops.spmm_csr(a, dev(synth.dense_b(csr.num_cols, 256)))
    use, times = a.tuned[(256, "reference")]
    assert use == (ops.autotune_pick(times) == 1), times
    head = ops.DeviceCSR.from_host(datasets.load_csr("n4c6-b13"))
    ops.spmm_csr(head, dev(synth.dense_b(head.num_cols, 128)))
    h_use, h_times = head.tuned[(128, "reference")]
    assert h_use == (ops.autotune_pick(h_times) == 1), h_times
    print("autotune ACTIVSg10K x 256:", use, times, "| n4c6-b13 x 128:", h_use, h_times)


@pytest.mark.parametrize("n", [128, 40, 6])
def test_general_csr_entry_bets_on_uniform_rows_and_recovers(oracle, n):
    """The general entry point (no structure hint) with nnz == M * w: the row-gather waves fetch their first entries from
    r * w before the row pointers have arrived.  Right for a uniform matrix (n4c6-b13: same bits, one hop less), wrong for
    a ragged matrix whose entry count merely divides by M -- there every wave must notice and fetch again.  Rows of 0
    entries, rows longer than the 16-entry window, a last partial wave; the guess can be switched off in the tuning build."""
    lens = [14, 0, 27, 1, 14, 30, 0, 26, 14, 14, 2, 40, 14, 14, 0, 14] * 9 + [14, 14, 28, 0, 14]   # 149 rows, mean 14 exactly
    assert sum(lens) == 14 * len(lens)
    for csr in (random_csr(len(lens), 333, lens, 3), random_csr(149, 333, [14] * 149, 4), datasets.load_csr("n4c6-b13")):
        assert csr.nnz % csr.num_rows == 0
        b = synth.dense_b(csr.num_cols, n)
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
        a = ops.DeviceCSR.from_host(csr, plan=False)
        got = ops.spmm_csr(a, dev(b), use_hint=False).cpu().numpy()
        assert "csr" in capi.last_kernel()
        assert np.array_equal(got, ref)
        if n == 128:
            outs = ops.spmm_csr_batch(ops.DeviceCSR(a.num_rows, a.num_cols, a.nnz, a.row_ptrs, a.col_idxs, a.data), [dev(b), dev(b)])
            assert np.array_equal(outs[1].cpu().numpy(), ref)


def test_csr_plan_is_kept_only_where_clustering_pays():
    """from_host(plan=None): a plan for the BASELINE matrices (clustering cuts the per-part distinct columns by 16-30 %),
    none for a matrix whose storage order is already the best the greedy walk finds, none for long-row matrices (span list)."""
    assert ops.DeviceCSR.from_host(datasets.load_csr("n4c6-b13")).plan is not None
    assert ops.DeviceCSR.from_host(datasets.load_csr("ch7-6-b5")).plan is None
    gl = ops.DeviceCSR.from_host(datasets.load_csr("GL7d25"))
    assert gl.plan is None and gl.spans is not None


@pytest.mark.parametrize("n,out_bf16", [(128, False), (128, True), (72, False), (256, True), (8, False)])
def test_bsrc_bf16_column_compacted_block_rows(oracle, n, out_bf16):
    """mispmm_bsrc_bf16 (BASELINE config 4's default kernel): per block row the occupied columns and the values gathered
    to them as 16 x 32 bf16 tiles, one MFMA step per 32 columns.  Same oracle and bound as the dense-block kernel."""
    csr = datasets.load_csr("ACTIVSg10K")
    bsr = formats.csr_to_bsr(csr, 16)
    a = ops.DeviceBSRC.from_host(bsr)
    nzmask = synth.bf16_round(csr.data) != 0       # the matrix stores a few explicit zeros: those columns are not occupied
    occupied = sum(len(np.unique(csr.col_idxs[csr.row_ptrs[r]:csr.row_ptrs[r + 16]][nzmask[csr.row_ptrs[r]:csr.row_ptrs[r + 16]]]))
                   for r in range(0, csr.num_rows, 16))
    assert int((a.cols.cpu().numpy().view(np.uint32) != 0xFFFFFFFF).sum()) == occupied
    b = synth.dense_b(csr.num_cols, n)
    a16 = synth.bf16_round(bsr.data.reshape(-1)).reshape(bsr.data.shape)
    b16 = synth.bf16_round(b.reshape(-1)).reshape(b.shape)
    ref = oracle.spmm_bsr(bsr.num_rows, 16, 16, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
    c = ops.spmm_bsrc_bf16(a, ops.f32_to_bf16(dev(b)), out_bf16=out_bf16)
    got = (ops.bf16_to_f32(c) if out_bf16 else c).cpu().numpy()
    scale = abs_scale(formats.CSR(csr.num_rows, csr.num_cols, csr.row_ptrs, csr.col_idxs, synth.bf16_round(csr.data)), b16)
    if out_bf16:
        assert np.all(np.abs(got - ref) <= 2 ** -8 * np.abs(ref) + 2e-6 * scale + 1e-30)
    else:
        assert np.all(np.abs(got.astype(np.float64) - ref) <= 2e-6 * scale + 1e-30)
    again = ops.spmm_bsrc_bf16(a, ops.f32_to_bf16(dev(b)), out_bf16=out_bf16)
    assert torch.equal(c, again)                                     # deterministic


def test_bsrc_bf16_small_and_ragged_block_rows(oracle):
    """Block rows with 0, 1 and more than 32 occupied columns, rectangular 16 x 8 blocks, a width that is refused."""
    rng = np.random.default_rng(11)
    mb, kb, bc = 9, 40, 8
    ptrs, idxs, blocks = [0], [], []
    for r in range(mb):
        cnt = [0, 1, 12, 3, 0, 7, 25, 2, 5][r]
        cols = np.sort(rng.choice(kb, size=cnt, replace=False))
        idxs += list(cols)
        for _ in range(cnt):
            blk = np.where(rng.random((16, bc)) < 0.3, rng.uniform(-2, 2, (16, bc)), 0.0).astype(np.float32)
            blocks.append(blk)
        ptrs.append(len(idxs))
    data = np.stack(blocks) if blocks else np.zeros((0, 16, bc), np.float32)
    bsr = formats.BSR(mb * 16, kb * bc, int(data.size), 16, bc, np.array(ptrs, np.uint32), np.array(idxs, np.uint32), data)
    n = 64
    b = synth.dense_b(kb * bc, n)
    a16 = synth.bf16_round(data.reshape(-1)).reshape(data.shape)
    b16 = synth.bf16_round(b.reshape(-1)).reshape(b.shape)
    ref = oracle.spmm_bsr(mb * 16, 16, bc, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
    c = ops.spmm_bsrc_bf16(ops.DeviceBSRC.from_host(bsr), ops.f32_to_bf16(dev(b))).cpu().numpy()
    scale = np.abs(bsr_to_dense(bsr, a16)).astype(np.float64) @ np.abs(b16).astype(np.float64)
    assert np.all(np.abs(c.astype(np.float64) - ref) <= 2e-6 * scale + 1e-30)
    assert np.all(c[4 * 16:5 * 16] == 0)                             # an empty block row is overwritten with zeros
    with pytest.raises(capi.MispmmError):
        ops.spmm_bsrc_bf16(ops.DeviceBSRC.from_host(bsr), ops.f32_to_bf16(dev(synth.dense_b(kb * bc, 12))))


@pytest.mark.parametrize("n,out_bf16", [(128, False), (128, True), (72, False), (256, True), (8, False)])
def test_bsrc_slots_bf16_workgroup_per_block_row(oracle, n, out_bf16):
    """mispmm_bsrc_slots_bf16 (BASELINE config 4's default kernel from round 3): the compacted steps of a block row in 4
    fixed slots, one wave each, partial tiles added in wave order through LDS.  Same oracle and bound as the other bf16
    kernels; the layout keeps exactly the occupied columns mispmm_bsrc_bf16 keeps."""
    csr = datasets.load_csr("ACTIVSg10K")
    bsr = formats.csr_to_bsr(csr, 16)
    a = ops.DeviceBSRCSlots.from_host(bsr)
    old = ops.DeviceBSRC.from_host(bsr)
    assert a.used_steps == old.num_steps and a.num_steps >= 4 * bsr.num_block_rows
    acols = a.cols.cpu().numpy().view(np.uint32)
    assert int((acols != 0xFFFFFFFF).sum()) == int((old.cols.cpu().numpy().view(np.uint32) != 0xFFFFFFFF).sum())
    b = synth.dense_b(csr.num_cols, n)
    a16 = synth.bf16_round(bsr.data.reshape(-1)).reshape(bsr.data.shape)
    b16 = synth.bf16_round(b.reshape(-1)).reshape(b.shape)
    ref = oracle.spmm_bsr(bsr.num_rows, 16, 16, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
    c = ops.spmm_bsrc_slots_bf16(a, ops.f32_to_bf16(dev(b)), out_bf16=out_bf16)
    assert "bsrc_slots" in capi.last_kernel()
    got = (ops.bf16_to_f32(c) if out_bf16 else c).cpu().numpy()
    scale = abs_scale(formats.CSR(csr.num_rows, csr.num_cols, csr.row_ptrs, csr.col_idxs, synth.bf16_round(csr.data)), b16)
    if out_bf16:
        assert np.all(np.abs(got - ref) <= 2 ** -8 * np.abs(ref) + 2e-6 * scale + 1e-30)
    else:
        assert np.all(np.abs(got.astype(np.float64) - ref) <= 2e-6 * scale + 1e-30)
    again = ops.spmm_bsrc_slots_bf16(a, ops.f32_to_bf16(dev(b)), out_bf16=out_bf16)
    assert torch.equal(c, again)                                     # deterministic


def test_bsrc_slots_bf16_ragged_rows_extra_steps_and_strides(oracle):
    """Block rows with 0 .. 11 steps (more than 4 = extra steps behind the slots), 16 x 8 blocks, C and B with a
    leading dimension larger than N (the gap in C keeps its sentinel), a width that is refused."""
    rng = np.random.default_rng(12)
    mb, kb, bc = 10, 60, 8
    ptrs, idxs, blocks = [0], [], []
    for r in range(mb):
        cnt = [0, 1, 12, 3, 0, 7, 25, 2, 44, 60][r]
        cols = np.sort(rng.choice(kb, size=cnt, replace=False))
        idxs += list(cols)
        for _ in range(cnt):
            blocks.append(np.where(rng.random((16, bc)) < 0.6, rng.uniform(-2, 2, (16, bc)), 0.0).astype(np.float32))
        ptrs.append(len(idxs))
    data = np.stack(blocks)
    bsr = formats.BSR(mb * 16, kb * bc, int(data.size), 16, bc, np.array(ptrs, np.uint32), np.array(idxs, np.uint32), data)
    a = ops.DeviceBSRCSlots.from_host(bsr)
    extra = a.extra_ptrs.cpu().numpy().view(np.uint32)
    assert extra[-1] > 0 and a.num_steps == 4 * mb + int(extra[-1])  # the wide block rows really spill into extra steps
    n, ld = 136, 144
    b = synth.dense_b(kb * bc, n)
    a16 = synth.bf16_round(data.reshape(-1)).reshape(data.shape)
    b16 = synth.bf16_round(b.reshape(-1)).reshape(b.shape)
    ref = oracle.spmm_bsr(mb * 16, 16, bc, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
    scale = np.abs(bsr_to_dense(bsr, a16)).astype(np.float64) @ np.abs(b16).astype(np.float64)
    bw = torch.zeros((kb * bc, ld), dtype=torch.int16, device="cuda")
    bw[:, :n] = ops.f32_to_bf16(dev(b))
    for sc1 in (True,):
        cw = torch.full((mb * 16, ld), -7.0, device="cuda")
        ops.spmm_bsrc_slots_bf16(a, bw[:, :n], out=cw[:, :n])
        got = cw.cpu().numpy()
        assert np.all(np.abs(got[:, :n].astype(np.float64) - ref) <= 2e-6 * scale + 1e-30)
        assert np.all(got[:, n:] == -7.0)
        assert np.all(got[4 * 16:5 * 16, :n] == 0)                   # an empty block row is overwritten with zeros
    dense = ops.spmm_bsrc_bf16(ops.DeviceBSRC.from_host(bsr), ops.f32_to_bf16(dev(b))).cpu().numpy()
    assert np.all(np.abs(dense.astype(np.float64) - got[:, :n]) <= 4e-6 * scale + 1e-30)
    with pytest.raises(capi.MispmmError):
        ops.spmm_bsrc_slots_bf16(a, ops.f32_to_bf16(dev(synth.dense_b(kb * bc, 12))))


def test_bsrc_slots_shared_reduce_returns_the_bits_of_the_wave0_reduce():
    """Block rows of at most 4 steps (no extra steps: config 4) take the kernel instance that deals the reduce and the
    store over the four waves; it adds the partial tiles in the same wave order as the instance in which wave 0 does both
    (what every layout with extra steps runs, and what a bf16 C keeps: there the shared reduce measured slower), so the two
    must agree bit for bit.  MISPMM_BSR_SHARE=0 / 2 (tuning build, read once per process: child processes) force the latter /
    the former for both C types; the default takes the shared reduce for an fp32 C only."""
    import subprocess
    import sys
    import tempfile
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-optimization-for-spmm_amd")
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {pkg!r})\n"
        "from mispmm import capi, datasets, formats, ops, synth\n"
        "csr = datasets.load_csr('ACTIVSg10K'); a = ops.DeviceBSRCSlots.from_host(formats.csr_to_bsr(csr, 16))\n"
        "out = {}\n"
        "for n in (128, 72, 256):\n"
        "    b = ops.f32_to_bf16(torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda())\n"
        "    for c16 in (0, 1):\n"
        "        out['%d_%d' % (n, c16)] = ops.spmm_bsrc_slots_bf16(a, b, out_bf16=bool(c16)).cpu().numpy()\n"
        "        out['tag_%d_%d' % (n, c16)] = np.array(capi.last_kernel())\n"
        "np.savez(sys.argv[1], tag=np.array(capi.last_kernel()), **out)\n")
    tune = os.path.join(pkg, "libmispmm_tune.so")
    assert os.path.exists(tune), "run `make -C cuda-optimization-for-spmm_amd tune`"
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for share in ("0", "1", "2"):
            path = os.path.join(tmp, f"out{share}.npz")
            p = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, MISPMM_BSR_SHARE=share, MISPMM_LIB=tune),
                               capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-2000:]
            res[share] = dict(np.load(path))
        for key in res["0"]:
            if key.startswith("tag"):
                continue
            n, c16 = key.split("_")
            assert "share" not in str(res["0"]["tag_" + key]) and "share" in str(res["2"]["tag_" + key]), key
            assert ("share" in str(res["1"]["tag_" + key])) == (c16 == "0"), key          # the default: an fp32 C only
            for other in ("1", "2"):
                assert np.array_equal(res["0"][key].view(np.uint8), res[other][key].view(np.uint8)), (key, other)


def bsr_to_dense(bsr, data):
    d = np.zeros((bsr.num_rows, bsr.num_cols), dtype=np.float32)
    br, bc = bsr.block_row_size, bsr.block_col_size
    for r in range(bsr.num_rows // br):
        for k in range(int(bsr.block_row_ptrs[r]), int(bsr.block_row_ptrs[r + 1])):
            cidx = int(bsr.block_col_idxs[k])
            d[r * br:(r + 1) * br, cidx * bc:(cidx + 1) * bc] = data[k]
    return d


def test_bsr_bf16_lds_staged_kernel(oracle):
    """The opt-in LDS-staged kernel (MISPMM_BSR_LDS=1: LDS-DMA ring + transposed LDS reads) against the same oracle and
    bound as the register-staged default (test_bsr_bf16_mfma), incl. a partial last column tile (N = 72) and bf16
    output.  Run in a child process: the choice of kernel is read once per process."""
    import subprocess
    import sys
    import tempfile
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cuda-optimization-for-spmm_amd")
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {pkg!r})\n"
        "from mispmm import capi, datasets, formats, ops, synth\n"
        "csr = datasets.load_csr('ACTIVSg10K'); bsr = formats.csr_to_bsr(csr, 16); a = ops.DeviceBSR.from_host(bsr)\n"
        "out = {}\n"
        "for n in (128, 72, 256):\n"
        "    b = torch.from_numpy(synth.dense_b(csr.num_cols, n)).cuda()\n"
        "    for c16 in (0, 1):\n"
        "        c = ops.spmm_bsr_bf16(a, ops.f32_to_bf16(a.data), ops.f32_to_bf16(b), out_bf16=bool(c16))\n"
        "        out['%d_%d' % (n, c16)] = (ops.bf16_to_f32(c) if c16 else c).cpu().numpy()\n"
        "np.savez(sys.argv[1], tag=np.array(capi.last_kernel()), **out)\n")
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "out.npz")
        # the knob is live only in the tuning build of the library (libmispmm_tune.so, -DMISPMM_TUNING)
        tune = os.path.join(pkg, "libmispmm_tune.so")
        assert os.path.exists(tune), "run `make -C cuda-optimization-for-spmm_amd tune`"
        p = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, MISPMM_BSR_LDS="1", MISPMM_LIB=tune),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res = np.load(path)
        assert "bsr_bf16_lds" in str(res["tag"])
        csr = datasets.load_csr("ACTIVSg10K")
        bsr = formats.csr_to_bsr(csr, 16)
        a16 = synth.bf16_round(bsr.data.reshape(-1)).reshape(bsr.data.shape)
        for n in (128, 72, 256):
            b16 = synth.bf16_round(synth.dense_b(csr.num_cols, n).reshape(-1)).reshape(csr.num_cols, n)
            ref = oracle.spmm_bsr(bsr.num_rows, 16, 16, bsr.block_row_ptrs, bsr.block_col_idxs, a16, b16)
            scale = abs_scale(formats.CSR(csr.num_rows, csr.num_cols, csr.row_ptrs, csr.col_idxs, synth.bf16_round(csr.data)), b16)
            assert np.all(np.abs(res[f"{n}_0"].astype(np.float64) - ref) <= 2e-6 * scale + 1e-30), n
            assert np.all(np.abs(res[f"{n}_1"] - ref) <= 2 ** -8 * np.abs(ref) + 2e-6 * scale + 1e-30), n


# --------------------------------------------------------------------------- dense helpers
@pytest.mark.parametrize("shape", [(1, 1), (3, 130), (64, 64), (257, 65), (1000, 31)])
def test_dense_transpose(shape):
    x = torch.arange(shape[0] * shape[1], dtype=torch.float32, device="cuda").view(shape)
    assert torch.equal(ops.dense_transpose(x), x.t().contiguous())


def test_bf16_conversion_round_to_nearest_even():
    x = np.array([1.0, 1.00390625, 1.01171875, -3.3, 0.0, -0.0, np.inf, 65504.0, 1e-40], np.float32)
    bits = ops.f32_to_bf16(dev(x))
    back = ops.bf16_to_f32(bits).cpu().numpy()
    assert np.array_equal(back, synth.bf16_round(x))
    assert np.isnan(ops.bf16_to_f32(ops.f32_to_bf16(dev(np.array([np.nan], np.float32)))).cpu().numpy()[0])


# --------------------------------------------------------------------------- > 2 GiB dense operand
def test_wide_address_fallbacks_for_b_beyond_2gib(oracle):
    """A B whose rows are 32 KiB apart spans 2.2 GiB: buffer offsets no longer fit, every format must
    take its 64-bit-address kernel and still be bit-exact.  Only N = 8 columns are real data."""
    k_rows, ldb, n = 70000, 8192, 8
    free, _ = torch.cuda.mem_get_info()
    if free < 6 * 2 ** 30:
        pytest.skip("needs ~2.3 GiB of device memory")
    rng = np.random.default_rng(21)
    lens = rng.integers(0, 40, size=300)
    csr = random_csr(300, k_rows, lens, seed=22)
    b = synth.dense_b(k_rows, n)
    big = torch.zeros((k_rows, ldb), dtype=torch.float32, device="cuda")
    bview = big[:, :n]
    bview.copy_(dev(b))
    assert bview.stride(0) * k_rows * 4 > 2 ** 31
    ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b)
    a = ops.DeviceCSR.from_host(csr)
    for kern in (0, 1, 3, 5):
        assert np.array_equal(ops.spmm_csr(a, bview, kernel=kern).cpu().numpy(), ref), kern
    coo = formats.csr_to_coo(csr)
    ref32 = oracle.spmm_coo(coo.num_rows, coo.row_idxs, coo.col_idxs, coo.data, b)
    for ws in (True, False):
        assert np.array_equal(ops.spmm_coo(ops.DeviceCOO.from_host(coo), bview, workspace=ws).cpu().numpy(), ref32)
    ell = formats.csr_to_ell_rowmajor(csr)
    assert np.array_equal(ops.spmm_ell(ops.DeviceELL.from_host(ell), bview).cpu().numpy(), ref32)
    pad = (-csr.num_cols) % 4
    bsr = formats.csr_to_bsr(formats.CSR(300, k_rows + pad, csr.row_ptrs, csr.col_idxs, csr.data), 4)
    bb = torch.zeros((k_rows + pad, ldb), dtype=torch.float32, device="cuda")[:, :n]
    bb[:k_rows].copy_(dev(b))
    refb = oracle.spmm_bsr(300, 4, 4, bsr.block_row_ptrs, bsr.block_col_idxs, bsr.data,
                           np.vstack([b, np.zeros((pad, n), np.float32)]))
    assert np.array_equal(ops.spmm_bsr(ops.DeviceBSR.from_host(bsr), bb, kernel=1).cpu().numpy(), refb)
    # a C whose rows are 8 MiB apart (2.5 GiB span): write-through buffer stores no longer fit, plain stores take over
    del big, bb
    torch.cuda.empty_cache()
    small = random_csr(300, 500, rng.integers(0, 20, size=300), seed=23)
    bs = synth.dense_b(500, 64)
    huge_c = torch.full((300, 2 ** 21), -1.0, dtype=torch.float32, device="cuda")
    cview = huge_c[:, :64]
    ops.spmm_csr(ops.DeviceCSR.from_host(small), dev(bs), out=cview)
    assert np.array_equal(cview.cpu().numpy(), oracle.spmm_csr(small.row_ptrs, small.col_idxs, small.data, bs))
    assert float(huge_c[:, 64:128].max()) == -1.0
    del huge_c
    torch.cuda.empty_cache()
    # the MFMA kernels decline instead of truncating offsets
    bsr16 = formats.csr_to_bsr(formats.CSR(304, k_rows + (-k_rows) % 16, np.concatenate([csr.row_ptrs, np.full(4, csr.row_ptrs[-1], np.uint32)]),
                                           csr.col_idxs, csr.data), 16)
    with pytest.raises(capi.MispmmError) as e:
        ops.spmm_bsr(ops.DeviceBSR.from_host(bsr16), torch.zeros((bsr16.num_cols, ldb), device="cuda")[:, :n], kernel=2, acc="fast")
    assert e.value.status == capi.ERR_UNSUPPORTED


def test_vendor_check_agrees_within_fp32_accumulate_bound(oracle):
    """rocSPARSE behind mispmm_vendor_spmm_f32 (the cusparseTest replacement) against the oracle: CSR and COO, a
    BASELINE matrix and a dense-ish one with values in (-100, 100) like the sparsity sweep.  The vendor library sums in
    fp32, so the bound is n * eps relative to sum |a||b| -- the reference's own allclose (rtol 1e-2 on the RESULT) is not
    met on the second matrix, which is why the sweep prints `correct 0` for kernel -1 there."""
    import ctypes
    rng = np.random.default_rng(42)
    dense_ish = random_csr(256, 512, [int(x) for x in rng.integers(100, 400, 256)], seed=43)
    dense_ish = formats.CSR(dense_ish.num_rows, dense_ish.num_cols, dense_ish.row_ptrs, dense_ish.col_idxs,
                            rng.uniform(-100, 100, dense_ish.nnz).astype(np.float32))
    l = capi.lib()
    for csr, n, amp in ((datasets.load_csr("n4c6-b13"), 128, 1.0), (dense_ish, 64, 100.0)):
        b = (synth.dense_b(csr.num_cols, n) * amp).astype(np.float32)
        ref = oracle.spmm_csr(csr.row_ptrs, csr.col_idxs, csr.data, b).astype(np.float64)
        scale = abs_scale(csr, b)
        longest = int(np.diff(csr.row_ptrs.astype(np.int64)).max())
        a = ops.DeviceCSR.from_host(csr)
        coo = ops.DeviceCOO.from_host(formats.csr_to_coo(csr))
        bd = dev(b)
        for fmt, first, cols, vals in ((0, a.row_ptrs, a.col_idxs, a.data), (1, coo.row_idxs, coo.col_idxs, coo.data)):
            c = torch.zeros((csr.num_rows, n), dtype=torch.float32, device="cuda")
            t = [ctypes.c_double() for _ in range(3)]
            capi.check(l.mispmm_vendor_spmm_f32(None, fmt, csr.num_rows, csr.num_cols, csr.nnz, 0, ops._p(first), ops._p(cols),
                                                ops._p(vals), ops._p(bd), n, n, ops._p(c), n, *[ctypes.byref(x) for x in t]))
            err = np.abs(c.cpu().numpy().astype(np.float64) - ref)
            assert np.all(err <= longest * 2.0 ** -23 * scale + 1e-30), (fmt, float(np.max(err / (scale + 1e-30))))
            assert t[1].value > 0


def test_uniform_entry_point_edges():
    """rowNnz = 0 (every row empty) overwrites C with zeros; a B of 2 GiB or more is refused, not mis-addressed."""
    l = capi.lib()
    m, k, n = 70, 40, 24
    b = dev(synth.dense_b(k, n))
    c = torch.full((m, n), 5.0, dtype=torch.float32, device="cuda")
    one = torch.zeros(1, dtype=torch.int32, device="cuda")
    capi.check(l.mispmm_csr_uniform_f32(None, m, k, 0, ops._p(one), ops._p(one), ops._p(b), n, n, ops._p(c), n, 0))
    assert torch.all(c == 0)
    assert l.mispmm_csr_uniform_f32(None, m, 1 << 20, 1, ops._p(one), ops._p(one), ops._p(b), 1024, 1024, ops._p(c), 1024, 0) \
        == capi.ERR_UNSUPPORTED
