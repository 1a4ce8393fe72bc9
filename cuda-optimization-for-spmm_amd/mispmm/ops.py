"""SpMM entry points on torch tensors: thin plumbing from torch device memory and
streams to the C ABI (data_ptr() in, data_ptr() out).  PyTorch is used for
allocation and stream handles only -- all arithmetic happens in libmispmm.so."""
import ctypes
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import capi, formats


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def _dev_u32(a, device):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return torch.from_numpy(a.view(np.int32).copy()).to(device)


def _dev_f32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None and t.numel() else 0)


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise ValueError("mispmm ops take device tensors; there is no CPU path")


def _dense_ld(t):
    if t.dim() != 2 or t.dtype != torch.float32 or t.stride(1) != 1 and t.shape[1] > 1:
        raise ValueError("dense operands must be 2-D float32 with unit column stride")
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))


def uniform_row_nnz(row_ptrs):
    """w if every row holds exactly w entries (rowPtrs[r] == r * w), else 0.  O(M) host scan."""
    rp = np.asarray(row_ptrs, dtype=np.int64)
    if rp.shape[0] < 2 or rp[0] != 0:
        return 0
    w = int(rp[1])
    return w if w > 0 and np.array_equal(rp, np.arange(rp.shape[0], dtype=np.int64) * w) else 0


def csr_spans_by_length(row_ptrs, share_len=0):
    """The span list mispmm_csr_split_f32 walks (mispmm_csr_spans_by_length_host): [count, 4] uint32, the rows as
    (row, start, end, 0) longest first, preceded by the rows of more than share_len entries (0 = 128) as 4 chunks
    (row, start, end, 1) each."""
    rp = np.ascontiguousarray(row_ptrs, dtype=np.uint32)
    m = max(rp.shape[0] - 1, 0)
    count = ctypes.c_uint32(0)
    capi.check(capi.lib().mispmm_csr_spans_by_length_host(m, rp.ctypes.data, int(share_len), ctypes.byref(count), None))
    spans = np.zeros((count.value, 4), dtype=np.uint32)
    if count.value:
        capi.check(capi.lib().mispmm_csr_spans_by_length_host(m, rp.ctypes.data, int(share_len), ctypes.byref(count), spans.ctypes.data))
    return spans


HYBRID_ROW_LEN = 32   # rows of more entries go to the split body of mispmm_csr_hybrid_f32, the others to its row-gather body
LONGEST_ROW_FOR_SPANS = 64   # a matrix of short rows with a row this long (tols4000: mean 2.2, longest 90) gets a span list too


def wants_spans(row_ptrs):
    """(build a span list, for the two-body launch only): long rows on average (24 entries or more: the split kernel's
    domain), or short rows on average with a few long ones -- a lane group walks a row of L entries in L / 8 memory round
    trips whatever the rest of the GPU does, so those few rows decide the launch; the two-body launch gives them to the
    split kernel's body.  Without such a launch for the shape the second kind keeps its format's own kernel."""
    rp = np.asarray(row_ptrs, dtype=np.int64)
    m = rp.shape[0] - 1
    if m <= 0 or rp[-1] == 0:
        return False, False
    if int(rp[-1]) // m >= 24:
        return True, False
    build = int(np.diff(rp).max()) >= LONGEST_ROW_FOR_SPANS
    return build, build


def spans_long_count(spans, threshold=HYBRID_ROW_LEN):
    """How many leading positions of a span list hold the long rows (mispmm_csr_spans_long_count_host)."""
    sp = np.ascontiguousarray(spans, dtype=np.uint32).reshape(-1, 4)
    n = ctypes.c_uint32(0)
    capi.check(capi.lib().mispmm_csr_spans_long_count_host(sp.shape[0], sp.ctypes.data, int(threshold), ctypes.byref(n)))
    return n.value


def cluster_rows(csr, parts=4):
    """Greedy row clustering (mispmm_csr_cluster_rows_host): (order, natural_distinct, clustered_distinct) -- order[i] = the
    original row at position i; the two counts are the distinct columns summed over `parts` equal row parts before / after."""
    rp = np.ascontiguousarray(csr.row_ptrs, dtype=np.uint32)
    ci = np.ascontiguousarray(csr.col_idxs, dtype=np.uint32)
    order = np.empty(csr.num_rows, dtype=np.uint32)
    nat, clu = ctypes.c_uint64(0), ctypes.c_uint64(0)
    capi.check(capi.lib().mispmm_csr_cluster_rows_host(csr.num_rows, csr.num_cols, rp.ctypes.data, ci.ctypes.data, int(parts),
                                                        order.ctypes.data, ctypes.byref(nat), ctypes.byref(clu)))
    return order, nat.value, clu.value


def permute_rows(csr, order):
    """The CSR whose row i is row order[i] of `csr` (mispmm_csr_permute_rows_host)."""
    rp = np.ascontiguousarray(csr.row_ptrs, dtype=np.uint32)
    ci = np.ascontiguousarray(csr.col_idxs, dtype=np.uint32)
    va = np.ascontiguousarray(csr.data, dtype=np.float32)
    od = np.ascontiguousarray(order, dtype=np.uint32)
    rp2, ci2, va2 = np.empty_like(rp), np.empty_like(ci), np.empty_like(va)
    capi.check(capi.lib().mispmm_csr_permute_rows_host(csr.num_rows, rp.ctypes.data, ci.ctypes.data, va.ctypes.data, od.ctypes.data,
                                                        rp2.ctypes.data, ci2.ctypes.data, va2.ctypes.data))
    return formats.CSR(csr.num_rows, csr.num_cols, rp2, ci2, va2)


@dataclass
class CsrPlan:
    """The rows of a CSR in a clustered order (rows that read the same B rows together), for mispmm_csr_plan_f32."""
    row_ptrs: torch.Tensor
    col_idxs: torch.Tensor
    data: torch.Tensor
    row_map: torch.Tensor        # row_map[i] = the C row array row i produces
    parts: int
    natural_distinct: int        # distinct columns summed over the parts, storage order ...
    clustered_distinct: int      # ... and plan order


PLAN_PARTS = 8                   # clusters: K=512 in-process A/B (profiles/r3/plan_order.log): 4 clusters -3.3 %, 8 -4.5 %, 16 -3.0 %, 64 -0.6 %
PLAN_MIN_GAIN = 0.10             # keep a plan only if it cuts the per-part distinct columns by this much
# When the plan is USED.  Measured on MI355X, in-process A/B on one set of operands (profiles/r3/plan_order.log): the
# clustered order pays where the slice of B an XCD reads -- every B row x the XCD's column part -- does not fit its 4 MiB
# L2 (n4c6-b13 x K=512: 6.5 MB per XCD, 13.60 -> 12.99 us) and LOSES elsewhere (K=128 3.47 -> 3.69 us, K=256 with 3.3 MB
# per XCD 5.75 -> 6.51: the scattered C rows cost more than the order saves).  MISPMM_PLAN_MIN_N=<n> (measurement aid) replaces
# the rule by "from n columns on".
PLAN_MIN_N = int(os.environ["MISPMM_PLAN_MIN_N"]) if "MISPMM_PLAN_MIN_N" in os.environ else None
L2_BYTES = 4 << 20


def plan_pays(num_cols, n):
    """The footprint RULE (the prior, used when the choice cannot be measured): True where the B slice one XCD reads (num_cols
    rows x its column part of n / 8 columns, fp32) exceeds the L2."""
    if PLAN_MIN_N is not None:
        return n >= PLAN_MIN_N
    return n % 512 == 0 and num_cols * (n // 8) * 4 > L2_BYTES


AUTOTUNE = os.environ.get("MISPMM_AUTOTUNE", "1") != "0"   # MISPMM_AUTOTUNE=0: the footprint rule alone (measurement aid)


def autotune_pick(times_us, min_gain=0.02):
    """mispmm_autotune_pick: index of the candidate to take given its timings -- a pure function (candidate 0 = the default
    is kept unless another is at least min_gain faster)."""
    arr = (ctypes.c_float * len(times_us))(*[float(t) for t in times_us])
    return int(capi.lib().mispmm_autotune_pick(arr, len(times_us), ctypes.c_float(min_gain)))


def use_plan(a, n, acc="reference", stream=None):
    """Plan order or storage order for a product of `a` with a dense operand of n columns: MEASURED on the first product of that
    width (mispmm_csr_autotune_plan_f32: both candidates timed on scratch operands, ~1-2 ms) and remembered in a.tuned; the
    footprint rule where it cannot be measured (stream being captured, autotune switched off, MISPMM_PLAN_MIN_N given)."""
    if a.plan is None:
        return False
    if PLAN_MIN_N is not None or not AUTOTUNE:
        return plan_pays(a.num_cols, n)
    key = (int(n), acc)
    if key not in a.tuned:
        p = a.plan
        use, times = ctypes.c_int(0), (ctypes.c_float * 2)()
        st = capi.lib().mispmm_csr_autotune_plan_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.nnz, _p(a.row_ptrs), _p(a.col_idxs), _p(a.data),
                                                     a.uniform_row_nnz, _p(p.row_ptrs), _p(p.col_idxs), _p(p.data), _p(p.row_map), int(n),
                                                     capi.ACC_MODES[acc], 0, ctypes.byref(use), times)
        if st != capi.OK:                       # e.g. the stream is being captured: nothing measured, nothing remembered
            return plan_pays(a.num_cols, n)
        a.tuned[key] = (bool(use.value), (float(times[0]), float(times[1])))
    return a.tuned[key][0]


@dataclass
class DeviceCSR:
    num_rows: int
    num_cols: int
    nnz: int
    row_ptrs: torch.Tensor
    col_idxs: torch.Tensor
    data: torch.Tensor
    uniform_row_nnz: int = 0     # > 0: structure hint checked on the host when A was uploaded
    spans: torch.Tensor = None   # rows longest first, for the split kernel; built at upload for long-row matrices
    plan: CsrPlan = None         # rows in a clustered order, kept when the clustering cuts the B rows an XCD must fetch
    long_spans: int = 0          # leading positions of `spans` that hold the long rows (the split body of the two-body launch)
    spans_hybrid_only: bool = False   # short rows on average: the list is for the two-body launch only, never the split kernel
    tuned: dict = field(default_factory=dict)   # (n, acc) -> (use the plan?, (storage-order us, plan-order us)) as measured by use_plan

    @staticmethod
    def from_host(csr, device="cuda", spans=None, share_len=0, plan=None):
        """spans: True / False to build the span list of the split kernel or not; None = when the mean row holds 24 entries
        or more (where the library's kernel 0 takes the split kernel).  share_len: rows longer than this are dealt to the 4
        waves of a workgroup (0 = the library's default, 128).  plan: True / False to keep a clustered row order or not;
        None = when the matrix has short rows (no span list), at least 1024 rows, and the clustering cuts the distinct
        columns per row part by PLAN_MIN_GAIN or more (a once-per-upload analysis like the two above)."""
        hybrid_only = False
        if spans is None:
            spans, hybrid_only = wants_spans(csr.row_ptrs)
        sp_host = csr_spans_by_length(csr.row_ptrs, share_len) if spans else None
        sp = _dev_u32(sp_host.reshape(-1), device) if spans else None
        n_long = spans_long_count(sp_host) if spans else 0
        pl = None
        if plan or (plan is None and not spans and csr.num_rows >= 1024 and csr.nnz > 0):
            order, nat, clu = cluster_rows(csr, PLAN_PARTS)
            if plan or clu <= (1.0 - PLAN_MIN_GAIN) * nat:
                pc = permute_rows(csr, order)
                pl = CsrPlan(_dev_u32(pc.row_ptrs, device), _dev_u32(pc.col_idxs, device), _dev_f32(pc.data, device),
                             _dev_u32(order, device), PLAN_PARTS, nat, clu)
        return DeviceCSR(csr.num_rows, csr.num_cols, csr.nnz, _dev_u32(csr.row_ptrs, device),
                         _dev_u32(csr.col_idxs, device), _dev_f32(csr.data, device), uniform_row_nnz(csr.row_ptrs), sp, pl, n_long, hybrid_only)


@dataclass
class DeviceELL:
    num_rows: int
    num_cols: int
    width: int
    col_idxs: torch.Tensor
    data: torch.Tensor
    compact: tuple = None     # (nnz, rowPtrs, colIdxs, vals) of the occupied slots, when most of the ELL is padding

    @staticmethod
    def from_host(ell, device="cuda", compact=None):
        """compact: True / False to list the occupied slots (mispmm_ell_compact_host) or not; None = when more than half of
        the slots are padding."""
        if isinstance(ell, formats.ELLColMajor):
            ell = colmajor_ell_to_rowmajor(ell)
        cols = np.ascontiguousarray(ell.col_idxs, dtype=np.uint32).reshape(-1)
        vals = np.ascontiguousarray(ell.data, dtype=np.float32).reshape(-1)
        listed = None
        if compact is None:
            compact = cols.size > 0 and int(np.count_nonzero(cols == 0xFFFFFFFF)) * 2 > cols.size
        if compact:
            nnz = ctypes.c_uint32(0)
            head = (ell.num_rows, ell.width, cols.ctypes.data, vals.ctypes.data, ctypes.byref(nnz))
            capi.check(capi.lib().mispmm_ell_compact_host(*head, None, None, None))
            rp = np.zeros(ell.num_rows + 1, np.uint32)
            ci, va = np.zeros(max(nnz.value, 1), np.uint32), np.zeros(max(nnz.value, 1), np.float32)
            capi.check(capi.lib().mispmm_ell_compact_host(*head, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))
            listed = (nnz.value, _dev_u32(rp, device), _dev_u32(ci, device), _dev_f32(va, device), _row_spans(rp, device))
        return DeviceELL(ell.num_rows, ell.num_cols, ell.width, _dev_u32(ell.col_idxs, device),
                         _dev_f32(ell.data, device), listed)


@dataclass
class DeviceBSR:
    num_rows: int
    num_cols: int
    block_row_size: int
    block_col_size: int
    num_blocks: int
    block_row_ptrs: torch.Tensor
    block_col_idxs: torch.Tensor
    data: torch.Tensor       # float32 blocks, or int16-viewed bf16 bit patterns

    @staticmethod
    def from_host(bsr, device="cuda"):
        return DeviceBSR(bsr.num_rows, bsr.num_cols, bsr.block_row_size, bsr.block_col_size, bsr.num_blocks,
                         _dev_u32(bsr.block_row_ptrs, device), _dev_u32(bsr.block_col_idxs, device),
                         _dev_f32(bsr.data.reshape(-1), device))


@dataclass
class RowSpans:
    """One span per row of a COO / ELL / BSR list, longest first, on the device (mispmm_rows_split_f32 / mispmm_rows_hybrid_f32)."""
    spans: torch.Tensor
    long_spans: int      # leading positions that hold rows of more than HYBRID_ROW_LEN entries: the split shape's share of the two-body launch
    hybrid_only: bool    # short rows on average with a few long ones: the two-body launch or the format's own kernel, never the split kernel


def _row_spans(row_ptrs, device):
    """One span per row, longest first (the fp32 arithmetic of COO / ELL / BSR cannot deal a row to several waves), for a
    list of 24 entries per row or more; else None."""
    build, hybrid_only = wants_spans(row_ptrs)
    if not build:
        return None
    host = csr_spans_by_length(row_ptrs, 0xFFFFFFFF)
    return RowSpans(_dev_u32(host.reshape(-1), device), spans_long_count(host), hybrid_only)


def _rows_split(rs, num_rows, num_cols, nnz, col_idxs, data, b, c, acc, stream):
    """rs: RowSpans or None.  The two-body launch (mispmm_rows_hybrid_f32), else mispmm_rows_split_f32, if list and operands
    allow it; False = take the format's own entry point."""
    if rs is None:
        return False
    if 0 < rs.long_spans < num_rows and os.environ.get("MISPMM_NO_HYBRID") != "1":
        st = capi.lib().mispmm_rows_hybrid_f32(_stream_ptr(stream), num_rows, num_cols, nnz, _p(col_idxs), _p(data), _p(rs.spans), num_rows,
                                               rs.long_spans, _p(b), b.shape[1], _dense_ld(b), _p(c), _dense_ld(c), capi.ACC_MODES[acc])
        if st != capi.ERR_UNSUPPORTED:
            capi.check(st)
            return True
    if rs.hybrid_only:
        return False
    st = capi.lib().mispmm_rows_split_f32(_stream_ptr(stream), num_rows, num_cols, nnz, _p(col_idxs), _p(data), _p(rs.spans), num_rows,
                                          _p(b), b.shape[1], _dense_ld(b), _p(c), _dense_ld(c), capi.ACC_MODES[acc])
    if st == capi.ERR_UNSUPPORTED:
        return False
    capi.check(st)
    return True


@dataclass
class DeviceCOO:
    num_rows: int
    num_cols: int
    nnz: int
    row_idxs: torch.Tensor
    col_idxs: torch.Tensor
    data: torch.Tensor
    spans: "RowSpans" = None     # long rows: (row, start, end, 0) per row, longest first -- carries the row boundaries

    @staticmethod
    def from_host(coo, device="cuda"):
        order = np.lexsort((coo.col_idxs, coo.row_idxs))
        rows = np.asarray(coo.row_idxs)[order]
        row_ptrs = np.searchsorted(rows, np.arange(coo.num_rows + 1)).astype(np.uint32)
        return DeviceCOO(coo.num_rows, coo.num_cols, coo.nnz, _dev_u32(rows, device),
                         _dev_u32(coo.col_idxs[order], device), _dev_f32(coo.data[order], device), _row_spans(row_ptrs, device))


def _out(m, n, b, out):
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=b.device)
    _require_gpu(out)
    if out.shape != (m, n):
        raise ValueError(f"out has shape {tuple(out.shape)}, expected {(m, n)}")
    return out


def spmm_csr(a, b, out=None, kernel=0, acc="reference", stream=None, use_hint=True):
    """C = A @ B.  a: DeviceCSR, b: [K, N] float32 device tensor (row-major, any row stride).
    With the default kernel (0 / 5) a CSR whose rows all have the same length goes through
    mispmm_csr_uniform_f32 (no row-pointer fetch), one that carries `spans` (long rows) through mispmm_csr_split_f32 with
    its rows longest first; use_hint=False forces the general entry point."""
    _require_gpu(a.row_ptrs, b)
    if b.shape[0] != a.num_cols:
        raise ValueError(f"B has {b.shape[0]} rows, A has {a.num_cols} columns")
    n = b.shape[1]
    c = _out(a.num_rows, n, b, out)
    hints = use_hint and os.environ.get("MISPMM_NO_HINT") != "1"
    if hints and a.plan is not None and int(kernel) in (0, 5) and use_plan(a, n, acc, stream):
        # rows in the clustered order of the plan (same bits: every row keeps its entries in storage order)
        if _csr_plan(a, [b], [c], acc, stream):
            return c
    if use_hint and os.environ.get("MISPMM_NO_HINT") != "1" and a.uniform_row_nnz and int(kernel) in (0, 5) and a.num_cols * _dense_ld(b) * 4 <= 0x7FFFFFFF:
        capi.check(capi.lib().mispmm_csr_uniform_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.uniform_row_nnz,
                                                     _p(a.col_idxs), _p(a.data), _p(b), n, _dense_ld(b), _p(c),
                                                     _dense_ld(c), capi.ACC_MODES[acc]))
        return c
    # the split kernel with the rows longest first: kernel 6 always, kernel 0 / 5 where the library itself would split
    # (REFERENCE mode: up to 383 columns)
    wants_split = int(kernel) == 6 or (int(kernel) in (0, 5) and (acc == "fast" or n < 384))
    if use_hint and os.environ.get("MISPMM_NO_HINT") != "1" and a.spans is not None and wants_split:
        # kernel 0 / 5: the long rows by the split body and the short rows by the row-gather body of ONE launch, where the
        # shape has such a launch (else, and for kernel 6 by name, the split kernel on the whole list)
        if int(kernel) != 6 and 0 < a.long_spans < a.spans.numel() // 4 and os.environ.get("MISPMM_NO_HYBRID") != "1":
            st = capi.lib().mispmm_csr_hybrid_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.nnz, _p(a.col_idxs), _p(a.data),
                                                  _p(a.spans), a.spans.numel() // 4, a.long_spans, _p(b), n, _dense_ld(b), _p(c),
                                                  _dense_ld(c), capi.ACC_MODES[acc])
            if st != capi.ERR_UNSUPPORTED:
                capi.check(st)
                return c
        if a.spans_hybrid_only and int(kernel) != 6:
            st = capi.ERR_UNSUPPORTED        # short rows on average: the wave-per-row split kernel is not for this matrix
        else:
            st = capi.lib().mispmm_csr_split_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.nnz, _p(a.row_ptrs), _p(a.col_idxs),
                                                 _p(a.data), _p(a.spans), a.spans.numel() // 4, _p(b), n, _dense_ld(b), _p(c),
                                                 _dense_ld(c), capi.ACC_MODES[acc])
        if st != capi.ERR_UNSUPPORTED:  # operands that are not 16-byte vectors take the general entry point below
            capi.check(st)
            return c
    capi.check(capi.lib().mispmm_csr_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.nnz, _p(a.row_ptrs),
                                         _p(a.col_idxs), _p(a.data), _p(b), n, _dense_ld(b), _p(c), _dense_ld(c),
                                         int(kernel), capi.ACC_MODES[acc]))
    return c


def _csr_plan(a, bs, cs, acc, stream):
    """mispmm_csr_plan_f32 over the plan's arrays; False = the shape is not taken (B of 2 GiB or more)."""
    p = a.plan
    blist = (ctypes.c_void_p * len(bs))(*[b.data_ptr() for b in bs])
    clist = (ctypes.c_void_p * len(bs))(*[c.data_ptr() for c in cs])
    st = capi.lib().mispmm_csr_plan_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.nnz, _p(p.row_ptrs), _p(p.col_idxs), _p(p.data),
                                        a.uniform_row_nnz, _p(p.row_map), len(bs), blist, bs[0].shape[1], _dense_ld(bs[0]), clist,
                                        _dense_ld(cs[0]), capi.ACC_MODES[acc])
    if st == capi.ERR_UNSUPPORTED:
        return False
    capi.check(st)
    return True


def spmm_csr_batch(a, bs, outs=None, acc="reference", stream=None):
    """outs[i] = A @ bs[i] for a list of equally shaped dense operands, ONE launch per 16 of them
    (mispmm_csr_batch_f32).  Returns the list of results."""
    if not bs:
        return []
    n, ldb = bs[0].shape[1], _dense_ld(bs[0])
    for b in bs:
        _require_gpu(b)
        if b.shape != bs[0].shape or _dense_ld(b) != ldb:
            raise ValueError("batched operands must share one shape and row stride")
        if b.shape[0] != a.num_cols:
            raise ValueError(f"B has {b.shape[0]} rows, A has {a.num_cols} columns")
    if outs is None:
        outs = [torch.empty((a.num_rows, n), dtype=torch.float32, device=bs[0].device) for _ in bs]
    ldc = _dense_ld(outs[0])
    if a.spans is not None:
        # long rows: the library issues one launch per operand anyway (there is no batched split kernel); go through the
        # span list like spmm_csr does, so that both give the same bits in FAST mode as well
        for b, c in zip(bs, outs):
            spmm_csr(a, b, out=c, acc=acc, stream=stream)
        return outs
    if a.plan is not None and os.environ.get("MISPMM_NO_HINT") != "1" and use_plan(a, n, acc, stream):
        if _csr_plan(a, bs, outs, acc, stream):
            return outs
    blist = (ctypes.c_void_p * len(bs))(*[b.data_ptr() for b in bs])
    clist = (ctypes.c_void_p * len(bs))(*[c.data_ptr() for c in outs])
    capi.check(capi.lib().mispmm_csr_batch_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.nnz, _p(a.row_ptrs), _p(a.col_idxs),
                                               _p(a.data), a.uniform_row_nnz, len(bs), blist, n, ldb, clist, ldc,
                                               capi.ACC_MODES[acc]))
    return outs


@dataclass
class DeviceCSRTiles:
    """A CSR with rows of one width grouped into LDS tiles (mispmm_csr_tiles_host): rows that share B rows sit in one tile of
    <= 16 rows / <= 128 distinct columns, an entry names its column by its position in the tile's list."""
    num_rows: int
    num_cols: int
    nnz: int
    row_nnz: int
    num_tiles: int
    num_listed: int              # sum of the tiles' column lists = B-row slices staged per column part (nnz without sharing)
    tile_row_ptrs: torch.Tensor
    tile_col_ptrs: torch.Tensor
    tile_cols: torch.Tensor
    slots: torch.Tensor          # uint8 per entry, plan order
    data: torch.Tensor           # values, plan order
    row_map: torch.Tensor        # plan position -> C row

    @staticmethod
    def from_host(csr, device="cuda", max_rows=16, max_cols=128):
        w = uniform_row_nnz(csr.row_ptrs)
        if w == 0 or w > 16:
            raise ValueError("LDS tiles take rows of one width of 1..16 entries")
        l = capi.lib()
        rp = np.ascontiguousarray(csr.row_ptrs, dtype=np.uint32)
        ci = np.ascontiguousarray(csr.col_idxs, dtype=np.uint32)
        nt, nl = ctypes.c_uint32(0), ctypes.c_uint32(0)
        head = (csr.num_rows, csr.num_cols, rp.ctypes.data, ci.ctypes.data, int(max_rows), int(max_cols), ctypes.byref(nt), ctypes.byref(nl))
        capi.check(l.mispmm_csr_tiles_host(*head, None, None, None, None, None))
        trp, tcp = np.zeros(nt.value + 1, np.uint32), np.zeros(nt.value + 1, np.uint32)
        tc, order, slots = np.zeros(max(nl.value, 1), np.uint32), np.zeros(csr.num_rows, np.uint32), np.zeros(max(csr.nnz, 1), np.uint8)
        capi.check(l.mispmm_csr_tiles_host(*head, trp.ctypes.data, tcp.ctypes.data, tc.ctypes.data, order.ctypes.data, slots.ctypes.data))
        planned = permute_rows(csr, order)
        return DeviceCSRTiles(csr.num_rows, csr.num_cols, csr.nnz, w, nt.value, nl.value, _dev_u32(trp, device), _dev_u32(tcp, device),
                              _dev_u32(tc, device), torch.from_numpy(slots).to(device), _dev_f32(planned.data, device), _dev_u32(order, device))


def spmm_csr_tiles(a, b, out=None, acc="reference", stream=None):
    """C = A @ B with the B rows of every tile staged in LDS (mispmm_csr_lds_tile_f32): a: DeviceCSRTiles."""
    _require_gpu(a.tile_row_ptrs, b)
    if b.shape[0] != a.num_cols:
        raise ValueError(f"B has {b.shape[0]} rows, A has {a.num_cols} columns")
    n = b.shape[1]
    c = _out(a.num_rows, n, b, out)
    capi.check(capi.lib().mispmm_csr_lds_tile_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.row_nnz, a.num_tiles, _p(a.tile_row_ptrs),
                                                  _p(a.tile_col_ptrs), _p(a.tile_cols), _p(a.slots), _p(a.data), _p(a.row_map), _p(b), n,
                                                  _dense_ld(b), _p(c), _dense_ld(c), capi.ACC_MODES[acc]))
    return c


def spmm_ell(a, b, out=None, kernel=0, acc="reference", stream=None):
    _require_gpu(a.col_idxs, b)
    if b.shape[0] != a.num_cols:
        raise ValueError(f"B has {b.shape[0]} rows, A has {a.num_cols} columns")
    n = b.shape[1]
    c = _out(a.num_rows, n, b, out)
    if a.compact is not None and int(kernel) in (0, 1):   # mostly padding: multiply from the list of occupied slots
        nnz, rp, ci, va, spans = a.compact
        if _rows_split(spans, a.num_rows, a.num_cols, nnz, ci, va, b, c, acc, stream):
            return c
        st = capi.lib().mispmm_ell_compact_f32(_stream_ptr(stream), a.num_rows, a.num_cols, nnz, _p(rp), _p(ci), _p(va), _p(b), n,
                                               _dense_ld(b), _p(c), _dense_ld(c), capi.ACC_MODES[acc])
        if st != capi.ERR_UNSUPPORTED:
            capi.check(st)
            return c
    capi.check(capi.lib().mispmm_ell_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.width, _p(a.col_idxs),
                                         _p(a.data), _p(b), n, _dense_ld(b), _p(c), _dense_ld(c), int(kernel),
                                         capi.ACC_MODES[acc]))
    return c


def spmm_bsr(a, b, out=None, kernel=0, acc="reference", stream=None):
    _require_gpu(a.block_row_ptrs, b)
    if b.shape[0] != a.num_cols:
        raise ValueError(f"B has {b.shape[0]} rows, A has {a.num_cols} columns")
    n = b.shape[1]
    c = _out(a.num_rows, n, b, out)
    capi.check(capi.lib().mispmm_bsr_f32(_stream_ptr(stream), a.num_rows // a.block_row_size, a.num_cols,
                                         a.block_row_size, a.block_col_size, a.num_blocks, _p(a.block_row_ptrs),
                                         _p(a.block_col_idxs), _p(a.data), _p(b), n, _dense_ld(b), _p(c), _dense_ld(c),
                                         int(kernel), capi.ACC_MODES[acc]))
    return c


def bsr_nonzeros(bsr, device="cuda"):
    """The non-zero block entries of a host BSR as a DeviceCSR in the reference's order of addition
    (mispmm_bsr_nonzeros_host) -- the once-per-upload analysis step of the zero-skipping BSR path."""
    l = capi.lib()
    ptrs = np.ascontiguousarray(bsr.block_row_ptrs, dtype=np.uint32)
    cols = np.ascontiguousarray(bsr.block_col_idxs, dtype=np.uint32)
    data = np.ascontiguousarray(bsr.data, dtype=np.float32).reshape(-1)
    nnz = ctypes.c_uint32(0)
    args = (bsr.num_block_rows, bsr.block_row_size, bsr.block_col_size, bsr.num_blocks, ptrs.ctypes.data, cols.ctypes.data,
            data.ctypes.data, ctypes.byref(nnz))
    capi.check(l.mispmm_bsr_nonzeros_host(*args, None, None, None))
    rp = np.empty(bsr.num_rows + 1, dtype=np.uint32)
    ci = np.empty(max(1, nnz.value), dtype=np.uint32)
    va = np.empty(max(1, nnz.value), dtype=np.float32)
    capi.check(l.mispmm_bsr_nonzeros_host(*args, rp.ctypes.data, ci.ctypes.data, va.ctypes.data))
    return DeviceCSR(bsr.num_rows, bsr.num_cols, nnz.value, _dev_u32(rp, device), _dev_u32(ci[:nnz.value], device),
                     _dev_f32(va[:nnz.value], device), 0, _row_spans(rp, device))


def spmm_bsr_nonzeros(nz, b, out=None, acc="reference", stream=None):
    """C = A @ B from the non-zero list of a BSR (bsr_nonzeros): fp32 product, fp32 add in the reference's order."""
    _require_gpu(nz.row_ptrs, b)
    if b.shape[0] != nz.num_cols:
        raise ValueError(f"B has {b.shape[0]} rows, A has {nz.num_cols} columns")
    n = b.shape[1]
    c = _out(nz.num_rows, n, b, out)
    if _rows_split(nz.spans, nz.num_rows, nz.num_cols, nz.nnz, nz.col_idxs, nz.data, b, c, acc, stream):
        return c
    capi.check(capi.lib().mispmm_bsr_nonzeros_f32(_stream_ptr(stream), nz.num_rows, nz.num_cols, nz.nnz, _p(nz.row_ptrs),
                                                  _p(nz.col_idxs), _p(nz.data), _p(b), n, _dense_ld(b), _p(c), _dense_ld(c),
                                                  capi.ACC_MODES[acc]))
    return c


def coo_row_bounds(a, stream=None):
    """The (M+1)-entry row-boundary array of a row-sorted device COO (the once-per-upload analysis step)."""
    _require_gpu(a.row_idxs)
    ws = torch.empty(a.num_rows + 1, dtype=torch.int32, device=a.row_idxs.device)
    capi.check(capi.lib().mispmm_coo_row_bounds(_stream_ptr(stream), a.num_rows, a.nnz, _p(a.row_idxs), _p(ws)))
    return ws


def spmm_coo(a, b, out=None, kernel=0, acc="reference", stream=None, workspace=True):
    """workspace=True allocates the (M+1)-entry row-boundary scratch; False uses binary search; a tensor from
    coo_row_bounds() is used as is (pass kernel=2 to skip rebuilding it)."""
    _require_gpu(a.row_idxs, b)
    if isinstance(workspace, torch.Tensor):
        ws = workspace
    else:
        ws = torch.empty(a.num_rows + 1, dtype=torch.int32, device=b.device) if workspace else None
    if b.shape[0] != a.num_cols:
        raise ValueError(f"B has {b.shape[0]} rows, A has {a.num_cols} columns")
    n = b.shape[1]
    c = _out(a.num_rows, n, b, out)
    if workspace is not False and int(kernel) in (0, 1, 2) and _rows_split(a.spans, a.num_rows, a.num_cols, a.nnz, a.col_idxs, a.data, b, c, acc, stream):
        return c
    capi.check(capi.lib().mispmm_coo_f32(_stream_ptr(stream), a.num_rows, a.num_cols, a.nnz, _p(a.row_idxs),
                                         _p(a.col_idxs), _p(a.data), _p(b), n, _dense_ld(b), _p(c), _dense_ld(c),
                                         _p(ws), int(kernel), capi.ACC_MODES[acc]))
    return c


def f32_to_bf16(x, stream=None):
    """Device fp32 tensor -> int16 tensor of bf16 bit patterns (round to nearest even)."""
    _require_gpu(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    capi.check(capi.lib().mispmm_f32_to_bf16(_stream_ptr(stream), x.numel(), _p(x), _p(out)))
    return out


def bf16_to_f32(x, stream=None):
    _require_gpu(x)
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    capi.check(capi.lib().mispmm_bf16_to_f32(_stream_ptr(stream), x.numel(), _p(x), _p(out)))
    return out


def spmm_bsr_bf16(a, blocks_bf16, b_bf16, out_bf16=False, out=None, stream=None):
    """a: DeviceBSR (index arrays used), blocks_bf16 / b_bf16: int16 tensors of bf16 bits."""
    _require_gpu(blocks_bf16, b_bf16)
    n = b_bf16.shape[1]
    if out is None:
        out = torch.empty((a.num_rows, n), dtype=torch.int16 if out_bf16 else torch.float32, device=b_bf16.device)
    capi.check(capi.lib().mispmm_bsr_bf16(_stream_ptr(stream), a.num_rows // a.block_row_size, a.num_cols,
                                          a.block_row_size, a.block_col_size, a.num_blocks, _p(a.block_row_ptrs),
                                          _p(a.block_col_idxs), _p(blocks_bf16), _p(b_bf16), n, b_bf16.stride(0),
                                          _p(out), out.stride(0), int(bool(out_bf16))))
    return out


@dataclass
class DeviceBSRC:
    """Column-compacted block rows of a 16-row BSR (mispmm_bsr_compact_bf16_host)."""
    num_rows: int
    num_cols: int
    num_steps: int
    step_ptrs: torch.Tensor
    cols: torch.Tensor
    tiles: torch.Tensor      # int16 view of bf16 bits, [num_steps, 16, 32]

    @staticmethod
    def from_host(bsr, device="cuda"):
        l = capi.lib()
        ptrs = np.ascontiguousarray(bsr.block_row_ptrs, dtype=np.uint32)
        cols = np.ascontiguousarray(bsr.block_col_idxs, dtype=np.uint32)
        data = np.ascontiguousarray(bsr.data, dtype=np.float32).reshape(-1)
        n = ctypes.c_uint32(0)
        head = (bsr.num_block_rows, bsr.block_row_size, bsr.block_col_size, bsr.num_blocks, ptrs.ctypes.data, cols.ctypes.data,
                data.ctypes.data, ctypes.byref(n))
        capi.check(l.mispmm_bsr_compact_bf16_host(*head, None, None, None))
        sp = np.empty(bsr.num_block_rows + 1, dtype=np.uint32)
        cl = np.empty(max(1, n.value) * 32, dtype=np.uint32)
        tl = np.empty(max(1, n.value) * 512, dtype=np.uint16)
        capi.check(l.mispmm_bsr_compact_bf16_host(*head, sp.ctypes.data, cl.ctypes.data, tl.ctypes.data))
        return DeviceBSRC(bsr.num_rows, bsr.num_cols, n.value, _dev_u32(sp, device), _dev_u32(cl, device),
                          torch.from_numpy(tl.view(np.int16).copy()).to(device))


def spmm_bsrc_bf16(a, b_bf16, out_bf16=False, out=None, stream=None):
    """a: DeviceBSRC, b_bf16: int16 tensor of bf16 bits [K, N]; fp32 (or bf16) C."""
    _require_gpu(a.step_ptrs, b_bf16)
    n = b_bf16.shape[1]
    if out is None:
        out = torch.empty((a.num_rows, n), dtype=torch.int16 if out_bf16 else torch.float32, device=b_bf16.device)
    capi.check(capi.lib().mispmm_bsrc_bf16(_stream_ptr(stream), a.num_rows // 16, a.num_cols, a.num_steps, _p(a.step_ptrs),
                                           _p(a.cols), _p(a.tiles), _p(b_bf16), n, b_bf16.stride(0), _p(out), out.stride(0),
                                           int(bool(out_bf16))))
    return out


@dataclass
class DeviceBSRCSlots:
    """Column-compacted block rows of a 16-row BSR in fixed step slots, one workgroup per block row
    (mispmm_bsr_compact_slots_bf16_host)."""
    num_rows: int
    num_cols: int
    num_steps: int           # extent of cols / tiles: 4 slots per block row + extra steps
    used_steps: int          # steps that hold values (MFMA K steps executed)
    extra_ptrs: torch.Tensor
    cols: torch.Tensor
    tiles: torch.Tensor      # int16 view of bf16 bits, [num_steps, 16, 32]

    @staticmethod
    def from_host(bsr, device="cuda"):
        l = capi.lib()
        ptrs = np.ascontiguousarray(bsr.block_row_ptrs, dtype=np.uint32)
        cols = np.ascontiguousarray(bsr.block_col_idxs, dtype=np.uint32)
        data = np.ascontiguousarray(bsr.data, dtype=np.float32).reshape(-1)
        n, used = ctypes.c_uint32(0), ctypes.c_uint32(0)
        head = (bsr.num_block_rows, bsr.block_row_size, bsr.block_col_size, bsr.num_blocks, ptrs.ctypes.data, cols.ctypes.data,
                data.ctypes.data, ctypes.byref(n), ctypes.byref(used))
        capi.check(l.mispmm_bsr_compact_slots_bf16_host(*head, None, None, None))
        ep = np.empty(bsr.num_block_rows + 1, dtype=np.uint32)
        cl = np.empty(max(1, n.value) * 32, dtype=np.uint32)
        tl = np.empty(max(1, n.value) * 512, dtype=np.uint16)
        capi.check(l.mispmm_bsr_compact_slots_bf16_host(*head, ep.ctypes.data, cl.ctypes.data, tl.ctypes.data))
        return DeviceBSRCSlots(bsr.num_rows, bsr.num_cols, n.value, used.value, _dev_u32(ep, device), _dev_u32(cl, device),
                               torch.from_numpy(tl.view(np.int16).copy()).to(device))

    def operand_bytes(self):
        """Bytes of A one product must read: every slot's column list, the tiles of the used steps, the extra pointers."""
        return self.num_steps * 128 + self.used_steps * 1024 + (self.num_rows // 16 + 1) * 4


def spmm_bsrc_slots_bf16(a, b_bf16, out_bf16=False, out=None, stream=None):
    """a: DeviceBSRCSlots, b_bf16: int16 tensor of bf16 bits [K, N]; fp32 (or bf16) C."""
    _require_gpu(a.extra_ptrs, b_bf16)
    n = b_bf16.shape[1]
    if out is None:
        out = torch.empty((a.num_rows, n), dtype=torch.int16 if out_bf16 else torch.float32, device=b_bf16.device)
    capi.check(capi.lib().mispmm_bsrc_slots_bf16(_stream_ptr(stream), a.num_rows // 16, a.num_cols, a.num_steps, _p(a.extra_ptrs),
                                                 _p(a.cols), _p(a.tiles), _p(b_bf16), n, b_bf16.stride(0), _p(out), out.stride(0),
                                                 int(bool(out_bf16))))
    return out


def dense_transpose(x, stream=None):
    _require_gpu(x)
    x = x.contiguous()
    out = torch.empty((x.shape[1], x.shape[0]), dtype=torch.float32, device=x.device)
    capi.check(capi.lib().mispmm_dense_transpose_f32(_stream_ptr(stream), x.shape[0], x.shape[1], _p(x), _p(out)))
    return out


def colmajor_ell_to_rowmajor(ell):
    """Host conversion through the library's own helper (the path the C++ host uses)."""
    ridx = np.ascontiguousarray(ell.row_idxs, dtype=np.uint32)
    vals = np.ascontiguousarray(ell.data, dtype=np.float32)
    width = ctypes.c_uint32(0)
    l = capi.lib()
    capi.check(l.mispmm_ell_colmajor_to_rowmajor_host(ell.num_rows, ell.num_cols, ell.max_col_nnz, ridx.ctypes.data,
                                                      vals.ctypes.data, ctypes.byref(width), None, None))
    w = width.value
    cols = np.empty((ell.num_rows, w), dtype=np.uint32)
    out = np.empty((ell.num_rows, w), dtype=np.float32)
    capi.check(l.mispmm_ell_colmajor_to_rowmajor_host(ell.num_rows, ell.num_cols, ell.max_col_nnz, ridx.ctypes.data,
                                                      vals.ctypes.data, ctypes.byref(width), cols.ctypes.data,
                                                      out.ctypes.data))
    return formats.ELLRowMajor(ell.num_rows, ell.num_cols, ell.nnz, w, cols, out)


def shard_rows_by_nnz(row_ptrs, parts):
    rp = np.ascontiguousarray(row_ptrs, dtype=np.uint32)
    bounds = np.zeros(parts + 1, dtype=np.uint32)
    capi.check(capi.lib().mispmm_shard_rows_by_nnz_host(rp.shape[0] - 1, rp.ctypes.data, parts, bounds.ctypes.data))
    return bounds
