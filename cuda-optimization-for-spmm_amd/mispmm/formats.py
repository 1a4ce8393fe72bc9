"""Host-side sparse/dense containers and the reference's on-disk text formats.

Python mirror (numpy only) of the data layer either side of the SpMM hot path:
the MatrixMarket inputs under the reference's data/ and the text wire formats
its C++ constructors parse.  Written from scratch; the wire formats are those of
  dense.in            /root/reference/src/formats/dense.cu:9-36
  *.csr               /root/reference/src/formats/sparse_csr.cu:12-51
  *.coo               /root/reference/src/formats/sparse_coo.cu (header + "r c v" lines)
  *.bsr               /root/reference/src/formats/sparse_bsr.cu:17-61
  *_rowind.ell / *_values_colmajor.ell  (column-major ELL, the pair the CLI loads)
                      /root/reference/src/formats/sparse_ell.cu:12-53
  *_colind.ell / *_values.ell           (row-major ELL, written by the converter,
                      /root/reference/utils/python_utils/convert_mtx.py:198-239)
and the writers reproduce what utils/python_utils/convert_mtx.py emits.

Index arrays are uint32 (the reference's MT), values float32 (its DT) unless a
caller asks for float64.  ELL padding index is 0xFFFFFFFF (text "-1").
"""
from dataclasses import dataclass

import numpy as np

ELL_PAD = np.uint32(0xFFFFFFFF)


# --------------------------------------------------------------------------
# containers
# --------------------------------------------------------------------------
@dataclass
class Dense:
    data: np.ndarray  # [num_rows, num_cols], row-major

    @property
    def num_rows(self):
        return self.data.shape[0]

    @property
    def num_cols(self):
        return self.data.shape[1]


@dataclass
class CSR:
    num_rows: int
    num_cols: int
    row_ptrs: np.ndarray
    col_idxs: np.ndarray
    data: np.ndarray

    @property
    def nnz(self):
        return int(self.col_idxs.shape[0])

    def to_dense(self):
        d = np.zeros((self.num_rows, self.num_cols), dtype=self.data.dtype)
        rows = np.repeat(np.arange(self.num_rows), np.diff(self.row_ptrs.astype(np.int64)))
        d[rows, self.col_idxs.astype(np.int64)] = self.data
        return d


@dataclass
class COO:
    num_rows: int
    num_cols: int
    row_idxs: np.ndarray
    col_idxs: np.ndarray
    data: np.ndarray

    @property
    def nnz(self):
        return int(self.data.shape[0])


@dataclass
class ELLColMajor:
    """The reference's SparseMatrixELL: one line per column of A."""
    num_rows: int
    num_cols: int
    nnz: int
    max_col_nnz: int
    row_idxs: np.ndarray  # [num_cols, max_col_nnz] uint32, pad ELL_PAD
    data: np.ndarray      # [num_cols, max_col_nnz]


@dataclass
class ELLRowMajor:
    """Row-major ELL: what the row-parallel HIP kernel consumes."""
    num_rows: int
    num_cols: int
    nnz: int
    width: int
    col_idxs: np.ndarray  # [num_rows, width] uint32, pad ELL_PAD
    data: np.ndarray      # [num_rows, width]


@dataclass
class BSR:
    num_rows: int
    num_cols: int
    nnz: int               # stored elements incl. explicit zeros (convert_mtx.py:41)
    block_row_size: int
    block_col_size: int
    block_row_ptrs: np.ndarray
    block_col_idxs: np.ndarray
    data: np.ndarray       # [num_blocks, block_row_size, block_col_size]

    @property
    def num_blocks(self):
        return int(self.block_col_idxs.shape[0])

    @property
    def num_block_rows(self):
        return self.num_rows // self.block_row_size

    def to_dense(self):
        d = np.zeros((self.num_rows, self.num_cols), dtype=self.data.dtype)
        br, bc = self.block_row_size, self.block_col_size
        for i in range(self.num_block_rows):
            for b in range(int(self.block_row_ptrs[i]), int(self.block_row_ptrs[i + 1])):
                j = int(self.block_col_idxs[b])
                d[i * br:(i + 1) * br, j * bc:(j + 1) * bc] = self.data[b]
        return d


# --------------------------------------------------------------------------
# MatrixMarket (coordinate) reader, from scratch
# --------------------------------------------------------------------------
def read_mtx(path):
    """Read a MatrixMarket *coordinate* file -> COO in file order with symmetric
    entries expanded (mirror entries appended after the stored ones, diagonal
    not duplicated).  Values are float64; `pattern` entries become 1.0.
    Returns (COO, field) where field is 'real' | 'integer' | 'pattern'."""
    with open(path, "r") as f:
        header = f.readline().split()
        if len(header) < 5 or header[0] != "%%MatrixMarket" or header[1].lower() != "matrix":
            raise ValueError(f"{path}: not a MatrixMarket matrix file")
        layout, field, symmetry = header[2].lower(), header[3].lower(), header[4].lower()
        if layout != "coordinate":
            raise ValueError(f"{path}: only coordinate layout is supported, got {layout}")
        if field not in ("real", "integer", "pattern", "double"):
            raise ValueError(f"{path}: unsupported field {field}")
        if symmetry not in ("general", "symmetric", "skew-symmetric"):
            raise ValueError(f"{path}: unsupported symmetry {symmetry}")
        line = f.readline()
        while line.startswith("%") or not line.strip():
            line = f.readline()
        rows, cols, entries = (int(x) for x in line.split()[:3])
        body = np.loadtxt(f, dtype=np.float64, ndmin=2, comments="%") if entries else np.zeros((0, 3))
    if body.shape[0] != entries:
        raise ValueError(f"{path}: expected {entries} entries, found {body.shape[0]}")
    r = body[:, 0].astype(np.int64) - 1
    c = body[:, 1].astype(np.int64) - 1
    v = np.ones(entries, dtype=np.float64) if field == "pattern" else body[:, 2].astype(np.float64)
    if symmetry != "general":
        off = r != c
        sign = -1.0 if symmetry == "skew-symmetric" else 1.0
        r, c, v = (np.concatenate([r, c[off]]), np.concatenate([c, r[off]]),
                   np.concatenate([v, sign * v[off]]))
    if entries and (r.min() < 0 or c.min() < 0 or r.max() >= rows or c.max() >= cols):
        raise ValueError(f"{path}: index out of range")
    return COO(rows, cols, r.astype(np.uint32), c.astype(np.uint32), v), field


# --------------------------------------------------------------------------
# conversions
# --------------------------------------------------------------------------
def coo_to_csr(coo, dtype=np.float32):
    """Row-major sort, duplicates summed (what scipy's coo.tocsr() yields)."""
    r = coo.row_idxs.astype(np.int64)
    c = coo.col_idxs.astype(np.int64)
    v = coo.data.astype(np.float64)
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    if r.size:
        key = r * coo.num_cols + c
        first = np.concatenate([[True], key[1:] != key[:-1]])
        seg = np.cumsum(first) - 1
        v = np.bincount(seg, weights=v, minlength=int(seg[-1]) + 1)
        r, c = r[first], c[first]
    row_ptrs = np.zeros(coo.num_rows + 1, dtype=np.int64)
    np.add.at(row_ptrs, r + 1, 1)
    row_ptrs = np.cumsum(row_ptrs)
    return CSR(coo.num_rows, coo.num_cols, row_ptrs.astype(np.uint32), c.astype(np.uint32), v.astype(dtype))


def csr_to_coo(csr):
    rows = np.repeat(np.arange(csr.num_rows, dtype=np.uint32), np.diff(csr.row_ptrs.astype(np.int64)))
    return COO(csr.num_rows, csr.num_cols, rows, csr.col_idxs.copy(), csr.data.copy())


def csr_to_ell_rowmajor(csr, width=None):
    counts = np.diff(csr.row_ptrs.astype(np.int64))
    w = int(counts.max()) if counts.size and width is None else int(width or 0)
    if counts.size and counts.max() > w:
        raise ValueError("ELL width smaller than the longest row")
    col = np.full((csr.num_rows, w), ELL_PAD, dtype=np.uint32)
    val = np.zeros((csr.num_rows, w), dtype=csr.data.dtype)
    rows = np.repeat(np.arange(csr.num_rows), counts)
    slot = np.arange(csr.nnz) - np.repeat(csr.row_ptrs[:-1].astype(np.int64), counts)
    col[rows, slot] = csr.col_idxs
    val[rows, slot] = csr.data
    return ELLRowMajor(csr.num_rows, csr.num_cols, csr.nnz, w, col, val)


def csr_to_ell_colmajor(csr, reference_width=False):
    """Column-major ELL (one padded line per column of A, rows ascending).
    reference_width=True sizes the padding like the reference converter does --
    by the longest ROW (convert_mtx.py:252 takes getnnz(axis=1) of the CSC) --
    so files match its output; False uses the true longest column."""
    r = np.repeat(np.arange(csr.num_rows, dtype=np.int64), np.diff(csr.row_ptrs.astype(np.int64)))
    c = csr.col_idxs.astype(np.int64)
    order = np.lexsort((r, c))
    r, c, v = r[order], c[order], csr.data[order]
    col_counts = np.bincount(c, minlength=csr.num_cols)
    row_counts = np.diff(csr.row_ptrs.astype(np.int64))
    w = int(row_counts.max() if reference_width else col_counts.max()) if csr.nnz else 0
    if csr.nnz and col_counts.max() > w:
        raise ValueError("reference-width column-major ELL cannot hold the longest column")
    col_start = np.concatenate([[0], np.cumsum(col_counts)[:-1]])
    slot = np.arange(csr.nnz) - np.repeat(col_start, col_counts)
    ridx = np.full((csr.num_cols, w), ELL_PAD, dtype=np.uint32)
    val = np.zeros((csr.num_cols, w), dtype=csr.data.dtype)
    ridx[c, slot] = r.astype(np.uint32)
    val[c, slot] = v
    return ELLColMajor(csr.num_rows, csr.num_cols, csr.nnz, w, ridx, val)


def ell_colmajor_to_csr(ell):
    """Row-parallel view of a column-major ELL.  Within each row the entries keep
    ascending column order, then ascending slot -- the order in which the
    reference's spmmELLCpu (spmm_ell.cpp:16-29) accumulates into that row."""
    valid = ell.row_idxs.astype(np.int32) >= 0
    cols, slots = np.nonzero(valid)            # C order: column, then slot
    rows = ell.row_idxs[cols, slots].astype(np.int64)
    vals = ell.data[cols, slots]
    order = np.argsort(rows, kind="stable")
    rows, cols, vals = rows[order], cols[order], vals[order]
    row_ptrs = np.zeros(ell.num_rows + 1, dtype=np.int64)
    np.add.at(row_ptrs, rows + 1, 1)
    return CSR(ell.num_rows, ell.num_cols, np.cumsum(row_ptrs).astype(np.uint32),
               cols.astype(np.uint32), vals.copy())


def ell_colmajor_to_rowmajor(ell):
    return csr_to_ell_rowmajor(ell_colmajor_to_csr(ell))


def csr_to_bsr(csr, block_row_size, block_col_size=None):
    """Blocks with at least one stored entry, block-row-major, block columns
    ascending, explicit zeros inside blocks (scipy tobsr semantics)."""
    br = int(block_row_size)
    bc = int(block_col_size or block_row_size)
    if csr.num_rows % br or csr.num_cols % bc:
        raise ValueError("matrix shape is not a multiple of the block shape")
    r = np.repeat(np.arange(csr.num_rows, dtype=np.int64), np.diff(csr.row_ptrs.astype(np.int64)))
    c = csr.col_idxs.astype(np.int64)
    nbc = csr.num_cols // bc
    key = (r // br) * nbc + (c // bc)
    ukeys, inv = np.unique(key, return_inverse=True)
    data = np.zeros((ukeys.shape[0], br, bc), dtype=csr.data.dtype)
    data[inv, r % br, c % bc] = csr.data
    brow = ukeys // nbc
    ptrs = np.zeros(csr.num_rows // br + 1, dtype=np.int64)
    np.add.at(ptrs, brow + 1, 1)
    return BSR(csr.num_rows, csr.num_cols, int(data.size), br, bc, np.cumsum(ptrs).astype(np.uint32),
               (ukeys % nbc).astype(np.uint32), data)


# --------------------------------------------------------------------------
# text readers (the formats the reference's C++ constructors parse)
# --------------------------------------------------------------------------
def _tokens(line, count, dtype):
    arr = np.array(line.split()[:count], dtype=np.float64 if dtype is float else np.int64)
    if arr.shape[0] != count:
        raise ValueError(f"expected {count} values, found {arr.shape[0]}")
    return arr


def read_dense(path, dtype=np.float32):
    """`rows cols [nnz]` header (rest of the line ignored, dense.cu:23-24) then
    one text line per row."""
    with open(path, "r") as f:
        head = f.readline().split()
        rows, cols = int(head[0]), int(head[1])
        data = np.empty((rows, cols), dtype=np.float64)
        for i in range(rows):
            data[i] = _tokens(f.readline(), cols, float)
    return Dense(data.astype(dtype))


def read_csr(path, dtype=np.float32):
    with open(path, "r") as f:
        rows, cols, nnz = (int(x) for x in f.readline().split()[:3])
        row_ptrs = _tokens(f.readline(), rows + 1, int)
        col_idxs = _tokens(f.readline(), nnz, int)
        data = _tokens(f.readline(), nnz, float)
    return CSR(rows, cols, row_ptrs.astype(np.uint32), col_idxs.astype(np.uint32), data.astype(dtype))


def read_coo(path, dtype=np.float32):
    with open(path, "r") as f:
        rows, cols, nnz = (int(x) for x in f.readline().split()[:3])
        body = np.loadtxt(f, dtype=np.float64, ndmin=2) if nnz else np.zeros((0, 3))
    if body.shape[0] != nnz:
        raise ValueError(f"{path}: expected {nnz} entries, found {body.shape[0]}")
    return COO(rows, cols, body[:, 0].astype(np.uint32), body[:, 1].astype(np.uint32), body[:, 2].astype(dtype))


def read_bsr(path, dtype=np.float32):
    with open(path, "r") as f:
        rows, cols, nnz, br, bc, nblocks = (int(x) for x in f.readline().split()[:6])
        ptrs = _tokens(f.readline(), rows // br + 1, int)
        idxs = _tokens(f.readline(), nblocks, int)
        vals = np.array(f.read().split(), dtype=np.float64)
    if vals.shape[0] != nblocks * br * bc:
        raise ValueError(f"{path}: expected {nblocks * br * bc} block values, found {vals.shape[0]}")
    return BSR(rows, cols, nnz, br, bc, ptrs.astype(np.uint32), idxs.astype(np.uint32),
               vals.reshape(nblocks, br, bc).astype(dtype))


def _read_ell_pair(index_path, values_path, lines_from_cols, dtype):
    with open(index_path, "r") as f:
        rows, cols, nnz, width = (int(x) for x in f.readline().split()[:4])
        n = cols if lines_from_cols else rows
        idx = np.array(f.read().split(), dtype=np.int64)
    with open(values_path, "r") as f:
        val = np.array(f.read().split(), dtype=np.float64)
    if idx.shape[0] != n * width or val.shape[0] != n * width:
        raise ValueError(f"{index_path}: expected {n * width} ELL slots")
    # "-1" -> 0xFFFFFFFF exactly as `istream >> uint32_t` wraps it (sparse_ell.cu:38-43)
    return rows, cols, nnz, width, (idx & 0xFFFFFFFF).astype(np.uint32).reshape(n, width), \
        val.reshape(n, width).astype(dtype)


def read_ell_colmajor(rowind_path, values_path, dtype=np.float32):
    rows, cols, nnz, w, idx, val = _read_ell_pair(rowind_path, values_path, True, dtype)
    return ELLColMajor(rows, cols, nnz, w, idx, val)


def read_ell_rowmajor(colind_path, values_path, dtype=np.float32):
    rows, cols, nnz, w, idx, val = _read_ell_pair(colind_path, values_path, False, dtype)
    return ELLRowMajor(rows, cols, nnz, w, idx, val)


# --------------------------------------------------------------------------
# text writers (what utils/python_utils/convert_mtx.py emits)
# --------------------------------------------------------------------------
def _fmt(values, integer):
    if integer:
        return " ".join(str(int(v)) for v in values)
    return " ".join(repr(float(v)) for v in values)


def write_dense(path, dense, integer=False):
    d = np.asarray(dense.data if isinstance(dense, Dense) else dense)
    with open(path, "w") as f:
        f.write(f"{d.shape[0]} {d.shape[1]} {int(np.count_nonzero(d))}\n")
        for row in d:
            f.write(_fmt(row, integer) + "\n")


def write_csr(path, csr, integer=False):
    with open(path, "w") as f:
        f.write(f"{csr.num_rows} {csr.num_cols} {csr.nnz}\n")
        f.write(" ".join(map(str, csr.row_ptrs.tolist())) + "\n")
        f.write(" ".join(map(str, csr.col_idxs.tolist())) + "\n")
        f.write(_fmt(csr.data, integer) + "\n")


def write_coo(path, coo, integer=False):
    order = np.lexsort((coo.col_idxs, coo.row_idxs))
    with open(path, "w") as f:
        f.write(f"{coo.num_rows} {coo.num_cols} {coo.nnz}\n")
        for i in order:
            f.write(f"{int(coo.row_idxs[i])} {int(coo.col_idxs[i])} {_fmt([coo.data[i]], integer)}\n")


def write_bsr(path, bsr, integer=False):
    with open(path, "w") as f:
        f.write(f"{bsr.num_rows} {bsr.num_cols} {bsr.nnz} {bsr.block_row_size} {bsr.block_col_size} "
                f"{bsr.num_blocks}\n")
        f.write(" ".join(map(str, bsr.block_row_ptrs.tolist())) + "\n")
        f.write(" ".join(map(str, bsr.block_col_idxs.tolist())) + "\n")
        for block in bsr.data:
            f.write(_fmt(block.reshape(-1), integer) + "\n")


def _write_ell(index_path, values_path, header, idx, val, integer):
    with open(index_path, "w") as f:
        f.write(header + "\n")
        for line in idx.astype(np.int32):   # 0xFFFFFFFF -> "-1"
            f.write(" ".join(map(str, line.tolist())) + "\n")
    with open(values_path, "w") as f:
        pad = idx == ELL_PAD
        for line, p in zip(val, pad):
            # the reference pads values with the int literal 0 (convert_mtx.py:210,255)
            f.write(" ".join("0" if pp else _fmt([v], integer) for v, pp in zip(line, p)) + "\n")


def write_ell_colmajor(rowind_path, values_path, ell, integer=False):
    _write_ell(rowind_path, values_path, f"{ell.num_rows} {ell.num_cols} {ell.nnz} {ell.max_col_nnz}",
               ell.row_idxs, ell.data, integer)


def write_ell_rowmajor(colind_path, values_path, ell, integer=False):
    _write_ell(colind_path, values_path, f"{ell.num_rows} {ell.num_cols} {ell.nnz} {ell.width}",
               ell.col_idxs, ell.data, integer)
