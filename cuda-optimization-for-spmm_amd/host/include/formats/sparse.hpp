// The four sparse containers of the engine in one place.  Each keeps the public members and method
// names of its counterpart in the reference (include/formats/sparse_{csr,coo,ell,bsr}.hpp) so code
// written against those classes compiles here; storage comes from the C ABI (pinned host or device
// memory, zero-filled), index type MT = uint32_t, value type DT = float or double.
#pragma once
#include <map>

#include "formats/dense.hpp"

namespace cuspmm {

// Device copy of the span list of a matrix whose rows are given by host row pointers (mispmm_csr_spans_by_length_host):
// rows longest first; shareLen = 0 lets rows of more than 128 entries become 4 chunks (CSR arithmetic only),
// 0xFFFFFFFF keeps one span per row (the fp32 arithmetic of COO / ELL / BSR).  Defined in sparse_csr.cpp.
// Does a list with these row boundaries get a span list, and is it for the two-body launch only?  24 entries per row or more
// on average: yes (the split kernel's domain).  Short rows on average but a row of 64 entries or more (tols4000: mean 2.2,
// longest 90): yes, for mispmm_csr_hybrid_f32 / mispmm_rows_hybrid_f32 only -- a lane group walks a row of L entries in
// L / 8 memory round trips, so those few rows decide the launch unless the split kernel's body takes them.
bool wantsRowSpans(uint32_t numRows, const uint32_t *rowPtrsHost, bool &hybridOnly);
// numLongSpans (optional): the leading positions that hold rows of more than 32 entries (mispmm_csr_spans_long_count_host)
uint32_t *uploadRowSpans(uint32_t numRows, const uint32_t *rowPtrsHost, uint32_t shareLen, uint32_t &numSpans, uint32_t *numLongSpans = nullptr);

// ----------------------------------------------------------------------------------------------------
// CSR matrix (/root/reference/include/formats/sparse_csr.hpp:11-39).
// `.csr` text file: "rows cols nnz" / rowPtrs (rows + 1) / colIdxs / values, one line each.
template <typename _dataT, typename _metaT> class SparseMatrixCSR : public SparseMatrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT *rowPtrs = nullptr;
    MT *colIdxs = nullptr;
    // > 0 on a device copy whose rows all hold exactly this many entries (checked on the host by
    // copy2Device): lets the wrapper take mispmm_csr_uniform_f32, which never reads rowPtrs
    MT uniformRowNnz = 0;
    // device copies made by copy2Device() of a long-row matrix (24 entries per row or more on average) only: the span
    // list mispmm_csr_split_f32 walks -- rows longest first, the longest as 4 chunks each (mispmm_csr_spans_by_length_host)
    MT *rowSpans = nullptr;
    MT numSpans = 0;
    // the leading span positions that hold the rows of more than 32 entries: mispmm_csr_hybrid_f32 gives those to the split
    // kernel's body and the others to the row-gather body of the same launch
    MT numLongSpans = 0;
    bool spansHybridOnly = false;  // short rows on average: the list serves the two-body launch, never the split kernel
    // device copies of a short-row matrix of 1024 rows or more whose rows cluster (mispmm_csr_cluster_rows_host cuts the
    // distinct columns per row part by 10 % or more): the same matrix with its rows in the clustered order, for
    // mispmm_csr_plan_f32 -- planRowMap[i] = the C row that array row i produces
    MT *planRowPtrs = nullptr;
    MT *planColIdxs = nullptr;
    _dataT *planData = nullptr;
    MT *planRowMap = nullptr;
    // plan order or storage order for a dense operand of N columns: measured on the first product of that width
    // (mispmm_csr_autotune_plan_f32 on scratch operands, outside every timed region) and remembered here; key = N * 2 + (FAST ? 1 : 0)
    std::map<uint64_t, bool> planChoice;
    bool usePlanFor(uint32_t N, int accMode);

    SparseMatrixCSR() = default;
    explicit SparseMatrixCSR(std::string filePath);
    SparseMatrixCSR(MT numRows, MT numCols, MT numNonZero, bool onDevice);
    ~SparseMatrixCSR() override;

    const char *formatName() const override { return "CSR"; }
    SparseMatrixCSR<DT, MT> *copy2Device();
    bool allocateSpace(bool onDevice);
    DenseMatrix<DT, MT> *toDense();

    template <typename U, typename M> friend std::ostream &operator<<(std::ostream &out, SparseMatrixCSR<U, M> &m);
};

// ----------------------------------------------------------------------------------------------------
// COO matrix (/root/reference/include/formats/sparse_coo.hpp).  `.coo` text file: "rows cols nnz" then
// one "row col value" line per entry, sorted row-major.
template <typename _dataT, typename _metaT> class SparseMatrixCOO : public SparseMatrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT *rowIdxs = nullptr;
    MT *colIdxs = nullptr;
    // device only: (numRows + 1) scratch the COO kernel fills with row boundaries
    MT *rowBoundsWorkspace = nullptr;
    bool rowBoundsReady = false;  // set once the boundaries of this device copy have been written
    // device only, for a COO of 24 entries per row or more: its rows as spans, longest first (mispmm_rows_split_f32)
    MT *rowSpans = nullptr;
    bool rowSpansHybridOnly = false;
    MT rowSpansLong = 0;  // its leading positions that hold rows of more than 32 entries (mispmm_rows_hybrid_f32)

    SparseMatrixCOO() = default;
    explicit SparseMatrixCOO(std::string filePath);
    SparseMatrixCOO(MT numRows, MT numCols, MT numNonZero, bool onDevice);
    ~SparseMatrixCOO() override;

    const char *formatName() const override { return "COO"; }
    SparseMatrixCOO<DT, MT> *copy2Device();
    bool allocateSpace(bool onDevice);
    DenseMatrix<DT, MT> *toDense();
    // true when entries are sorted by row (the order the converter writes and the kernel needs)
    bool isRowSorted() const;
};

// ----------------------------------------------------------------------------------------------------
// ELL matrix in the reference's COLUMN-major layout (/root/reference/include/formats/sparse_ell.hpp:11-37):
// rowIdxs / data are [numCols x maxColNnz], padding row index 0xFFFFFFFF (text "-1"), padding value 0.
// The device copy additionally carries the row-major view the row-parallel HIP kernel consumes.
template <typename _dataT, typename _metaT> class SparseMatrixELL : public SparseMatrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT *rowIdxs = nullptr;
    MT maxColNnz = 0;
    // device only, built by copy2Device(): [numRows x rowWidth] column indices / values, padded
    MT *rmColIdxs = nullptr;
    DT *rmData = nullptr;
    MT rowWidth = 0;
    // device only, built by copy2Device() when more than half of the slots are padding: the occupied slots per row in
    // slot order (mispmm_ell_compact_host) -- what mispmm_ell_compact_f32 multiplies from
    MT *cpRowPtrs = nullptr;
    MT *cpColIdxs = nullptr;
    DT *cpData = nullptr;
    MT cpCount = 0;
    MT *cpSpans = nullptr;  // ... and, from 24 occupied slots per row, its rows as spans, longest first
    bool cpSpansHybridOnly = false;
    MT cpSpansLong = 0;   // ... of which the leading positions hold rows of more than 32 occupied slots (mispmm_rows_hybrid_f32)

    SparseMatrixELL() = default;
    // files `<name>_rowind.ell` (header "rows cols nnz maxColNnz") and `<name>_values_colmajor.ell`
    SparseMatrixELL(std::string rowindPath, std::string valuesPath);
    SparseMatrixELL(MT numRows, MT numCols, MT numNonZero, MT maxColNnz, bool onDevice);
    ~SparseMatrixELL() override;

    const char *formatName() const override { return "ELL"; }
    bool allocateSpace(bool onDevice);
    SparseMatrixELL<DT, MT> *copy2Device();
    DenseMatrix<DT, MT> *toDense();
    size_t numSlots() const { return (size_t)this->numCols * (size_t)this->maxColNnz; }
};

// ----------------------------------------------------------------------------------------------------
// BSR matrix (/root/reference/include/formats/sparse_bsr.hpp:12-57).  `.bsr` text file:
// "rows cols storedElements blockRowSize blockColSize numBlocks" / blockRowPtrs / blockColIdxs /
// numBlocks * blockRowSize * blockColSize values (blocks row-major, block-CSR order).
template <typename _dataT, typename _metaT> class SparseMatrixBSR : public SparseMatrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT blockRowSize = 0;
    MT blockColSize = 0;
    MT numBlocks = 0;
    MT *blockRowPtrs = nullptr;
    MT *blockColIdxs = nullptr;
    MT numBlockRows = 0;  // numRows / blockRowSize
    MT numElements = 0;   // numBlocks * blockRowSize * blockColSize
    // device copies made by copy2Device() only (float): the block entries that are not zero, listed per row in
    // the order spmmBSRCpu adds them -- what the zero-skipping kernel 3 multiplies from (mispmm_bsr_nonzeros_*)
    MT *nzRowPtrs = nullptr;
    MT *nzColIdxs = nullptr;
    DT *nzVals = nullptr;
    MT nzCount = 0;
    MT *nzSpans = nullptr;  // from 24 list entries per row: the list's rows as spans, longest first (mispmm_rows_split_f32)
    bool nzSpansHybridOnly = false;
    MT nzSpansLong = 0;   // ... of which the leading positions hold rows of more than 32 entries (mispmm_rows_hybrid_f32)

    SparseMatrixBSR() = default;
    explicit SparseMatrixBSR(std::string filePath);
    SparseMatrixBSR(MT numRows, MT numCols, MT numNonZero, MT blockRowSize, MT blockColSize, MT numBlocks,
                    bool onDevice);
    SparseMatrixBSR(SparseMatrixBSR<DT, MT> *target, bool onDevice);
    ~SparseMatrixBSR() override;

    const char *formatName() const override { return "BSR"; }
    bool copyData(SparseMatrixBSR<DT, MT> *source, bool onDevice);
    SparseMatrixBSR<DT, MT> *copy2Device();
    void assertCheck();
    void assertSameShape(SparseMatrixBSR<DT, MT> *target);
    bool allocateSpace(bool onDevice);
    // Blocks with at least one non-zero, block columns ascending (the reference declares this and throws).
    static SparseMatrixBSR<DT, MT> *fromDense(DenseMatrix<DT, MT> *dense, MT blockRowSize, MT blockColSize);
    DenseMatrix<DT, MT> *toDense();
};

}  // namespace cuspmm
