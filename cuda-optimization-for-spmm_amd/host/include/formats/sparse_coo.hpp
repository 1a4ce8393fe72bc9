// COO matrix (/root/reference/include/formats/sparse_coo.hpp).  `.coo` text file: "rows cols nnz" then
// one "row col value" line per entry, sorted row-major.
#pragma once

#include "formats/dense.hpp"

namespace cuspmm {

template <typename _dataT, typename _metaT> class SparseMatrixCOO : public SparseMatrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT *rowIdxs = nullptr;
    MT *colIdxs = nullptr;
    // device only: (numRows + 1) scratch the COO kernel fills with row boundaries
    MT *rowBoundsWorkspace = nullptr;

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

}  // namespace cuspmm
