// ELL matrix in the reference's COLUMN-major layout (/root/reference/include/formats/sparse_ell.hpp:11-37):
// rowIdxs / data are [numCols x maxColNnz], padding row index 0xFFFFFFFF (text "-1"), padding value 0.
// The device copy additionally carries the row-major view the row-parallel HIP kernel consumes.
#pragma once

#include "formats/dense.hpp"

namespace cuspmm {

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

}  // namespace cuspmm
