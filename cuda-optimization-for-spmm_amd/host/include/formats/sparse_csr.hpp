// CSR matrix (/root/reference/include/formats/sparse_csr.hpp:11-39).
// `.csr` text file: "rows cols nnz" / rowPtrs (rows + 1) / colIdxs / values, one line each.
#pragma once

#include "formats/dense.hpp"

namespace cuspmm {

template <typename _dataT, typename _metaT> class SparseMatrixCSR : public SparseMatrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT *rowPtrs = nullptr;
    MT *colIdxs = nullptr;

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

}  // namespace cuspmm
