// BSR matrix (/root/reference/include/formats/sparse_bsr.hpp:12-57).  `.bsr` text file:
// "rows cols storedElements blockRowSize blockColSize numBlocks" / blockRowPtrs / blockColIdxs /
// numBlocks * blockRowSize * blockColSize values (blocks row-major, block-CSR order).
#pragma once

#include "formats/dense.hpp"

namespace cuspmm {

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
