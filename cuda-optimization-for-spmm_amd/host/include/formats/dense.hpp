// Dense matrix of the engine: B, C and the CPU reference result.
// Public surface of /root/reference/include/formats/dense.hpp:18-52 (constructors from a file, from a
// shape, from another matrix onto a side; copyData / copy2Device / copy2Host / toOrdering / save2File /
// allocateSpace / freeSpace), minus the cuSPARSE descriptor.
#pragma once

#include <string>

#include "commons.hpp"
#include "formats/matrix.hpp"

namespace cuspmm {

template <typename _dataT, typename _metaT> class DenseMatrix : public Matrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    DT *data = nullptr;
    ORDERING ordering = ORDERING::ROW_MAJOR;

    DenseMatrix() = default;
    // `dense.in`: "rows cols [anything]" then one text line per row (row-major)
    explicit DenseMatrix(std::string filePath);
    DenseMatrix(MT numRows, MT numCols, bool onDevice, ORDERING ordering = ORDERING::ROW_MAJOR);
    DenseMatrix(DenseMatrix<DT, MT> *source, bool onDevice);
    ~DenseMatrix() override;

    // Seeded synthetic operand (host, row-major): the counter-based generator of mispmm/synth.py.
    // mode 0: uniform on the 2^-23 grid of [-1, 1);  mode 1: multiples of 2^-8 ("exact").
    static DenseMatrix<DT, MT> *synthetic(MT numRows, MT numCols, uint64_t seed = 20241218, int mode = 0);

    bool copyData(DenseMatrix<DT, MT> *source);
    void assertSameShape(DenseMatrix<DT, MT> *target);
    DenseMatrix<DT, MT> *copy2Device();
    DenseMatrix<DT, MT> *copy2Host();
    // In-place change of storage order.  On the device this is one transpose kernel
    // (mispmm_dense_transpose_f32), not a round trip through the host.
    bool toOrdering(ORDERING newOrdering);
    bool save2File(std::string filePath);
    bool allocateSpace(bool onDevice);
    bool freeSpace();
    size_t numElements() const { return (size_t)this->numRows * (size_t)this->numCols; }
};

}  // namespace cuspmm
