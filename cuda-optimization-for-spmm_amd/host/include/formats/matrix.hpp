// Base classes of the format layer (same public members as /root/reference/include/formats/matrix.hpp:
// numRows / numCols / onDevice, and numNonZero / data for sparse matrices).
#pragma once

#include <cstdint>

#include "hip_utils.hpp"

namespace cuspmm {

enum ORDERING {
    ROW_MAJOR,
    COL_MAJOR,
};

// Buffers are pinned host memory (onDevice == false) or device memory, both obtained zero-filled
// through the C ABI, and owned by the matrix object.
template <typename T> T *allocateBuffer(size_t count, bool onDevice);
void releaseBuffer(void *ptr, bool onDevice);
void copyBuffer(void *dst, bool dstOnDevice, const void *src, bool srcOnDevice, size_t bytes);

template <typename _dataT, typename _metaT> class Matrix {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT numRows = 0, numCols = 0;
    bool onDevice = false;
    virtual ~Matrix() = default;
};

template <typename _dataT, typename _metaT> class SparseMatrix : public Matrix<_dataT, _metaT> {
  public:
    using DT = _dataT;
    using MT = _metaT;
    MT numNonZero = 0;
    DT *data = nullptr;
    // short format tag used in records ("CSR", "COO", "ELL", "BSR")
    virtual const char *formatName() const = 0;
};

}  // namespace cuspmm
