// Vendor-library cross-check: rocSPARSE generic SpMM in the role of the reference's cusparseTest
// (/root/reference/src/engine/cusparse.cu:9-57) -- except that the result IS compared.
#pragma once

#include "formats/dense.hpp"
#include "formats/matrix.hpp"

namespace cuspmm {

// Runs the vendor SpMM of `a` (device CSR or COO) with `b` into `c` (device, row-major) and reports
// prolog / kernel / epilog in microseconds.  Returns false when the library build has no vendor
// back end for this format (nothing is run).
template <typename DT, typename MT>
bool vendorTest(SparseMatrix<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *c, long &pro, long &kernel,
                long &epi);

}  // namespace cuspmm
