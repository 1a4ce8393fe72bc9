#include "engine/vendor.hpp"

#include "formats/sparse_bsr.hpp"
#include "formats/sparse_coo.hpp"
#include "formats/sparse_csr.hpp"

namespace cuspmm {

template <typename DT, typename MT>
bool vendorTest(SparseMatrix<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *c, long &pro, long &kernel,
                long &epi) {
    pro = kernel = epi = 0;
    if constexpr (!std::is_same_v<DT, float>) {
        return false;
    } else {
        assert(a->onDevice && b->onDevice && c->onDevice);
        b->toOrdering(ORDERING::ROW_MAJOR);
        double p = 0, k = 0, e = 0;
        int status = MISPMM_ERR_UNSUPPORTED;
        if (auto *csr = dynamic_cast<SparseMatrixCSR<DT, MT> *>(a)) {
            status = mispmm_vendor_spmm_f32(nullptr, MISPMM_VENDOR_CSR, csr->numRows, csr->numCols, csr->numNonZero, 0,
                                            csr->rowPtrs, csr->colIdxs, csr->data, b->data, b->numCols, b->numCols,
                                            c->data, c->numCols, &p, &k, &e);
        } else if (auto *coo = dynamic_cast<SparseMatrixCOO<DT, MT> *>(a)) {
            status = mispmm_vendor_spmm_f32(nullptr, MISPMM_VENDOR_COO, coo->numRows, coo->numCols, coo->numNonZero, 0,
                                            coo->rowIdxs, coo->colIdxs, coo->data, b->data, b->numCols, b->numCols,
                                            c->data, c->numCols, &p, &k, &e);
        } else if (auto *bsr = dynamic_cast<SparseMatrixBSR<DT, MT> *>(a)) {
            if (bsr->blockRowSize != bsr->blockColSize) return false;
            status = mispmm_vendor_spmm_f32(nullptr, MISPMM_VENDOR_BSR, bsr->numRows, bsr->numCols, bsr->numBlocks,
                                            bsr->blockRowSize, bsr->blockRowPtrs, bsr->blockColIdxs, bsr->data, b->data,
                                            b->numCols, b->numCols, c->data, c->numCols, &p, &k, &e);
        } else {
            return false;
        }
        mispmmCheckError(status);
        pro = (long)p;
        kernel = (long)k;
        epi = (long)e;
        return true;
    }
}

template bool vendorTest<float, uint32_t>(SparseMatrix<float, uint32_t> *, DenseMatrix<float, uint32_t> *,
                                          DenseMatrix<float, uint32_t> *, long &, long &, long &);
template bool vendorTest<double, uint32_t>(SparseMatrix<double, uint32_t> *, DenseMatrix<double, uint32_t> *,
                                           DenseMatrix<double, uint32_t> *, long &, long &, long &);

}  // namespace cuspmm
