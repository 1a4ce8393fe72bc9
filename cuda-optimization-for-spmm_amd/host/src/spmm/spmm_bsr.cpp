// BSR: sequential CPU engine (kernel 0) and the HIP wrapper.
#include "engine/engine_bsr.hpp"
#include "engine/wrapper_common.hpp"

namespace cuspmm {

// Kernel 0: block rows, then blocks in storage order, then block element (i, j), each adding
// blk[i][j] * B[colBase + j, :] into C[rowBase + i, :] in DT
// (/root/reference/src/spmm/bsr/spmm_bsr.cpp:17-38).  C arrives zero-filled.
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmBSRCpu(SparseMatrixBSR<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc) {
    assert(!ma->onDevice && !mb->onDevice && !mc->onDevice);
    mb->toOrdering(ORDERING::ROW_MAJOR);
    const size_t n = mb->numCols;
    const MT bR = ma->blockRowSize, bC = ma->blockColSize;
    for (MT R = 0; R < ma->numBlockRows; ++R) {
        for (MT b = ma->blockRowPtrs[R]; b < ma->blockRowPtrs[R + 1]; ++b) {
            const DT *blk = ma->data + (size_t)b * bR * bC;
            const size_t colBase = (size_t)ma->blockColIdxs[b] * bC;
            for (MT i = 0; i < bR; ++i) {
                DT *crow = mc->data + ((size_t)R * bR + i) * n;
                for (MT j = 0; j < bC; ++j) {
                    const DT v = blk[(size_t)i * bC + j];
                    const DT *brow = mb->data + (colBase + j) * n;
                    for (size_t k = 0; k < n; ++k) {
                        const DT prod = v * brow[k];
                        crow[k] += prod;
                    }
                }
            }
        }
    }
    return mc;
}

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmBSRWrapper(int kernelNum, SparseMatrixBSR<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        assert(a->onDevice && b->onDevice);
        b->toOrdering(ORDERING::ROW_MAJOR);
        const double n = b->numCols;
        const WrapperShape shape{"BSR", a->numRows, a->numCols, a->numNonZero, 2.0 * a->numElements * n,
                                 a->numElements * 4.0 + a->numBlocks * 4.0 + (a->numBlockRows + 1.0) * 4 +
                                     a->numCols * n * 4 + a->numRows * n * 4};
        // kernel 2 (fp32 MFMA) has fused numerics only; every other id follows AccT
        const int acc = kernelNum == 2 ? MISPMM_ACC_FAST : accModeOf<AccT>();
        if (kernelNum == 3) {
            // the zero-skipping kernel: the non-zero list built at upload (same bits as the CPU engine unless an explicit
            // zero of A meets an Inf / NaN of B: include/mispmm.h, mispmm_bsr_nonzeros_*); flops and bytes of the list
            const WrapperShape nzShape{"BSR", a->numRows, a->numCols, a->numNonZero, 2.0 * a->nzCount * n,
                                       a->nzCount * 8.0 + (a->numRows + 1.0) * 4 + a->numCols * n * 4 + a->numRows * n * 4};
            return runWrapper<DT, MT>(nzShape, kernelNum, b, ref, [&](float *c, uint32_t ldc) {
                return mispmm_bsr_nonzeros_f32(nullptr, a->numRows, a->numCols, a->nzCount, a->nzRowPtrs, a->nzColIdxs, a->nzVals,
                                               b->data, b->numCols, b->numCols, c, ldc, acc);
            });
        }
        return runWrapper<DT, MT>(shape, kernelNum, b, ref, [&](float *c, uint32_t ldc) {
            return mispmm_bsr_f32(nullptr, a->numBlockRows, a->numCols, a->blockRowSize, a->blockColSize, a->numBlocks,
                                  a->blockRowPtrs, a->blockColIdxs, a->data, b->data, b->numCols, b->numCols, c, ldc,
                                  kernelNum, acc);
        });
    }
}

#define CUSPMM_INST(DT)                                                                                              \
    template DenseMatrix<DT, uint32_t> *spmmBSRCpu<DT, uint32_t, double>(SparseMatrixBSR<DT, uint32_t> *,           \
                                                                         DenseMatrix<DT, uint32_t> *,               \
                                                                         DenseMatrix<DT, uint32_t> *);              \
    template DenseMatrix<DT, uint32_t> *spmmBSRWrapper<DT, uint32_t, double>(int, SparseMatrixBSR<DT, uint32_t> *,  \
                                                                             DenseMatrix<DT, uint32_t> *,           \
                                                                             DenseMatrix<DT, uint32_t> *);
CUSPMM_INST(float)
CUSPMM_INST(double)
#undef CUSPMM_INST

}  // namespace cuspmm
