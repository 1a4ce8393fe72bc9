// ELL: sequential CPU engine (kernel 0) and the HIP wrapper.
#include "engine/engine_ell.hpp"
#include "engine/wrapper_common.hpp"

namespace cuspmm {

// Kernel 0 walks the column-major slots (column, then slot) and scatters value * B[col, :] into
// C[row, :] in DT, skipping padding (/root/reference/src/spmm/ell/spmm_ell.cpp:16-29).
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmELLCpu(SparseMatrixELL<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc) {
    assert(!ma->onDevice && !mb->onDevice && !mc->onDevice);
    mb->toOrdering(ORDERING::ROW_MAJOR);
    const size_t n = mb->numCols;
    for (MT col = 0; col < ma->numCols; ++col) {
        const DT *brow = mb->data + (size_t)col * n;
        for (MT s = 0; s < ma->maxColNnz; ++s) {
            const size_t i = (size_t)col * ma->maxColNnz + s;
            const int32_t row = (int32_t)ma->rowIdxs[i];
            if (row < 0) continue;
            const DT v = ma->data[i];
            DT *crow = mc->data + (size_t)row * n;
            for (size_t j = 0; j < n; ++j) {
                const DT prod = v * brow[j];
                crow[j] += prod;
            }
        }
    }
    return mc;
}

// The HIP path is row-parallel and atomics-free; it consumes the row-major view built by
// SparseMatrixELL::copy2Device().  (The reference's wrapper scatters with atomicAdd straight into
// the host-pinned CPU result and neither checks nor reports: spmm_ell_k1.cu:38-63.  This one
// produces its own device C, checks it and reports like every other format.)
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmELLWrapper(int kernelNum, SparseMatrixELL<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        assert(a->onDevice && b->onDevice);
        b->toOrdering(ORDERING::ROW_MAJOR);
        const double n = b->numCols;
        const WrapperShape shape{"ELL", a->numRows, a->numCols, a->numNonZero, 2.0 * a->numNonZero * n,
                                 (double)a->numRows * a->rowWidth * 8.0 + a->numCols * n * 4 + a->numRows * n * 4};
        const int acc = accModeOf<AccT>();
        return runWrapper<DT, MT>(shape, kernelNum, b, ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
            if (a->cpSpans && a->cpSpansLong > 0 && a->cpSpansLong < a->numRows) {  // the long rows on the split shape, the short ones by lane groups, one launch
                const int st = mispmm_rows_hybrid_f32(stream, a->numRows, a->numCols, a->cpCount, a->cpColIdxs, a->cpData, a->cpSpans, a->numRows,
                                                      a->cpSpansLong, b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;
            }
            if (a->cpSpans && !a->cpSpansHybridOnly) {  // ... and long rows: longest first
                const int st = mispmm_rows_split_f32(stream, a->numRows, a->numCols, a->cpCount, a->cpColIdxs, a->cpData, a->cpSpans,
                                                     a->numRows, b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;
            }
            if (a->cpRowPtrs) {  // mostly padding: multiply from the list of occupied slots (same order, same bits)
                const int st = mispmm_ell_compact_f32(stream, a->numRows, a->numCols, a->cpCount, a->cpRowPtrs, a->cpColIdxs, a->cpData,
                                                      b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;
            }
            return mispmm_ell_f32(stream, a->numRows, a->numCols, a->rowWidth, a->rmColIdxs, a->rmData, b->data,
                                  b->numCols, b->numCols, c, ldc, kernelNum, acc);
        });
    }
}

#define CUSPMM_INST(DT)                                                                                              \
    template DenseMatrix<DT, uint32_t> *spmmELLCpu<DT, uint32_t, double>(SparseMatrixELL<DT, uint32_t> *,           \
                                                                         DenseMatrix<DT, uint32_t> *,               \
                                                                         DenseMatrix<DT, uint32_t> *);              \
    template DenseMatrix<DT, uint32_t> *spmmELLWrapper<DT, uint32_t, double>(int, SparseMatrixELL<DT, uint32_t> *,  \
                                                                             DenseMatrix<DT, uint32_t> *,           \
                                                                             DenseMatrix<DT, uint32_t> *);
CUSPMM_INST(float)
CUSPMM_INST(double)
#undef CUSPMM_INST

}  // namespace cuspmm
