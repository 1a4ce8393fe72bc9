// CSR: sequential CPU engine (kernel 0) and the HIP wrapper.
#include "engine/engine_csr.hpp"
#include "engine/wrapper_common.hpp"

namespace cuspmm {

// Kernel 0.  Same arithmetic as /root/reference/src/spmm/csr/spmm_csr.cpp:5-30: per output element
// the row's DT products are widened and summed in an AccT accumulator in storage order, then
// narrowed once.  It is the engine's sequential baseline and self-check reference -- the HIP
// wrappers never fall back to it.
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCSRCpu(SparseMatrixCSR<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc) {
    assert(!ma->onDevice && !mb->onDevice && !mc->onDevice);
    mb->toOrdering(ORDERING::ROW_MAJOR);
    const size_t n = mb->numCols;
    for (MT r = 0; r < ma->numRows; ++r) {
        const MT lo = ma->rowPtrs[r], hi = ma->rowPtrs[r + 1];
        DT *crow = mc->data + (size_t)r * n;
        for (size_t c = 0; c < n; ++c) {
            AccT acc = 0;
            for (MT i = lo; i < hi; ++i) {
                const DT prod = ma->data[i] * mb->data[(size_t)ma->colIdxs[i] * n + c];
                acc += prod;
            }
            crow[c] = (DT)acc;
        }
    }
    return mc;
}

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCSRWrapper(int kernelNum, SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");  // like the reference, only float kernels exist
    } else {
        assert(a->onDevice && b->onDevice);
        b->toOrdering(ORDERING::ROW_MAJOR);  // untimed, on the device
        const double n = b->numCols;
        const WrapperShape shape{"CSR", a->numRows, a->numCols, a->numNonZero, 2.0 * a->numNonZero * n,
                                 a->numNonZero * 8.0 + (a->numRows + 1.0) * 4 + a->numCols * n * 4 + a->numRows * n * 4};
        const int acc = accModeOf<AccT>();
        return runWrapper<DT, MT>(shape, kernelNum, b, ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
            if (a->uniformRowNnz > 0 && (kernelNum == MISPMM_KERNEL_AUTO || kernelNum == 5)) {
                const int st = mispmm_csr_uniform_f32(stream, a->numRows, a->numCols, a->uniformRowNnz, a->colIdxs, a->data,
                                                      b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;  // a B of 2 GiB or more falls through to the general call
            }
            // long rows (spans built by copy2Device): kernel 6, and the library's own choice where it would split, with the
            // rows longest first
            const bool splits = kernelNum == 6 || ((kernelNum == MISPMM_KERNEL_AUTO || kernelNum == 5) &&
                                                   (acc == MISPMM_ACC_FAST || b->numCols < 384));
            if (a->rowSpans && splits) {
                const int st = mispmm_csr_split_f32(stream, a->numRows, a->numCols, a->numNonZero, a->rowPtrs, a->colIdxs, a->data,
                                                    a->rowSpans, a->numSpans, b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;  // rows that are not 16-byte vectors take the general call
            }
            return mispmm_csr_f32(stream, a->numRows, a->numCols, a->numNonZero, a->rowPtrs, a->colIdxs, a->data, b->data,
                                  b->numCols, b->numCols, c, ldc, kernelNum, acc);
        });
    }
}

#define CUSPMM_INST(DT)                                                                                              \
    template DenseMatrix<DT, uint32_t> *spmmCSRCpu<DT, uint32_t, double>(SparseMatrixCSR<DT, uint32_t> *,           \
                                                                         DenseMatrix<DT, uint32_t> *,               \
                                                                         DenseMatrix<DT, uint32_t> *);              \
    template DenseMatrix<DT, uint32_t> *spmmCSRWrapper<DT, uint32_t, double>(int, SparseMatrixCSR<DT, uint32_t> *,  \
                                                                             DenseMatrix<DT, uint32_t> *,           \
                                                                             DenseMatrix<DT, uint32_t> *);
CUSPMM_INST(float)
CUSPMM_INST(double)
#undef CUSPMM_INST
template DenseMatrix<float, uint32_t> *spmmCSRCpu<float, uint32_t, float>(SparseMatrixCSR<float, uint32_t> *,
                                                                         DenseMatrix<float, uint32_t> *,
                                                                         DenseMatrix<float, uint32_t> *);
template DenseMatrix<float, uint32_t> *spmmCSRWrapper<float, uint32_t, float>(int, SparseMatrixCSR<float, uint32_t> *,
                                                                             DenseMatrix<float, uint32_t> *,
                                                                             DenseMatrix<float, uint32_t> *);

}  // namespace cuspmm
