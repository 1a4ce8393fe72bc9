// The part every spmm<F>Wrapper shares: the timing contract of the reference's wrappers
// (e.g. /root/reference/src/spmm/csr/spmm_csr_k3.cu:58-105) --
//   untimed : B brought to row-major;   prolog : allocate + zero C on the device;
//   kernel  : one launch + device sync; epilog : C copied back to the host;
// then the self-check against the CPU result and one record.  Optionally followed by a
// steady-state measurement (HIP events over back-to-back launches) for GFLOP/s and roofline.
#pragma once

#include <functional>

#include "commons.hpp"
#include "engine/engine_base.hpp"
#include "formats/dense.hpp"

namespace cuspmm {

template <typename AccT> inline int accModeOf() {
    const int o = engineOptions().accOverride;
    if (o >= 0) return o;
    return std::is_same_v<AccT, double> ? MISPMM_ACC_REFERENCE : MISPMM_ACC_FAST;
}

struct WrapperShape {
    const char *format;
    uint32_t rows, cols, nnz;  // of A, as the record prints them
    double flops;              // 2 * (useful non-zeros) * N
    double algorithmicBytes;   // compulsory traffic of one SpMM
    const char *dtype = nullptr;  // printed with the record when the kernel's arithmetic type is not DT
};

// launch(cData, ldc, stream) enqueues one SpMM into the device buffer on `stream` (NULL = the default stream, as the
// reference launches) and returns a mispmm status.
template <typename DT, typename MT>
DenseMatrix<DT, MT> *runWrapper(const WrapperShape &shape, int kernelNum, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref,
                                const std::function<int(DT *, uint32_t, mispmm_stream_t)> &launch);

}  // namespace cuspmm
