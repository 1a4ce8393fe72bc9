// CSR engine (/root/reference/include/engine/engine_csr.hpp:27-91): kernel 0 is the sequential CPU
// engine, kernels 1..numKernels are HIP kernels reached through mispmm_csr_f32.
#pragma once

#include "engine/engine_report.hpp"
#include "formats/sparse.hpp"

namespace cuspmm {

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCSRCpu(SparseMatrixCSR<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc);

// Shared body of the numbered wrappers: B to row-major (untimed), then prolog (allocate C) /
// kernel (launch + sync) / epilog (copy back), self-check against `ref`, one record.
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCSRWrapper(int kernelNum, SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref);

// `--gpus n`: A (host) row-sharded over n devices in this process, B replicated, C slabs gathered according to
// gatherMode (mispmm_gather_mode); one record with an extra "ngpus" key.  Returns the self-check verdict.
template <typename DT, typename MT, typename AccT>
bool spmmCSRMultiGpu(int ngpus, int gatherMode, SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref);

// `--batch n`: n dense operands (n device copies of B in buffers of their own) multiplied by the device matrix in ONE
// launch (mispmm_csr_batch_f32, or mispmm_csr_plan_f32 where the clustered row order pays); every result is checked
// against `ref`; one record with an extra "batch" key whose steady-state figures are per product.
template <typename DT, typename MT, typename AccT>
bool spmmCSRBatched(int batch, SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref);

#define CUSPMM_DECLARE_CSR_WRAPPER(N)                                                                          \
    template <typename DT, typename MT, typename AccT>                                                         \
    DenseMatrix<DT, MT> *spmmCSRWrapper##N(SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b,                 \
                                           DenseMatrix<DT, MT> *c) {                                           \
        return spmmCSRWrapper<DT, MT, AccT>(N, a, b, c);                                                       \
    }
CUSPMM_DECLARE_CSR_WRAPPER(1)
CUSPMM_DECLARE_CSR_WRAPPER(2)
CUSPMM_DECLARE_CSR_WRAPPER(3)
CUSPMM_DECLARE_CSR_WRAPPER(4)
CUSPMM_DECLARE_CSR_WRAPPER(5)
#undef CUSPMM_DECLARE_CSR_WRAPPER

template <typename DT, typename MT, typename AccT>
class EngineCSR : public EngineCommon<SparseMatrixCSR<DT, MT>, DenseMatrix<DT, MT>> {
  public:
    using MataT = SparseMatrixCSR<DT, MT>;
    using MatbT = DenseMatrix<DT, MT>;

    explicit EngineCSR(std::string dirPath) {
        this->numKernels = MISPMM_CSR_NUM_KERNELS;
        this->dirPath = dirPath;
        this->fmt = "CSR";
        this->SUPPORT_CUSPARSE = true;
    }

    void *runKernel(int num, void *_ma, void *_mb, void *_mc) override {
        auto ma = reinterpret_cast<MataT *>(_ma);
        auto mb = reinterpret_cast<MatbT *>(_mb);
        auto mc = reinterpret_cast<MatbT *>(_mc);
        if (num == 0) return spmmCSRCpu<DT, MT, AccT>(ma, mb, mc);
        if (num == -1) return spmmCSRWrapper<DT, MT, AccT>(MISPMM_KERNEL_AUTO, ma, mb, mc);
        if (num >= 1 && num <= this->numKernels) return spmmCSRWrapper<DT, MT, AccT>(num, ma, mb, mc);
        throw std::runtime_error("Not implemented");
    }
};

}  // namespace cuspmm
