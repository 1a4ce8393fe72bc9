// BSR engine (/root/reference/include/engine/engine_bsr.hpp): kernel 0 is the sequential CPU engine,
// kernel 1.. the HIP kernels behind mispmm_bsr_f32.
#pragma once

#include "engine/engine_report.hpp"
#include "formats/sparse_bsr.hpp"

namespace cuspmm {

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmBSRCpu(SparseMatrixBSR<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc);

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmBSRWrapper(int kernelNum, SparseMatrixBSR<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref);

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmBSRWrapper1(SparseMatrixBSR<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *c) {
    return spmmBSRWrapper<DT, MT, AccT>(1, a, b, c);
}

template <typename DT, typename MT, typename AccT>
class EngineBSR : public EngineCommon<SparseMatrixBSR<DT, MT>, DenseMatrix<DT, MT>> {
  public:
    using MataT = SparseMatrixBSR<DT, MT>;
    using MatbT = DenseMatrix<DT, MT>;

    explicit EngineBSR(std::string dirPath) {
        this->numKernels = MISPMM_BSR_NUM_KERNELS;
        this->dirPath = dirPath;
        this->fmt = "BSR";
        this->SUPPORT_CUSPARSE = false;
    }

    void *runKernel(int num, void *_ma, void *_mb, void *_mc) override {
        auto ma = reinterpret_cast<MataT *>(_ma);
        auto mb = reinterpret_cast<MatbT *>(_mb);
        auto mc = reinterpret_cast<MatbT *>(_mc);
        if (num == 0) return spmmBSRCpu<DT, MT, AccT>(ma, mb, mc);
        if (num == -1) return spmmBSRWrapper<DT, MT, AccT>(MISPMM_KERNEL_AUTO, ma, mb, mc);
        if (num >= 1 && num <= this->numKernels) return spmmBSRWrapper<DT, MT, AccT>(num, ma, mb, mc);
        throw std::runtime_error("Not implemented");
    }
};

}  // namespace cuspmm
