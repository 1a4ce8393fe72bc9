// ELL engine (/root/reference/include/engine/engine_ell.hpp): kernel 0 is the sequential CPU engine,
// kernel 1.. the HIP kernels behind mispmm_ell_f32.
#pragma once

#include "engine/engine_report.hpp"
#include "formats/sparse_ell.hpp"

namespace cuspmm {

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmELLCpu(SparseMatrixELL<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc);

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmELLWrapper(int kernelNum, SparseMatrixELL<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref);

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmELLWrapper1(SparseMatrixELL<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *c) {
    return spmmELLWrapper<DT, MT, AccT>(1, a, b, c);
}

template <typename DT, typename MT, typename AccT>
class EngineELL : public EngineCommon<SparseMatrixELL<DT, MT>, DenseMatrix<DT, MT>> {
  public:
    using MataT = SparseMatrixELL<DT, MT>;
    using MatbT = DenseMatrix<DT, MT>;

    explicit EngineELL(std::string dirPath) {
        this->numKernels = MISPMM_ELL_NUM_KERNELS;
        this->dirPath = dirPath;
        this->fmt = "ELL";
        this->SUPPORT_CUSPARSE = false;
    }

    void *runKernel(int num, void *_ma, void *_mb, void *_mc) override {
        auto ma = reinterpret_cast<MataT *>(_ma);
        auto mb = reinterpret_cast<MatbT *>(_mb);
        auto mc = reinterpret_cast<MatbT *>(_mc);
        if (num == 0) return spmmELLCpu<DT, MT, AccT>(ma, mb, mc);
        if (num == -1) return spmmELLWrapper<DT, MT, AccT>(MISPMM_KERNEL_AUTO, ma, mb, mc);
        if (num >= 1 && num <= this->numKernels) return spmmELLWrapper<DT, MT, AccT>(num, ma, mb, mc);
        throw std::runtime_error("Not implemented");
    }
};

}  // namespace cuspmm
