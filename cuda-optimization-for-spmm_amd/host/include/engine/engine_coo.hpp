// COO engine (/root/reference/include/engine/engine_coo.hpp): kernel 0 is the sequential CPU engine,
// kernel 1.. the HIP kernels behind mispmm_coo_f32.
#pragma once

#include "engine/engine_report.hpp"
#include "formats/sparse_coo.hpp"

namespace cuspmm {

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCOOCpu(SparseMatrixCOO<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc);

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCOOWrapper(int kernelNum, SparseMatrixCOO<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref);

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCOOWrapper1(SparseMatrixCOO<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *c) {
    return spmmCOOWrapper<DT, MT, AccT>(1, a, b, c);
}

template <typename DT, typename MT, typename AccT>
class EngineCOO : public EngineCommon<SparseMatrixCOO<DT, MT>, DenseMatrix<DT, MT>> {
  public:
    using MataT = SparseMatrixCOO<DT, MT>;
    using MatbT = DenseMatrix<DT, MT>;

    explicit EngineCOO(std::string dirPath) {
        this->numKernels = MISPMM_COO_NUM_KERNELS;
        this->dirPath = dirPath;
        this->fmt = "COO";
        this->SUPPORT_CUSPARSE = true;
    }

    void *runKernel(int num, void *_ma, void *_mb, void *_mc) override {
        auto ma = reinterpret_cast<MataT *>(_ma);
        auto mb = reinterpret_cast<MatbT *>(_mb);
        auto mc = reinterpret_cast<MatbT *>(_mc);
        if (num == 0) return spmmCOOCpu<DT, MT, AccT>(ma, mb, mc);
        if (num == -1) return spmmCOOWrapper<DT, MT, AccT>(MISPMM_KERNEL_AUTO, ma, mb, mc);
        if (num >= 1 && num <= this->numKernels) return spmmCOOWrapper<DT, MT, AccT>(num, ma, mb, mc);
        throw std::runtime_error("Not implemented");
    }
};

}  // namespace cuspmm
