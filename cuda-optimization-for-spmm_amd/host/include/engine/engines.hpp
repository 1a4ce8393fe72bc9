// COO, ELL and BSR engines (the CSR engine, with its five numbered wrappers, is engine_csr.hpp).
// Same duck-typed surface runEngine relies on in the reference (include/engine/engine_{coo,ell,bsr}.hpp):
// MataT / MatbT, numKernels, SUPPORT_CUSPARSE, fmt, dirPath, seqTime, logSeq, report, runKernel --
// kernel 0 is the sequential CPU engine, kernels 1.. are HIP kernels behind the C ABI, -1 the default one.
#pragma once

#include "engine/engine_report.hpp"
#include "formats/sparse.hpp"

namespace cuspmm {

#define CUSPMM_DEFINE_ENGINE(F, NUM_KERNELS, HAS_VENDOR_CHECK)                                                     \
    template <typename DT, typename MT, typename AccT>                                                            \
    DenseMatrix<DT, MT> *spmm##F##Cpu(SparseMatrix##F<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc); \
    template <typename DT, typename MT, typename AccT>                                                            \
    DenseMatrix<DT, MT> *spmm##F##Wrapper(int kernelNum, SparseMatrix##F<DT, MT> *a, DenseMatrix<DT, MT> *b,     \
                                          DenseMatrix<DT, MT> *ref);                                              \
    template <typename DT, typename MT, typename AccT>                                                            \
    DenseMatrix<DT, MT> *spmm##F##Wrapper1(SparseMatrix##F<DT, MT> *a, DenseMatrix<DT, MT> *b,                   \
                                           DenseMatrix<DT, MT> *c) {                                              \
        return spmm##F##Wrapper<DT, MT, AccT>(1, a, b, c);                                                        \
    }                                                                                                             \
    template <typename DT, typename MT, typename AccT>                                                            \
    class Engine##F : public EngineCommon<SparseMatrix##F<DT, MT>, DenseMatrix<DT, MT>> {                         \
      public:                                                                                                     \
        using MataT = SparseMatrix##F<DT, MT>;                                                                    \
        using MatbT = DenseMatrix<DT, MT>;                                                                        \
        explicit Engine##F(std::string dirPath) {                                                                 \
            this->numKernels = (NUM_KERNELS);                                                                     \
            this->dirPath = dirPath;                                                                              \
            this->fmt = #F;                                                                                       \
            this->SUPPORT_CUSPARSE = (HAS_VENDOR_CHECK);                                                          \
        }                                                                                                         \
        void *runKernel(int num, void *_ma, void *_mb, void *_mc) override {                                      \
            auto ma = reinterpret_cast<MataT *>(_ma);                                                             \
            auto mb = reinterpret_cast<MatbT *>(_mb);                                                             \
            auto mc = reinterpret_cast<MatbT *>(_mc);                                                             \
            if (num == 0) return spmm##F##Cpu<DT, MT, AccT>(ma, mb, mc);                                          \
            if (num == -1) return spmm##F##Wrapper<DT, MT, AccT>(MISPMM_KERNEL_AUTO, ma, mb, mc);                 \
            if (num >= 1 && num <= this->numKernels) return spmm##F##Wrapper<DT, MT, AccT>(num, ma, mb, mc);      \
            throw std::runtime_error("Not implemented");                                                          \
        }                                                                                                         \
    };

// `--dtype bf16`: the BSR product on the bf16 MFMA kernels (host operands a, b and their device copies), two records
template <typename DT, typename MT>
void spmmBSRBf16(SparseMatrixBSR<DT, MT> *a, SparseMatrixBSR<DT, MT> *da, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *db);

// `--ell --gpus n`: A (host, the class's column-major arrays) sharded by rows over n devices in this process through
// mispmm_multi_ell_f32, B replicated, C slabs gathered according to gatherMode; one record with an extra "ngpus" key
template <typename DT, typename MT, typename AccT>
bool spmmELLMultiGpu(int ngpus, int gatherMode, SparseMatrixELL<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref);

// vendor cross-check: rocSPARSE has CSR and COO SpMM wired in; BSR and ELL follow the reference (none)
CUSPMM_DEFINE_ENGINE(COO, MISPMM_COO_NUM_KERNELS, true)
CUSPMM_DEFINE_ENGINE(ELL, MISPMM_ELL_NUM_KERNELS, false)
// BSR: kernels 1, 2 of mispmm_bsr_f32 plus kernel 3, the zero-skipping one (mispmm_bsr_nonzeros_f32)
CUSPMM_DEFINE_ENGINE(BSR, MISPMM_BSR_NUM_KERNELS + 1, false)
#undef CUSPMM_DEFINE_ENGINE

}  // namespace cuspmm
