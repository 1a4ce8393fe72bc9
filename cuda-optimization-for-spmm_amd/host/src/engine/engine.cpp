#include "engine.hpp"

#include "format.hpp"

namespace cuspmm {

template <typename EngT>
void runEngine(EngT *engine, typename EngT::MataT *a, typename EngT::MatbT *b, float abs_tol, float rel_tol,
               bool skipSeq, bool cpuOnly, const std::string &savePath) {
    using ma_t = typename EngT::MataT;
    using mb_t = typename EngT::MatbT;
    using clock = std::chrono::high_resolution_clock;
    (void)abs_tol;
    (void)rel_tol;
    if (a->numCols != b->numRows) throw std::runtime_error("runEngine: A has " + std::to_string(a->numCols) +
                                                           " columns but B has " + std::to_string(b->numRows) + " rows");
    auto *c = new mb_t(a->numRows, b->numCols, false, ORDERING::ROW_MAJOR);  // zero-filled

    // 1. sequential CPU engine: the baseline time and the reference every HIP kernel is checked against
    const auto seqStart = clock::now();
    mb_t *cpuRes = c;
    if (!skipSeq) cpuRes = reinterpret_cast<mb_t *>(engine->runKernel(0, a, b, c));
    const double seqMs = (double)std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - seqStart).count() / 1000.0;
    engine->logSeq(seqMs);
    reportTime(testcase, a->numRows, a->numCols, a->numNonZero, engine->fmt, b->ordering, 0, 0, seqMs, 0, 1);
    if (!savePath.empty() && cpuOnly) cpuRes->save2File(savePath);

    if (!cpuOnly) {
        // 2. operands to the device
        ma_t *da = a->copy2Device();
        mb_t *db = b->copy2Device();

        // 3. every HIP kernel of this engine; each returns a fresh device C (or nullptr if it declines)
        mb_t *last = nullptr;
        for (int i = 1; i <= engine->numKernels; ++i) {
            auto *kRes = reinterpret_cast<mb_t *>(engine->runKernel(i, da, db, cpuRes));
            if (kRes != nullptr) {
                delete last;
                last = kRes;
            }
        }
        if (!savePath.empty() && last != nullptr) last->save2File(savePath);
        delete last;

        // 3a. `--dtype bf16`: the bf16 MFMA kernels for BSR (BASELINE config 4; the reference has no bf16)
        if constexpr (std::is_same_v<ma_t, SparseMatrixBSR<typename ma_t::DT, typename ma_t::MT>>) {
            if (engineOptions().bf16) spmmBSRBf16<typename ma_t::DT, typename ma_t::MT>(a, da, b, db);
        }

        // 3b. `--gpus n`: the same product row-sharded over n devices (CSR; new capability, src/main.cu:176 pins one)
        if constexpr (std::is_same_v<ma_t, SparseMatrixCSR<typename ma_t::DT, typename ma_t::MT>>) {
            if (engineOptions().gpus > 0)
                spmmCSRMultiGpu<typename ma_t::DT, typename ma_t::MT, double>(engineOptions().gpus, engineOptions().gatherMode, a,
                                                                              b, cpuRes);
        }

        if constexpr (std::is_same_v<ma_t, SparseMatrixELL<typename ma_t::DT, typename ma_t::MT>>) {   // ELL by rows
            if (engineOptions().gpus > 0)
                spmmELLMultiGpu<typename ma_t::DT, typename ma_t::MT, double>(engineOptions().gpus, engineOptions().gatherMode, a, b, cpuRes);
        }

        // 3c. `--batch n`: n dense operands multiplied by A in one launch (CSR; the reference multiplies by one dense.in per
        //     process, src/main.cu:185) -- what a caller with several right-hand sides gets for the launch boundary paid once
        if constexpr (std::is_same_v<ma_t, SparseMatrixCSR<typename ma_t::DT, typename ma_t::MT>>) {
            if (engineOptions().batch > 1)
                spmmCSRBatched<typename ma_t::DT, typename ma_t::MT, double>(engineOptions().batch, da, db, cpuRes);
        }

        // 4. vendor library, timed AND compared (the reference hard-codes correct = 1, engine.cpp:47-55)
        if (engine->SUPPORT_CUSPARSE && engineOptions().vendorCheck) {
            mb_t *dc = new mb_t(a->numRows, b->numCols, true, ORDERING::ROW_MAJOR);
            long pro = 0, kernel = 0, epi = 0;
            if (vendorTest<typename ma_t::DT, typename ma_t::MT>(da, db, dc, pro, kernel, epi)) {
                const auto t1 = clock::now();
                mb_t *host = dc->copy2Host();
                epi += std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t1).count();
                const bool same = allclose<typename ma_t::DT>(host->data, cpuRes->data, host->numElements(), REL_TOL, ABS_TOL);
                reportTime(testcase, a->numRows, a->numCols, a->numNonZero, engine->fmt, db->ordering, -1,
                           (double)pro / 1000, (double)kernel / 1000, (double)epi / 1000, same);
                delete host;
            }
            delete dc;
        }
        delete da;
        delete db;
    }
    delete c;
}

#define ENG_INST(fmt, dt, mt, acct)                                                                              \
    template void runEngine<Engine##fmt<dt, mt, acct>>(Engine##fmt<dt, mt, acct> *, Engine##fmt<dt, mt, acct>::MataT *, \
                                                       Engine##fmt<dt, mt, acct>::MatbT *, float, float, bool, bool, \
                                                       const std::string &);
ENG_INST(BSR, float, uint32_t, double)
ENG_INST(BSR, double, uint32_t, double)
ENG_INST(COO, float, uint32_t, double)
ENG_INST(COO, double, uint32_t, double)
ENG_INST(CSR, float, uint32_t, double)
ENG_INST(CSR, double, uint32_t, double)
ENG_INST(ELL, float, uint32_t, double)
ENG_INST(ELL, double, uint32_t, double)
#undef ENG_INST

}  // namespace cuspmm
