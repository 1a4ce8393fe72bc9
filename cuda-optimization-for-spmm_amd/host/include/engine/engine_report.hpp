// The per-engine record printer every Engine<F> carries (the reference duplicates this member in
// each engine header, e.g. include/engine/engine_csr.hpp:43-66; here it is one mix-in).
#pragma once

#include <string>

#include "commons.hpp"
#include "engine/engine_base.hpp"
#include "formats/dense.hpp"

namespace cuspmm {

template <typename MataT_, typename MatbT_> class EngineCommon : public EngineBase {
  public:
    using MataT = MataT_;
    using MatbT = MatbT_;
    bool SUPPORT_CUSPARSE = false;  // name kept from the reference: "has a vendor-library cross-check"
    std::string fmt;
    std::string dirPath;
    double seqTime = 1.f;

    void logSeq(double seq) { this->seqTime = seq; }

    void report(MataT *a, MatbT *b, int num, double pro, double kernel, double epilog, bool correct) {
        const char *ord = b->ordering == ORDERING::ROW_MAJOR ? "ROW_MAJOR" : "COL_MAJOR";
        const double density = (double)a->numNonZero / ((double)a->numRows * (double)a->numCols);
        std::cout << "{\n\"testcase\":\"" << this->dirPath << "\",\n"
                  << "\"sparsity\":\"" << density << "\",\n"
                  << "\"format\":\"" << this->fmt << "\",\n"
                  << "\"kernelType\":\"" << num << "\",\n"
                  << "\"denseOrdering\":\"" << ord << "\",\n"
                  << "\"correct\":\"" << correct << "\",\n";
        std::printf("\"cudaPrologTimeMs\":\"%lf\",\n\"cudaKernelTimeMs\":\"%lf\",\n\"cudaEpilogTimeMs\":\"%lf\",\n"
                    "\"cudaTotalTimeMs\":\"%lf\",\n\"sequentialTimeMs\":\"%lf\"\n},\n",
                    pro, kernel, epilog, pro + kernel + epilog, this->seqTime);
    }
};

}  // namespace cuspmm
