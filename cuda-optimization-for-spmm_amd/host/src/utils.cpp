#include "commons.hpp"
#include "engine/engine_base.hpp"
#include "formats/matrix.hpp"
#include "hip_utils.hpp"

std::string testcase;

void reportTime(const std::string &tc, uint32_t aNumRows, uint32_t aNumCols, uint32_t aNumNonZero,
                const std::string &format, int ordering, int kernelNum, double pro, double kernel, double epilog,
                bool correct, const SteadyStats *steady) {
    // the reference labels nnz / (rows * cols) "sparsity" and forms the product in 32 bits
    // (utils.hpp:39); the product is formed in double here so matrices beyond 65536^2 stay right
    const double density = (double)aNumNonZero / ((double)aNumRows * (double)aNumCols);
    std::cout << "{\n\"testcase\":\"" << tc << "\",\n"
              << "\"sparsity\":\"" << density << "\",\n"
              << "\"format\":\"" << format << "\",\n"
              << "\"kernelType\":\"" << kernelNum << "\",\n"
              << "\"denseOrdering\":\"" << (ordering == 0 ? "ROW_MAJOR" : "COL_MAJOR") << "\",\n"
              << "\"correct\":\"" << correct << "\",\n";
    std::printf("\"cudaPrologTimeMs\":\"%lf\",\n\"cudaKernelTimeMs\":\"%lf\",\n\"cudaEpilogTimeMs\":\"%lf\",\n"
                "\"cudaTotalTimeMs\":\"%lf\",\n\"sequentialTimeMs\":\"%lf\"",
                pro, kernel, epilog, pro + kernel + epilog, 0.0);
    if (steady && steady->iters > 0) {
        std::printf(",\n\"steadyIters\":\"%d\",\n\"steadyKernelUs\":\"%lf\",\n\"gflops\":\"%lf\",\n"
                    "\"hbmGBps\":\"%lf\",\n\"rooflineFrac\":\"%lf\"",
                    steady->iters, steady->usPerSpmm, steady->gflops, steady->hbmGBps, steady->rooflineFrac);
    }
    if (steady && steady->ngpus > 0) std::printf(",\n\"ngpus\":\"%d\"", steady->ngpus);
    if (steady && steady->dtype) std::printf(",\n\"dtype\":\"%s\"", steady->dtype);
    if (steady && steady->batch > 0) std::printf(",\n\"batch\":\"%d\"", steady->batch);
    if (steady && !steady->kernelTag.empty()) std::printf(",\n\"kernel\":\"%s\"", steady->kernelTag.c_str());
    std::printf("\n},\n");
    std::fflush(stdout);
}

namespace cuspmm {

EngineOptions &engineOptions() {
    static EngineOptions opts;
    return opts;
}

template <typename DT>
bool allclose(const DT *c, const DT *ref, size_t n, double rtol, double atol, double *maxAbsErr) {
    bool ok = true;
    double worst = 0;
    for (size_t i = 0; i < n; ++i) {
        const double x = c[i], y = ref[i];
        if (x != x || y != y) {  // NaN is never close (equal_nan = false)
            ok = false;
            continue;
        }
        if (x == y) continue;  // also covers equal infinities
        const double d = x > y ? x - y : y - x;
        if (d > worst) worst = d;
        if (!(d <= atol + rtol * (y < 0 ? -y : y))) ok = false;
    }
    if (maxAbsErr) *maxAbsErr = worst;
    return ok;
}

template bool allclose<float>(const float *, const float *, size_t, double, double, double *);
template bool allclose<double>(const double *, const double *, size_t, double, double, double *);

template <typename T> T *allocateBuffer(size_t count, bool onDevice) {
    void *p = nullptr;
    if (onDevice) mispmmCheckError(mispmm_malloc(&p, count * sizeof(T)));
    else mispmmCheckError(mispmm_host_alloc(&p, count * sizeof(T)));
    return static_cast<T *>(p);
}
template float *allocateBuffer<float>(size_t, bool);
template double *allocateBuffer<double>(size_t, bool);
template uint32_t *allocateBuffer<uint32_t>(size_t, bool);
template uint16_t *allocateBuffer<uint16_t>(size_t, bool);

void releaseBuffer(void *ptr, bool onDevice) {
    if (!ptr) return;
    if (onDevice) mispmmCheckError(mispmm_free(ptr));
    else mispmmCheckError(mispmm_host_free(ptr));
}

void copyBuffer(void *dst, bool dstOnDevice, const void *src, bool srcOnDevice, size_t bytes) {
    if (!dstOnDevice && !srcOnDevice) {  // host to host needs no device runtime (--cpu-only)
        if (bytes) std::memcpy(dst, src, bytes);
        return;
    }
    const int kind = srcOnDevice ? (dstOnDevice ? MISPMM_D2D : MISPMM_D2H) : (dstOnDevice ? MISPMM_H2D : MISPMM_H2H);
    mispmmCheckError(mispmm_memcpy(dst, src, bytes, kind));
}

}  // namespace cuspmm
