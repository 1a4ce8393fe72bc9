#include <chrono>
#include "engine/wrapper_common.hpp"

namespace cuspmm {

template <typename DT, typename MT>
DenseMatrix<DT, MT> *runWrapper(const WrapperShape &shape, int kernelNum, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref,
                                const std::function<int(DT *, uint32_t, mispmm_stream_t)> &launch) {
    using clock = std::chrono::high_resolution_clock;
    auto us = [](clock::time_point a, clock::time_point z) {
        return (double)std::chrono::duration_cast<std::chrono::microseconds>(z - a).count() / 1000.0;
    };
    assert(b->onDevice && b->ordering == ORDERING::ROW_MAJOR);
    const MT cols = b->numCols;

    const auto t1 = clock::now();
    auto *c = new DenseMatrix<DT, MT>(shape.rows, cols, true, ORDERING::ROW_MAJOR);
    const auto t2 = clock::now();
    const int status = launch(c->data, cols, nullptr);
    if (status == MISPMM_ERR_UNSUPPORTED) {
        // the kernel declines this shape: report zeros and hand back nothing, as the reference's
        // K4 does (/root/reference/src/spmm/csr/spmm_csr_k4.cu:97-101)
        delete c;
        reportTime(testcase, shape.rows, shape.cols, shape.nnz, shape.format, b->ordering, kernelNum, 0, 0, 0, false);
        return nullptr;
    }
    mispmmCheckError(status);
    const std::string kernelTag = mispmm_last_kernel();  // what this kernel id really launched
    mispmmCheckError(mispmm_device_sync());
    const auto t3 = clock::now();
    DenseMatrix<DT, MT> *res = c->copy2Host();
    const auto t4 = clock::now();

    bool correct = false;
    if (ref != nullptr && ref->numRows == res->numRows && ref->numCols == res->numCols)
        correct = allclose<DT>(res->data, ref->data, res->numElements(), REL_TOL, ABS_TOL);
    delete res;

    SteadyStats steady;
    steady.dtype = shape.dtype;
    steady.kernelTag = kernelTag;
    const int iters = engineOptions().steadyIters;
    if (iters > 0) {
        // steady state: `iters` back-to-back launches captured into hipGraphs (chunks of <= 1000) on a stream of the
        // wrapper and replayed -- eager launches of a 3.5 us kernel measure the host's launch rate, not the kernel
        mispmm_stream_t st = nullptr;
        mispmmCheckError(mispmm_stream_create(&st));
        mispmm_event_t e0 = nullptr, e1 = nullptr;
        mispmmCheckError(mispmm_event_create(&e0));
        mispmmCheckError(mispmm_event_create(&e1));
        const int chunk = iters < 1000 ? iters : 1000;
        mispmm_graph_t graph = nullptr;
        mispmmCheckError(mispmm_graph_begin(st));
        for (int i = 0; i < chunk; ++i) mispmmCheckError(launch(c->data, cols, st));
        mispmmCheckError(mispmm_graph_end(st, &graph));
        const int replays = (iters + chunk - 1) / chunk;
        // warm-up replays for ~20 ms (graph upload, caches, clocks at their busy level)
        for (auto w0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - w0 < std::chrono::milliseconds(20);) {
            mispmmCheckError(mispmm_graph_launch(graph, st));
            mispmmCheckError(mispmm_stream_sync(st));
        }
        mispmmCheckError(mispmm_event_record(e0, st));
        for (int r = 0; r < replays; ++r) mispmmCheckError(mispmm_graph_launch(graph, st));
        mispmmCheckError(mispmm_event_record(e1, st));
        mispmmCheckError(mispmm_event_sync(e1));
        float ms = 0;
        mispmmCheckError(mispmm_event_elapsed_ms(e0, e1, &ms));
        mispmmCheckError(mispmm_graph_destroy(graph));
        mispmmCheckError(mispmm_event_destroy(e0));
        mispmmCheckError(mispmm_event_destroy(e1));
        mispmmCheckError(mispmm_stream_destroy(st));
        const int timed = replays * chunk;
        const double sec = (double)ms * 1e-3 / timed;
        steady.iters = timed;
        steady.usPerSpmm = sec * 1e6;
        steady.gflops = shape.flops / sec / 1e9;
        steady.hbmGBps = shape.algorithmicBytes / sec / 1e9;
        steady.rooflineFrac = steady.hbmGBps / 8000.0;
    }
    reportTime(testcase, shape.rows, shape.cols, shape.nnz, shape.format, b->ordering, kernelNum, us(t1, t2), us(t2, t3),
               us(t3, t4), correct, &steady);
    return c;
}

template DenseMatrix<float, uint32_t> *runWrapper<float, uint32_t>(const WrapperShape &, int, DenseMatrix<float, uint32_t> *,
                                                                 DenseMatrix<float, uint32_t> *,
                                                                 const std::function<int(float *, uint32_t, mispmm_stream_t)> &);

}  // namespace cuspmm
