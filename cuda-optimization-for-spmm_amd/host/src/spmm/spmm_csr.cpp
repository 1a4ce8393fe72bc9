// CSR: sequential CPU engine (kernel 0) and the HIP wrapper.
#include <algorithm>
#include <chrono>
#include <vector>

#include "engine/engine_csr.hpp"
#include "engine/wrapper_common.hpp"

namespace cuspmm {

// Kernel 0.  Same arithmetic as /root/reference/src/spmm/csr/spmm_csr.cpp:5-30: per output element
// the row's DT products are widened and summed in an AccT accumulator in storage order, then
// narrowed once.  It is the engine's sequential baseline and self-check reference -- the HIP
// wrappers never fall back to it.
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCSRCpu(SparseMatrixCSR<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc) {
    assert(!ma->onDevice && !mb->onDevice && !mc->onDevice);
    mb->toOrdering(ORDERING::ROW_MAJOR);
    const size_t n = mb->numCols;
    for (MT r = 0; r < ma->numRows; ++r) {
        const MT lo = ma->rowPtrs[r], hi = ma->rowPtrs[r + 1];
        DT *crow = mc->data + (size_t)r * n;
        for (size_t c = 0; c < n; ++c) {
            AccT acc = 0;
            for (MT i = lo; i < hi; ++i) {
                const DT prod = ma->data[i] * mb->data[(size_t)ma->colIdxs[i] * n + c];
                acc += prod;
            }
            crow[c] = (DT)acc;
        }
    }
    return mc;
}

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmCSRWrapper(int kernelNum, SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");  // like the reference, only float kernels exist
    } else {
        assert(a->onDevice && b->onDevice);
        b->toOrdering(ORDERING::ROW_MAJOR);  // untimed, on the device
        const double n = b->numCols;
        const WrapperShape shape{"CSR", a->numRows, a->numCols, a->numNonZero, 2.0 * a->numNonZero * n,
                                 a->numNonZero * 8.0 + (a->numRows + 1.0) * 4 + a->numCols * n * 4 + a->numRows * n * 4};
        const int acc = accModeOf<AccT>();
        // rows in the clustered order copy2Device found, where that is FASTER for this width -- measured once per width on scratch
        // operands, before the timed region (round 3 had a footprint rule here: n4c6-b13 x K=512 13.60 -> 12.99 us with the plan,
        // K=128 3.47 -> 3.69; the rule missed ACTIVSg10K x K=256: 12.6 -> 11.1).  Same bits: every row keeps its entries in storage order.
        const bool planPays = (kernelNum == MISPMM_KERNEL_AUTO || kernelNum == 5) && a->usePlanFor(b->numCols, acc);
        return runWrapper<DT, MT>(shape, kernelNum, b, ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
            if (a->planRowMap && planPays) {
                const float *bl[1] = {b->data};
                float *cl[1] = {c};
                const int st = mispmm_csr_plan_f32(stream, a->numRows, a->numCols, a->numNonZero, a->planRowPtrs, a->planColIdxs,
                                                   a->planData, a->uniformRowNnz, a->planRowMap, 1, bl, b->numCols, b->numCols, cl, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;
            }
            if (a->uniformRowNnz > 0 && (kernelNum == MISPMM_KERNEL_AUTO || kernelNum == 5)) {
                const int st = mispmm_csr_uniform_f32(stream, a->numRows, a->numCols, a->uniformRowNnz, a->colIdxs, a->data,
                                                      b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;  // a B of 2 GiB or more falls through to the general call
            }
            // long rows (spans built by copy2Device): kernel 6, and the library's own choice where it would split, with the
            // rows longest first
            const bool splits = kernelNum == 6 || ((kernelNum == MISPMM_KERNEL_AUTO || kernelNum == 5) &&
                                                   (acc == MISPMM_ACC_FAST || b->numCols < 384));
            if (a->rowSpans && splits && kernelNum != 6 && a->numLongSpans > 0 && a->numLongSpans < a->numSpans) {
                // the long rows by the split kernel's body, the short ones by the row-gather body of the same launch
                const int st = mispmm_csr_hybrid_f32(stream, a->numRows, a->numCols, a->numNonZero, a->colIdxs, a->data, a->rowSpans,
                                                     a->numSpans, a->numLongSpans, b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;  // shapes without such a launch take the split kernel on the whole list
            }
            if (a->rowSpans && splits && (kernelNum == 6 || !a->spansHybridOnly)) {
                const int st = mispmm_csr_split_f32(stream, a->numRows, a->numCols, a->numNonZero, a->rowPtrs, a->colIdxs, a->data,
                                                    a->rowSpans, a->numSpans, b->data, b->numCols, b->numCols, c, ldc, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;  // rows that are not 16-byte vectors take the general call
            }
            return mispmm_csr_f32(stream, a->numRows, a->numCols, a->numNonZero, a->rowPtrs, a->colIdxs, a->data, b->data,
                                  b->numCols, b->numCols, c, ldc, kernelNum, acc);
        });
    }
}

template <typename DT, typename MT, typename AccT>
bool spmmCSRBatched(int batch, SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        using clock = std::chrono::high_resolution_clock;
        auto ms = [](clock::time_point x, clock::time_point z) {
            return (double)std::chrono::duration_cast<std::chrono::microseconds>(z - x).count() / 1000.0;
        };
        assert(a->onDevice && b->onDevice && batch > 1);
        b->toOrdering(ORDERING::ROW_MAJOR);
        const uint32_t M = a->numRows, K = a->numCols, N = b->numCols;
        const int acc = accModeOf<AccT>();
        // untimed: the operands of the batch, each in a buffer of its own (copies of B: every result must equal `ref`)
        std::vector<float *> bs((size_t)batch), cs((size_t)batch);
        for (int i = 0; i < batch; ++i) {
            bs[i] = allocateBuffer<float>((size_t)K * N, true);
            copyBuffer(bs[i], true, b->data, true, (size_t)K * N * sizeof(float));
        }
        const bool planPays = a->usePlanFor(N, acc);   // measured once per width, outside the timed regions
        const auto t1 = clock::now();
        for (int i = 0; i < batch; ++i) cs[i] = allocateBuffer<float>((size_t)M * N, true);  // prolog: zero-filled results
        auto launch = [&](mispmm_stream_t stream) {
            if (a->planRowMap && planPays) {
                const int st = mispmm_csr_plan_f32(stream, M, K, a->numNonZero, a->planRowPtrs, a->planColIdxs, a->planData,
                                                   a->uniformRowNnz, a->planRowMap, (uint32_t)batch, bs.data(), N, N, cs.data(), N, acc);
                if (st != MISPMM_ERR_UNSUPPORTED) return st;
            }
            return mispmm_csr_batch_f32(stream, M, K, a->numNonZero, a->rowPtrs, a->colIdxs, a->data, a->uniformRowNnz, (uint32_t)batch,
                                        bs.data(), N, N, cs.data(), N, acc);
        };
        const auto t2 = clock::now();
        mispmmCheckError(launch(nullptr));
        SteadyStats steady;
        steady.kernelTag = mispmm_last_kernel();
        steady.batch = batch;
        mispmmCheckError(mispmm_device_sync());
        const auto t3 = clock::now();
        bool correct = ref != nullptr && ref->numRows == M && ref->numCols == N;
        std::vector<float> host((size_t)M * N);
        for (int i = 0; i < batch && correct; ++i) {
            copyBuffer(host.data(), false, cs[i], true, host.size() * sizeof(float));
            correct = allclose<float>(host.data(), ref->data, host.size(), REL_TOL, ABS_TOL);
        }
        const auto t4 = clock::now();
        const int iters = engineOptions().steadyIters;
        if (iters > 0) {
            // back-to-back batched launches captured into one hipGraph; figures are PER PRODUCT
            mispmm_stream_t st = nullptr;
            mispmm_event_t e0 = nullptr, e1 = nullptr;
            mispmmCheckError(mispmm_stream_create(&st));
            mispmmCheckError(mispmm_event_create(&e0));
            mispmmCheckError(mispmm_event_create(&e1));
            const int launches = std::max(1, std::min(1000, iters / batch));
            mispmm_graph_t graph = nullptr;
            mispmmCheckError(mispmm_graph_begin(st));
            for (int i = 0; i < launches; ++i) mispmmCheckError(launch(st));
            mispmmCheckError(mispmm_graph_end(st, &graph));
            for (auto w0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::now() - w0 < std::chrono::milliseconds(20);) {
                mispmmCheckError(mispmm_graph_launch(graph, st));
                mispmmCheckError(mispmm_stream_sync(st));
            }
            const int replays = std::max(1, iters / (launches * batch));
            mispmmCheckError(mispmm_event_record(e0, st));
            for (int r = 0; r < replays; ++r) mispmmCheckError(mispmm_graph_launch(graph, st));
            mispmmCheckError(mispmm_event_record(e1, st));
            mispmmCheckError(mispmm_event_sync(e1));
            float evMs = 0;
            mispmmCheckError(mispmm_event_elapsed_ms(e0, e1, &evMs));
            mispmmCheckError(mispmm_graph_destroy(graph));
            mispmmCheckError(mispmm_event_destroy(e0));
            mispmmCheckError(mispmm_event_destroy(e1));
            mispmmCheckError(mispmm_stream_destroy(st));
            const double products = (double)replays * launches * batch;
            const double sec = (double)evMs * 1e-3 / products;
            steady.iters = (int)products;
            steady.usPerSpmm = sec * 1e6;
            steady.gflops = 2.0 * a->numNonZero * N / sec / 1e9;
            steady.hbmGBps = (a->numNonZero * 8.0 + (M + 1.0) * 4 + (double)K * N * 4 + (double)M * N * 4) / sec / 1e9;
            steady.rooflineFrac = steady.hbmGBps / 8000.0;
        }
        reportTime(testcase, M, K, a->numNonZero, "CSR", b->ordering, 5, ms(t1, t2), ms(t2, t3), ms(t3, t4), correct, &steady);
        for (int i = 0; i < batch; ++i) {
            releaseBuffer(bs[i], true);
            releaseBuffer(cs[i], true);
        }
        return correct;
    }
}

#define CUSPMM_INST(DT)                                                                                              \
    template DenseMatrix<DT, uint32_t> *spmmCSRCpu<DT, uint32_t, double>(SparseMatrixCSR<DT, uint32_t> *,           \
                                                                         DenseMatrix<DT, uint32_t> *,               \
                                                                         DenseMatrix<DT, uint32_t> *);              \
    template DenseMatrix<DT, uint32_t> *spmmCSRWrapper<DT, uint32_t, double>(int, SparseMatrixCSR<DT, uint32_t> *,  \
                                                                             DenseMatrix<DT, uint32_t> *,           \
                                                                             DenseMatrix<DT, uint32_t> *);
CUSPMM_INST(float)
CUSPMM_INST(double)
#undef CUSPMM_INST
template bool spmmCSRBatched<float, uint32_t, double>(int, SparseMatrixCSR<float, uint32_t> *, DenseMatrix<float, uint32_t> *,
                                                      DenseMatrix<float, uint32_t> *);
template bool spmmCSRBatched<double, uint32_t, double>(int, SparseMatrixCSR<double, uint32_t> *, DenseMatrix<double, uint32_t> *,
                                                       DenseMatrix<double, uint32_t> *);
template DenseMatrix<float, uint32_t> *spmmCSRCpu<float, uint32_t, float>(SparseMatrixCSR<float, uint32_t> *,
                                                                         DenseMatrix<float, uint32_t> *,
                                                                         DenseMatrix<float, uint32_t> *);
template DenseMatrix<float, uint32_t> *spmmCSRWrapper<float, uint32_t, float>(int, SparseMatrixCSR<float, uint32_t> *,
                                                                             DenseMatrix<float, uint32_t> *,
                                                                             DenseMatrix<float, uint32_t> *);

}  // namespace cuspmm
