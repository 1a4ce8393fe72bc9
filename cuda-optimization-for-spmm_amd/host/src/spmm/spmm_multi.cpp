// `cuspmm --csr --gpus n`: the CSR SpMM row-sharded over n devices of this node in ONE process, through
// mispmm_multi_csr_f32 (include/mispmm.h, multi-GPU section).  New capability: the reference drives a single
// device (cudaSetDevice(7), /root/reference/src/main.cu:176).  Timing sections as in the single-GPU wrappers
// (e.g. /root/reference/src/spmm/csr/spmm_csr_k3.cu:58-105): prolog = allocate C on every device, kernel =
// per-device kernels + slab gather + sync of every stream, epilog = the gathered C copied back from device 0;
// sharding A and replicating B are the multi-GPU equivalent of copy2Device and, like it, untimed.
#include <algorithm>
#include <chrono>
#include <vector>

#include "engine/engine_csr.hpp"
#include "engine/engine_ell.hpp"
#include "engine/wrapper_common.hpp"

namespace cuspmm {

namespace {
struct DeviceSlot {
    int ordinal = 0;
    mispmm_stream_t stream = nullptr;
    uint32_t *rowPtrs = nullptr, *colIdxs = nullptr;
    float *vals = nullptr, *b = nullptr, *c = nullptr;
    uint32_t nnz = 0, uniform = 0;
};

uint32_t uniformRowNnz(const uint32_t *ptr, uint32_t rows) {
    if (rows == 0) return 0;
    const uint32_t w = ptr[1] - ptr[0];
    if (w == 0) return 0;
    for (uint32_t r = 0; r < rows; ++r)
        if (ptr[r + 1] - ptr[r] != w) return 0;
    return w;
}
}  // namespace

template <typename DT, typename MT, typename AccT>
bool spmmCSRMultiGpu(int ngpus, int gatherMode, SparseMatrixCSR<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        using clock = std::chrono::high_resolution_clock;
        auto ms = [](clock::time_point x, clock::time_point z) {
            return (double)std::chrono::duration_cast<std::chrono::microseconds>(z - x).count() / 1000.0;
        };
        assert(!a->onDevice && !b->onDevice && ngpus >= 1);
        b->toOrdering(ORDERING::ROW_MAJOR);
        int count = 0, home = 0;
        mispmmCheckError(mispmm_device_count(&count));
        mispmmCheckError(mispmm_get_device(&home));
        if (ngpus > count) throw std::runtime_error("--gpus " + std::to_string(ngpus) + ": this node has " + std::to_string(count) + " device(s)");
        const uint32_t M = a->numRows, K = a->numCols, N = b->numCols;
        std::vector<uint32_t> bounds((size_t)ngpus + 1);
        mispmmCheckError(mispmm_shard_rows_by_nnz_host(M, a->rowPtrs, (uint32_t)ngpus, bounds.data()));
        // RCCL gather: ONE ncclAllGather needs slabs of equal height.  Equal row chunks are taken instead of the
        // nnz-balanced ranges when their heaviest slab carries at most 2 % more entries (always so for the uniform-row
        // matrices of the BASELINE configs); C is then padded to ngpus * chunk rows.  Otherwise the uneven slabs travel
        // as grouped broadcasts.
        size_t cRows = M;
        if (gatherMode == MISPMM_GATHER_ALL_RCCL && ngpus > 1) {
            const uint32_t chunk = (M + (uint32_t)ngpus - 1) / (uint32_t)ngpus;
            uint64_t heavyBalanced = 0, heavyEqual = 0;
            for (int d = 0; d < ngpus; ++d) {
                const uint32_t e0 = std::min<uint64_t>(M, (uint64_t)d * chunk), e1 = std::min<uint64_t>(M, (uint64_t)(d + 1) * chunk);
                heavyBalanced = std::max<uint64_t>(heavyBalanced, a->rowPtrs[bounds[d + 1]] - a->rowPtrs[bounds[d]]);
                heavyEqual = std::max<uint64_t>(heavyEqual, a->rowPtrs[e1] - a->rowPtrs[e0]);
            }
            if (heavyEqual * 100 <= heavyBalanced * 102) {
                for (int d = 0; d <= ngpus; ++d) bounds[d] = (uint32_t)std::min<uint64_t>(M, (uint64_t)d * chunk);
                gatherMode = MISPMM_GATHER_ALL_RCCL_EQUAL;
                cRows = (size_t)chunk * ngpus;
            }
        }

        // untimed: device d gets its row slice (row pointers rebased), a replica of B
        std::vector<DeviceSlot> slots((size_t)ngpus);
        std::vector<int> ordinals((size_t)ngpus);
        for (int d = 0; d < ngpus; ++d) {
            DeviceSlot &s = slots[d];
            s.ordinal = ordinals[d] = (home + d) % count;
            mispmmCheckError(mispmm_set_device(s.ordinal));
            mispmmCheckError(mispmm_stream_create(&s.stream));
            const uint32_t r0 = bounds[d], rows = bounds[d + 1] - r0, e0 = a->rowPtrs[r0];
            s.nnz = a->rowPtrs[bounds[d + 1]] - e0;
            std::vector<uint32_t> rebased((size_t)rows + 1);
            for (uint32_t r = 0; r <= rows; ++r) rebased[r] = a->rowPtrs[r0 + r] - e0;
            s.uniform = uniformRowNnz(rebased.data(), rows);
            s.rowPtrs = allocateBuffer<uint32_t>((size_t)rows + 1, true);
            s.colIdxs = allocateBuffer<uint32_t>(s.nnz ? s.nnz : 1, true);
            s.vals = allocateBuffer<float>(s.nnz ? s.nnz : 1, true);
            s.b = allocateBuffer<float>((size_t)K * N, true);
            copyBuffer(s.rowPtrs, true, rebased.data(), false, ((size_t)rows + 1) * sizeof(uint32_t));
            copyBuffer(s.colIdxs, true, a->colIdxs + e0, false, (size_t)s.nnz * sizeof(uint32_t));
            copyBuffer(s.vals, true, a->data + e0, false, (size_t)s.nnz * sizeof(float));
            copyBuffer(s.b, true, b->data, false, (size_t)K * N * sizeof(float));
        }
        if (ngpus > 1) mispmmCheckError(mispmm_enable_peer_access((uint32_t)ngpus, ordinals.data()));
        mispmm_comm_t comm = nullptr;
        if (gatherMode == MISPMM_GATHER_ALL_RCCL || gatherMode == MISPMM_GATHER_ALL_RCCL_EQUAL) mispmmCheckError(mispmm_comm_create(&comm, (uint32_t)ngpus, ordinals.data()));

        const auto t1 = clock::now();
        for (DeviceSlot &s : slots) {  // prolog: C (zero-filled) on every device
            mispmmCheckError(mispmm_set_device(s.ordinal));
            s.c = allocateBuffer<float>(cRows * N, true);
        }
        std::vector<mispmm_stream_t> streams;
        std::vector<const uint32_t *> rowPtrs, colIdxs;
        std::vector<const float *> vals, bs;
        std::vector<float *> cs;
        std::vector<uint32_t> nnz, uniform;
        for (DeviceSlot &s : slots) {
            streams.push_back(s.stream);
            rowPtrs.push_back(s.rowPtrs);
            colIdxs.push_back(s.colIdxs);
            vals.push_back(s.vals);
            bs.push_back(s.b);
            cs.push_back(s.c);
            nnz.push_back(s.nnz);
            uniform.push_back(s.uniform);
        }
        const int acc = accModeOf<AccT>();
        auto launch = [&] {
            return mispmm_multi_csr_f32((uint32_t)ngpus, ordinals.data(), streams.data(), bounds.data(), K, rowPtrs.data(),
                                        colIdxs.data(), vals.data(), nnz.data(), uniform.data(), bs.data(), N, N, cs.data(), N,
                                        MISPMM_KERNEL_AUTO, acc, gatherMode, comm);
        };
        auto syncAll = [&] {
            for (DeviceSlot &s : slots) mispmmCheckError(mispmm_stream_sync(s.stream));
        };
        const auto t2 = clock::now();
        mispmmCheckError(launch());
        syncAll();
        const auto t3 = clock::now();
        mispmmCheckError(mispmm_set_device(slots[0].ordinal));
        auto *res = new DenseMatrix<DT, MT>(M, N, false, ORDERING::ROW_MAJOR);
        if (gatherMode == MISPMM_GATHER_NONE) {  // C left row-sharded: every device hands back its own rows
            for (int d = 0; d < ngpus; ++d) {
                mispmmCheckError(mispmm_set_device(slots[d].ordinal));
                const size_t off = (size_t)bounds[d] * N, n = (size_t)(bounds[d + 1] - bounds[d]) * N;
                copyBuffer(res->data + off, false, slots[d].c + off, true, n * sizeof(float));
            }
        } else {
            copyBuffer(res->data, false, slots[0].c, true, (size_t)M * N * sizeof(float));
        }
        const auto t4 = clock::now();
        bool correct = false;
        if (ref != nullptr && ref->numRows == M && ref->numCols == N)
            correct = allclose<DT>(res->data, ref->data, res->numElements(), REL_TOL, ABS_TOL);
        delete res;

        SteadyStats steady;
        steady.ngpus = ngpus;
        const int iters = engineOptions().steadyIters;
        if (iters > 0) {
            for (int i = 0; i < 10; ++i) mispmmCheckError(launch());
            syncAll();
            const auto s0 = clock::now();
            for (int i = 0; i < iters; ++i) mispmmCheckError(launch());
            syncAll();
            const double sec = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - s0).count() * 1e-9 / iters;
            steady.iters = iters;
            steady.usPerSpmm = sec * 1e6;
            steady.gflops = 2.0 * a->numNonZero * N / sec / 1e9;
            steady.hbmGBps = (a->numNonZero * 8.0 + (M + 1.0) * 4 + (double)K * N * 4 + (double)M * N * 4) / sec / 1e9;
            steady.rooflineFrac = steady.hbmGBps / (8000.0 * ngpus);
        }
        reportTime(testcase, M, K, a->numNonZero, "CSR", b->ordering, MISPMM_CSR_NUM_KERNELS, ms(t1, t2), ms(t2, t3), ms(t3, t4),
                   correct, &steady);
        if (comm) mispmmCheckError(mispmm_comm_destroy(comm));
        for (DeviceSlot &s : slots) {
            mispmmCheckError(mispmm_set_device(s.ordinal));
            releaseBuffer(s.rowPtrs, true);
            releaseBuffer(s.colIdxs, true);
            releaseBuffer(s.vals, true);
            releaseBuffer(s.b, true);
            releaseBuffer(s.c, true);
            mispmmCheckError(mispmm_stream_destroy(s.stream));
        }
        mispmmCheckError(mispmm_set_device(home));
        return correct;
    }
}

// `cuspmm --ell --gpus n`: the ELL SpMM sharded by ROWS over n devices ("ELL by rows"; no sharded
// counterpart of /root/reference/src/spmm/ell/spmm_ell_k1.cu:10-35 exists) through mispmm_multi_ell_f32.  The class's
// column-major arrays are turned into the row-major view once on the host (mispmm_ell_colmajor_to_rowmajor_host, as
// copy2Device does); rows are cut into contiguous ranges balanced by occupied slots.  Timing sections as above.
template <typename DT, typename MT, typename AccT>
bool spmmELLMultiGpu(int ngpus, int gatherMode, SparseMatrixELL<DT, MT> *a, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        using clock = std::chrono::high_resolution_clock;
        auto ms = [](clock::time_point x, clock::time_point z) {
            return (double)std::chrono::duration_cast<std::chrono::microseconds>(z - x).count() / 1000.0;
        };
        assert(!a->onDevice && !b->onDevice && ngpus >= 1);
        b->toOrdering(ORDERING::ROW_MAJOR);
        int count = 0, home = 0;
        mispmmCheckError(mispmm_device_count(&count));
        mispmmCheckError(mispmm_get_device(&home));
        if (ngpus > count) throw std::runtime_error("--gpus " + std::to_string(ngpus) + ": this node has " + std::to_string(count) + " device(s)");
        if (gatherMode == MISPMM_GATHER_ALL_RCCL_EQUAL) gatherMode = MISPMM_GATHER_ALL_RCCL;   // nnz-balanced ranges: grouped broadcasts
        const uint32_t M = a->numRows, K = a->numCols, N = b->numCols;
        uint32_t width = 0;
        mispmmCheckError(mispmm_ell_colmajor_to_rowmajor_host(M, K, a->maxColNnz, a->rowIdxs, a->data, &width, nullptr, nullptr));
        std::vector<uint32_t> cols((size_t)M * std::max(width, 1u));
        std::vector<float> vals(cols.size());
        if (width) mispmmCheckError(mispmm_ell_colmajor_to_rowmajor_host(M, K, a->maxColNnz, a->rowIdxs, a->data, &width, cols.data(), vals.data()));
        std::vector<uint32_t> occupied((size_t)M + 1, 0), bounds((size_t)ngpus + 1);
        for (uint32_t r = 0; r < M; ++r) {
            uint32_t c = 0;
            for (uint32_t s = 0; s < width; ++s) c += cols[(size_t)r * width + s] != 0xFFFFFFFFu;
            occupied[r + 1] = occupied[r] + c;
        }
        mispmmCheckError(mispmm_shard_rows_by_nnz_host(M, occupied.data(), (uint32_t)ngpus, bounds.data()));

        std::vector<DeviceSlot> slots((size_t)ngpus);            // untimed: every device gets its rows and a replica of B
        std::vector<int> ordinals((size_t)ngpus);
        for (int d = 0; d < ngpus; ++d) {
            DeviceSlot &s = slots[d];
            s.ordinal = ordinals[d] = (home + d) % count;
            mispmmCheckError(mispmm_set_device(s.ordinal));
            mispmmCheckError(mispmm_stream_create(&s.stream));
            const size_t r0 = bounds[d], n = (size_t)(bounds[d + 1] - bounds[d]) * width;
            s.colIdxs = allocateBuffer<uint32_t>(n ? n : 1, true);
            s.vals = allocateBuffer<float>(n ? n : 1, true);
            s.b = allocateBuffer<float>((size_t)K * N, true);
            if (n) {
                copyBuffer(s.colIdxs, true, cols.data() + r0 * width, false, n * sizeof(uint32_t));
                copyBuffer(s.vals, true, vals.data() + r0 * width, false, n * sizeof(float));
            }
            copyBuffer(s.b, true, b->data, false, (size_t)K * N * sizeof(float));
        }
        if (ngpus > 1) mispmmCheckError(mispmm_enable_peer_access((uint32_t)ngpus, ordinals.data()));
        mispmm_comm_t comm = nullptr;
        if (gatherMode == MISPMM_GATHER_ALL_RCCL) mispmmCheckError(mispmm_comm_create(&comm, (uint32_t)ngpus, ordinals.data()));

        const auto t1 = clock::now();
        std::vector<mispmm_stream_t> streams;
        std::vector<const uint32_t *> colIdxs;
        std::vector<const float *> vs, bs;
        std::vector<float *> cs;
        for (DeviceSlot &s : slots) {                            // prolog: C (zero-filled) on every device
            mispmmCheckError(mispmm_set_device(s.ordinal));
            s.c = allocateBuffer<float>((size_t)M * N, true);
            streams.push_back(s.stream);
            colIdxs.push_back(s.colIdxs);
            vs.push_back(s.vals);
            bs.push_back(s.b);
            cs.push_back(s.c);
        }
        const int acc = accModeOf<AccT>();
        auto launch = [&] {
            return mispmm_multi_ell_f32((uint32_t)ngpus, ordinals.data(), streams.data(), bounds.data(), K, width, colIdxs.data(), vs.data(),
                                        bs.data(), N, N, cs.data(), N, MISPMM_KERNEL_AUTO, acc, gatherMode, comm);
        };
        auto syncAll = [&] {
            for (DeviceSlot &s : slots) mispmmCheckError(mispmm_stream_sync(s.stream));
        };
        const auto t2 = clock::now();
        mispmmCheckError(launch());
        syncAll();
        const auto t3 = clock::now();
        auto *res = new DenseMatrix<DT, MT>(M, N, false, ORDERING::ROW_MAJOR);
        if (gatherMode == MISPMM_GATHER_NONE) {
            for (int d = 0; d < ngpus; ++d) {
                mispmmCheckError(mispmm_set_device(slots[d].ordinal));
                const size_t off = (size_t)bounds[d] * N, n = (size_t)(bounds[d + 1] - bounds[d]) * N;
                copyBuffer(res->data + off, false, slots[d].c + off, true, n * sizeof(float));
            }
        } else {
            mispmmCheckError(mispmm_set_device(slots[0].ordinal));
            copyBuffer(res->data, false, slots[0].c, true, (size_t)M * N * sizeof(float));
        }
        const auto t4 = clock::now();
        const bool correct = ref != nullptr && ref->numRows == M && ref->numCols == N &&
                             allclose<DT>(res->data, ref->data, res->numElements(), REL_TOL, ABS_TOL);
        delete res;
        SteadyStats steady;
        steady.ngpus = ngpus;
        const int iters = engineOptions().steadyIters;
        if (iters > 0) {
            for (int i = 0; i < 10; ++i) mispmmCheckError(launch());
            syncAll();
            const auto s0 = clock::now();
            for (int i = 0; i < iters; ++i) mispmmCheckError(launch());
            syncAll();
            const double sec = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - s0).count() * 1e-9 / iters;
            steady.iters = iters;
            steady.usPerSpmm = sec * 1e6;
            steady.gflops = 2.0 * a->numNonZero * N / sec / 1e9;
            steady.hbmGBps = ((double)M * width * 8.0 + (double)K * N * 4 + (double)M * N * 4) / sec / 1e9;
            steady.rooflineFrac = steady.hbmGBps / (8000.0 * ngpus);
        }
        reportTime(testcase, M, K, a->numNonZero, "ELL", b->ordering, MISPMM_ELL_NUM_KERNELS, ms(t1, t2), ms(t2, t3), ms(t3, t4), correct, &steady);
        if (comm) mispmmCheckError(mispmm_comm_destroy(comm));
        for (DeviceSlot &s : slots) {
            mispmmCheckError(mispmm_set_device(s.ordinal));
            releaseBuffer(s.colIdxs, true);
            releaseBuffer(s.vals, true);
            releaseBuffer(s.b, true);
            releaseBuffer(s.c, true);
            mispmmCheckError(mispmm_stream_destroy(s.stream));
        }
        mispmmCheckError(mispmm_set_device(home));
        return correct;
    }
}

template bool spmmELLMultiGpu<float, uint32_t, double>(int, int, SparseMatrixELL<float, uint32_t> *, DenseMatrix<float, uint32_t> *,
                                                       DenseMatrix<float, uint32_t> *);
template bool spmmELLMultiGpu<double, uint32_t, double>(int, int, SparseMatrixELL<double, uint32_t> *, DenseMatrix<double, uint32_t> *,
                                                        DenseMatrix<double, uint32_t> *);
template bool spmmCSRMultiGpu<float, uint32_t, double>(int, int, SparseMatrixCSR<float, uint32_t> *, DenseMatrix<float, uint32_t> *,
                                                       DenseMatrix<float, uint32_t> *);
template bool spmmCSRMultiGpu<double, uint32_t, double>(int, int, SparseMatrixCSR<double, uint32_t> *, DenseMatrix<double, uint32_t> *,
                                                        DenseMatrix<double, uint32_t> *);

}  // namespace cuspmm
