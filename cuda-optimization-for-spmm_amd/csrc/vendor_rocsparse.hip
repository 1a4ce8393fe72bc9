// Vendor cross-check: rocSPARSE generic SpMM (alpha = 1, beta = 0, fp32 compute), in the role of
// the reference's cusparseTest (/root/reference/src/engine/cusparse.cu:9-57).  Same three timed
// sections: prolog = handle + descriptors + buffer query + allocation + preprocess, kernel =
// the compute stage + sync, epilog = teardown.  Not on the hot path and not captured in graphs
// (it allocates and synchronises).
#include <rocsparse/rocsparse.h>

#include <chrono>

#include "mispmm_internal.hpp"

using namespace mispmm;

namespace {

// everything the call creates, released on every return path
struct VendorState {
    rocsparse_handle handle = nullptr;
    rocsparse_spmat_descr matA = nullptr;
    rocsparse_dnmat_descr matB = nullptr, matC = nullptr;
    void *buffer = nullptr;
    ~VendorState() {
        if (matA) rocsparse_destroy_spmat_descr(matA);
        if (matB) rocsparse_destroy_dnmat_descr(matB);
        if (matC) rocsparse_destroy_dnmat_descr(matC);
        if (handle) rocsparse_destroy_handle(handle);
        if (buffer) (void)hipFree(buffer);
    }
};

}  // namespace

#define MISPMM_ROCSPARSE_TRY(expr)                                                                        \
    do {                                                                                                  \
        rocsparse_status s_ = (expr);                                                                     \
        if (s_ != rocsparse_status_success)                                                               \
            return fail(MISPMM_ERR_HIP, "%s failed with rocsparse_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
    } while (0)

extern "C" int mispmm_vendor_spmm_f32(mispmm_stream_t stream, int format, uint32_t M, uint32_t K, uint32_t nnz,
                                      uint32_t block_dim, const uint32_t *ptrs_or_rows, const uint32_t *cols,
                                      const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C,
                                      uint32_t ldc, double *pro_us, double *kernel_us, double *epi_us) {
    using clock = std::chrono::high_resolution_clock;
    auto us = [](clock::time_point a, clock::time_point z) {
        return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(z - a).count() / 1000.0;
    };
    if (format < MISPMM_VENDOR_CSR || format > MISPMM_VENDOR_BSR) return fail(MISPMM_ERR_INVALID_ARG, "vendor: unknown format %d", format);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!ptrs_or_rows || !B || !C || (nnz && (!cols || !vals))) return fail(MISPMM_ERR_INVALID_ARG, "vendor: null pointer");
    if (format == MISPMM_VENDOR_BSR && (block_dim == 0 || M % block_dim || K % block_dim))
        return fail(MISPMM_ERR_INVALID_ARG, "vendor: BSR needs square blocks that tile the matrix");
    hipStream_t st = as_stream(stream);

    const auto t1 = clock::now();
    clock::time_point t2, t3;
    {
        VendorState v;
        MISPMM_ROCSPARSE_TRY(rocsparse_create_handle(&v.handle));
        MISPMM_ROCSPARSE_TRY(rocsparse_set_stream(v.handle, st));
        rocsparse_spmm_alg alg = rocsparse_spmm_alg_default;
        void *p0 = const_cast<uint32_t *>(ptrs_or_rows), *p1 = const_cast<uint32_t *>(cols), *pv = const_cast<float *>(vals);
        if (format == MISPMM_VENDOR_CSR) {
            MISPMM_ROCSPARSE_TRY(rocsparse_create_csr_descr(&v.matA, M, K, nnz, p0, p1, pv, rocsparse_indextype_i32,
                                                            rocsparse_indextype_i32, rocsparse_index_base_zero,
                                                            rocsparse_datatype_f32_r));
            alg = rocsparse_spmm_alg_csr_row_split;
        } else if (format == MISPMM_VENDOR_COO) {
            MISPMM_ROCSPARSE_TRY(rocsparse_create_coo_descr(&v.matA, M, K, nnz, p0, p1, pv, rocsparse_indextype_i32,
                                                            rocsparse_index_base_zero, rocsparse_datatype_f32_r));
            alg = rocsparse_spmm_alg_coo_segmented;
        } else {
            // block storage order: the reference's blocks are row-major inside (sparse_bsr.cu:138-155: CUSPARSE_ORDER_ROW)
            MISPMM_ROCSPARSE_TRY(rocsparse_create_bsr_descr(&v.matA, M / block_dim, K / block_dim, nnz, rocsparse_direction_row,
                                                            block_dim, p0, p1, pv, rocsparse_indextype_i32,
                                                            rocsparse_indextype_i32, rocsparse_index_base_zero,
                                                            rocsparse_datatype_f32_r));
            alg = rocsparse_spmm_alg_bsr;
        }
        MISPMM_ROCSPARSE_TRY(rocsparse_create_dnmat_descr(&v.matB, K, N, ldb, const_cast<float *>(B), rocsparse_datatype_f32_r,
                                                          rocsparse_order_row));
        MISPMM_ROCSPARSE_TRY(rocsparse_create_dnmat_descr(&v.matC, M, N, ldc, C, rocsparse_datatype_f32_r, rocsparse_order_row));
        const float alpha = 1.f, beta = 0.f;
        size_t buffer_size = 0;
        MISPMM_ROCSPARSE_TRY(rocsparse_spmm(v.handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, v.matA, v.matB,
                                            &beta, v.matC, rocsparse_datatype_f32_r, alg, rocsparse_spmm_stage_buffer_size,
                                            &buffer_size, nullptr));
        MISPMM_HIP_TRY(hipMalloc(&v.buffer, buffer_size ? buffer_size : 4));
        MISPMM_ROCSPARSE_TRY(rocsparse_spmm(v.handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, v.matA, v.matB,
                                            &beta, v.matC, rocsparse_datatype_f32_r, alg, rocsparse_spmm_stage_preprocess,
                                            &buffer_size, v.buffer));
        MISPMM_HIP_TRY(hipStreamSynchronize(st));
        t2 = clock::now();
        MISPMM_ROCSPARSE_TRY(rocsparse_spmm(v.handle, rocsparse_operation_none, rocsparse_operation_none, &alpha, v.matA, v.matB,
                                            &beta, v.matC, rocsparse_datatype_f32_r, alg, rocsparse_spmm_stage_compute,
                                            &buffer_size, v.buffer));
        MISPMM_HIP_TRY(hipStreamSynchronize(st));
        t3 = clock::now();
    }  // epilog = teardown of everything above
    const auto t4 = clock::now();
    if (pro_us) *pro_us = us(t1, t2);
    if (kernel_us) *kernel_us = us(t2, t3);
    if (epi_us) *epi_us = us(t3, t4);
    return MISPMM_OK;
}
