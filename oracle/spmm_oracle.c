/*
 * oracle/spmm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's sequential SpMM engine
 * (mli43/Cuda-Optimization-for-SpMM @ 2024-12-18).  Only tests/, the smoke()
 * check in __graft_entry__.py and the `cpu_baseline` leg of bench.py may load
 * this library, and only as the checker / the timed CPU baseline.  The product
 * path (libmispmm.so, HIP) never links or calls anything in here.
 *
 * Parity pin: every function below is checked in tests/test_oracle.py against
 *   - the reference's own committed fixtures data/small_10x10/result.expect and
 *     data/small_32x32/result.expect (copied as data under tests/golden/), and
 *   - golden vectors generated in the build container by running the
 *     reference's Python tooling (utils/python_utils/convert_mtx.py,
 *     convert_matrix.py, validate.py) -- see tests/golden/make_golden.py.
 * The reference's C++ CPU path itself is unbuildable here (it needs the CUDA
 * toolkit headers, cuSPARSE and libtorch: include/cuda_utils.hpp:3-7,
 * include/spmm_cusparse.hpp:1), so there is no oracle/_ref.
 *
 * Numerics contract (what "same as the reference" means, per function):
 *   CSR : float*float product rounded to fp32, widened, summed in a double
 *         accumulator in CSR storage order, cast to fp32 once
 *         (src/spmm/csr/spmm_csr.cpp:15-27 with AccT=double, src/main.cu:196).
 *   COO / ELL / BSR : fp32 `+=` of an fp32-rounded product into a zeroed C,
 *         in storage order (spmm_coo.cpp:16-24, spmm_ell.cpp:16-29,
 *         spmm_bsr.cpp:17-38).  AccT is unused by the reference there.
 * Compile with -ffp-contract=off so no mul+add pair is fused: the reference's
 * default x86-64 build has no FMA instructions to contract into.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* C = A_csr * B.  B row-major [aNumCols x bNumCols], C row-major
 * [aNumRows x bNumCols], overwritten (no dependence on prior C).
 * Loop nest, accumulator type and rounding follow
 * /root/reference/src/spmm/csr/spmm_csr.cpp:15-27 exactly (row, then output
 * column, then the row's non-zeros: the column-of-B inner walk is the
 * reference's, and is what the cpu_baseline times). */
void oracle_spmm_csr_f32(uint32_t aNumRows, const uint32_t *rowPtrs,
                         const uint32_t *colIdxs, const float *aData,
                         const float *bData, uint32_t bNumCols, float *cData)
{
    for (uint32_t r = 0; r < aNumRows; r++) {
        uint32_t row_start = rowPtrs[r];
        uint32_t row_end = rowPtrs[r + 1];
        for (uint32_t c = 0; c < bNumCols; c++) {
            double acc = 0.f;
            for (uint32_t idx = row_start; idx < row_end; idx++) {
                uint32_t k = colIdxs[idx];
                float prod = aData[idx] * bData[(size_t)k * bNumCols + c];
                acc += prod;
            }
            cData[(size_t)r * bNumCols + c] = (float)acc;
        }
    }
}

/* The same CSR engine with the row loop split over `threads` host threads (OpenMP static schedule).
 * Rows are independent and each row's arithmetic is the sequential function's, so the result is
 * bit-identical; this is the "all host cores" figure next to the sequential engine in bench.py
 * (the reference itself is single-threaded: src/engine/engine.cpp:31 calls one spmm<F>Cpu). */
void oracle_spmm_csr_f32_mt(uint32_t aNumRows, const uint32_t *rowPtrs,
                            const uint32_t *colIdxs, const float *aData,
                            const float *bData, uint32_t bNumCols, float *cData, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t r = 0; r < (int64_t)aNumRows; r++) {
        uint32_t row_start = rowPtrs[r];
        uint32_t row_end = rowPtrs[r + 1];
        for (uint32_t c = 0; c < bNumCols; c++) {
            double acc = 0.f;
            for (uint32_t idx = row_start; idx < row_end; idx++) {
                uint32_t k = colIdxs[idx];
                float prod = aData[idx] * bData[(size_t)k * bNumCols + c];
                acc += prod;
            }
            cData[(size_t)r * bNumCols + c] = (float)acc;
        }
    }
}

/* C += A_coo * B, C must be zeroed by the caller (the reference allocates a
 * zero-filled C: src/engine/engine.cpp:20, src/formats/dense.cu:234-251).
 * /root/reference/src/spmm/coo/spmm_coo.cpp:16-24. */
void oracle_spmm_coo_f32(uint32_t numNonZero, const uint32_t *rowIdxs,
                         const uint32_t *colIdxs, const float *aData,
                         const float *bData, uint32_t bNumCols, float *cData)
{
    for (uint32_t idx = 0; idx < numNonZero; idx++) {
        uint32_t r = rowIdxs[idx];
        uint32_t c = colIdxs[idx];
        float value = aData[idx];
        for (uint32_t j = 0; j < bNumCols; j++) {
            float prod = value * bData[(size_t)c * bNumCols + j];
            cData[(size_t)r * bNumCols + j] += prod;
        }
    }
}

/* C += A_ell * B for the reference's COLUMN-major ELL: rowIdxs/aData are
 * [aNumCols x maxColNnz], pad row index 0xFFFFFFFF (text "-1"), pad value 0.
 * C must be zeroed.  /root/reference/src/spmm/ell/spmm_ell.cpp:16-29
 * (`int row = ...; if (row >= 0)`). */
void oracle_spmm_ell_colmajor_f32(uint32_t aNumCols, uint32_t maxColNnz,
                                  const uint32_t *rowIdxs, const float *aData,
                                  const float *bData, uint32_t bNumCols,
                                  float *cData)
{
    for (uint32_t col = 0; col < aNumCols; col++) {
        for (uint32_t slot = 0; slot < maxColNnz; slot++) {
            int row = (int)rowIdxs[(size_t)col * maxColNnz + slot];
            float value = aData[(size_t)col * maxColNnz + slot];
            if (row >= 0) {
                for (uint32_t j = 0; j < bNumCols; j++) {
                    float prod = value * bData[(size_t)col * bNumCols + j];
                    cData[(size_t)row * bNumCols + j] += prod;
                }
            }
        }
    }
}

/* C += A_bsr * B.  Blocks are row-major blockRowSize x blockColSize, stored
 * contiguously in block-CSR order.  C must be zeroed.
 * /root/reference/src/spmm/bsr/spmm_bsr.cpp:17-38. */
void oracle_spmm_bsr_f32(uint32_t numBlockRows, uint32_t blockRowSize,
                         uint32_t blockColSize, const uint32_t *blockRowPtrs,
                         const uint32_t *blockColIdxs, const float *aData,
                         const float *bData, uint32_t bNumCols, float *cData)
{
    for (uint32_t blockRow = 0; blockRow < numBlockRows; blockRow++) {
        uint32_t blockRowStart = blockRowPtrs[blockRow];
        uint32_t blockRowEnd = blockRowPtrs[blockRow + 1];
        for (uint32_t b = blockRowStart; b < blockRowEnd; b++) {
            uint32_t blockCol = blockColIdxs[b];
            const float *blockData =
                aData + (size_t)blockRowSize * blockColSize * b;
            uint32_t denseRowStart = blockRow * blockRowSize;
            uint32_t denseColStart = blockCol * blockColSize;
            for (uint32_t i = 0; i < blockRowSize; i++) {
                uint32_t ar = denseRowStart + i;
                for (uint32_t j = 0; j < blockColSize; j++) {
                    uint32_t ac = denseColStart + j;
                    float a = blockData[(size_t)i * blockColSize + j];
                    for (uint32_t bc = 0; bc < bNumCols; bc++) {
                        float prod = a * bData[(size_t)ac * bNumCols + bc];
                        cData[(size_t)ar * bNumCols + bc] += prod;
                    }
                }
            }
        }
    }
}

/* Host transpose between the two dense orderings, the reference's
 * DenseMatrix::toOrdering (/root/reference/src/formats/dense.cu:159-173).
 * to_col_major != 0: src is row-major, dst col-major; else the reverse. */
void oracle_dense_reorder_f32(uint32_t numRows, uint32_t numCols,
                              const float *src, float *dst, int to_col_major)
{
    for (uint32_t r = 0; r < numRows; r++) {
        for (uint32_t c = 0; c < numCols; c++) {
            size_t rm = (size_t)r * numCols + c;
            size_t cm = (size_t)c * numRows + r;
            if (to_col_major)
                dst[cm] = src[rm];
            else
                dst[rm] = src[cm];
        }
    }
}

/* torch::allclose(c, ref, rtol, atol) as the reference's wrappers call it
 * (e.g. /root/reference/src/spmm/csr/spmm_csr_k3.cu:97-99, tolerances
 * include/utils.hpp:10-11): every element |c - ref| <= atol + rtol*|ref|,
 * NaNs never close.  Returns 1 when close. */
int oracle_allclose_f32(size_t n, const float *c, const float *ref,
                        double rtol, double atol)
{
    for (size_t i = 0; i < n; i++) {
        double a = c[i], b = ref[i];
        if (a != a || b != b)
            return 0;
        if (a == b)
            continue;
        double d = a - b;
        if (d < 0)
            d = -d;
        double bb = b < 0 ? -b : b;
        if (!(d <= atol + rtol * bb))
            return 0;
    }
    return 1;
}

#ifdef __cplusplus
}
#endif
