// Dense helpers on the device: layout change and precision conversion around the SpMM kernels.
#include "mispmm_internal.hpp"

namespace mispmm {

// dst[cols x rows] = src[rows x cols]^T through a padded 64x64 LDS tile: reads and writes are both
// row-contiguous 256-byte segments.  Replaces the host double loop of DenseMatrix::toOrdering
// (/root/reference/src/formats/dense.cu:159-173).
__global__ __launch_bounds__(256) void transpose_f32(uint32_t rows, uint32_t cols, const float *__restrict__ src,
                                                     float *__restrict__ dst) {
    __shared__ float tile[64][65];
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    const uint32_t r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll 4
    for (uint32_t i = ty; i < 64; i += 4) {
        const uint32_t r = r0 + i, c = c0 + tx;
        if (r < rows && c < cols) tile[i][tx] = src[static_cast<size_t>(r) * cols + c];
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t i = ty; i < 64; i += 4) {
        const uint32_t c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[static_cast<size_t>(c) * rows + r] = tile[tx][i];
    }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(size_t n, const float *__restrict__ src,
                                                          uint16_t *__restrict__ dst) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        dst[i] = __builtin_bit_cast(uint16_t, static_cast<__bf16>(src[i]));  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(size_t n, const uint16_t *__restrict__ src,
                                                          float *__restrict__ dst) {
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        dst[i] = __uint_as_float(static_cast<uint32_t>(src[i]) << 16);
    }
}

}  // namespace mispmm

using namespace mispmm;

extern "C" int mispmm_dense_transpose_f32(mispmm_stream_t stream, uint32_t rows, uint32_t cols, const float *src,
                                          float *dst) {
    if (rows == 0 || cols == 0) return MISPMM_OK;
    if (!src || !dst) return fail(MISPMM_ERR_INVALID_ARG, "transpose: null pointer");
    if (src == dst) return fail(MISPMM_ERR_INVALID_ARG, "transpose: in-place is not supported");
    dim3 grid(ceil_div(cols, 64), ceil_div(rows, 64));
    if (grid.y > 65535u) return fail(MISPMM_ERR_UNSUPPORTED, "transpose: more than 65535*64 rows");
    hipLaunchKernelGGL(transpose_f32, grid, dim3(256), 0, as_stream(stream), rows, cols, src, dst);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

static uint32_t elementwise_grid(size_t n) {
    const size_t blocks = (n + 255) / 256;
    return static_cast<uint32_t>(blocks < 2048 ? blocks : 2048);  // grid-stride past 8 blocks per CU
}

extern "C" int mispmm_f32_to_bf16(mispmm_stream_t stream, size_t n, const float *src, uint16_t *dst) {
    if (n == 0) return MISPMM_OK;
    if (!src || !dst) return fail(MISPMM_ERR_INVALID_ARG, "f32_to_bf16: null pointer");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(elementwise_grid(n)), dim3(256), 0, as_stream(stream), n, src, dst);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

extern "C" int mispmm_bf16_to_f32(mispmm_stream_t stream, size_t n, const uint16_t *src, float *dst) {
    if (n == 0) return MISPMM_OK;
    if (!src || !dst) return fail(MISPMM_ERR_INVALID_ARG, "bf16_to_f32: null pointer");
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(elementwise_grid(n)), dim3(256), 0, as_stream(stream), n, src, dst);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}
