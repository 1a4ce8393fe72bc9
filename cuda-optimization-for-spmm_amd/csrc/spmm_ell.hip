// ELL x dense SpMM for gfx950, row-major ELL (colIdxs/vals are [M x width], pad 0xFFFFFFFF).
//
// Replaces /root/reference/src/spmm/ell/spmm_ell_k1.cu and spmm_ell_k2.cu, which scatter a
// COLUMN-major ELL with one atomicAdd per (slot, output column).  Here the shared row-group
// gather kernel (row_gather.hpp, EllRows) runs it: a row group of G lanes owns one C row, lane i
// fetches slot i of the row (one coalesced load per array, no row-pointer hop: the row's slots
// sit at row * width), the B-row byte offsets are broadcast by lane shuffle and all `width` B
// reads of a row go out back to back as bounds-checked buffer loads (padding slots are dropped
// by the range check, not branched around), 2-D XCD tiling, non-temporal C stores.  Products are summed in slot order = ascending column order, the order in which the
// reference's spmmELLCpu reaches a given C row (spmm_ell.cpp:16-29), so REFERENCE mode is
// bit-identical to it.  Roofline: HBM; algorithmic bytes = M*width*8 + K*N*4 + M*N*4.
#include "row_gather.hpp"
#include "row_stream.hpp"

namespace mispmm {

// 64-bit-address fallback for a B of 2 GiB or more: padding / tail slots are skipped by predicate.
template <int G, int VEC, class Acc>
__global__ __launch_bounds__(256) void ell_wide(uint32_t M, uint32_t width, const uint32_t *__restrict__ colIdxs,
                                                const float *__restrict__ vals, const float *__restrict__ B, uint32_t N,
                                                uint32_t ldb, float *__restrict__ C, uint32_t ldc) {
    constexpr int GROUPS = 256 / G;
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t lane = threadIdx.x % G;
    const uint32_t row = blockIdx.x * GROUPS + threadIdx.x / G;
    const uint32_t col0 = blockIdx.y * (G * VEC) + lane * VEC;
    if (row >= M || col0 >= N) return;
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    for (uint32_t s = 0; s < width; ++s) {
        const uint32_t c = colIdxs[static_cast<size_t>(row) * width + s];
        if (c == 0xFFFFFFFFu) continue;
        const float a = vals[static_cast<size_t>(row) * width + s];
        const vec_t b = load_vec<VEC>(B + static_cast<size_t>(c) * ldb + col0);
#pragma unroll
        for (int v = 0; v < VEC; ++v) Acc::mac(acc[v], a, vec_get<VEC>(b, v));
    }
    vec_t out;
#pragma unroll
    for (int v = 0; v < VEC; ++v) vec_set<VEC>(out, v, Acc::finish(acc[v]));
    store_vec<VEC>(C + static_cast<size_t>(row) * ldc + col0, out);
}

struct EllArgs {
    hipStream_t stream;
    uint32_t M, K, width;
    const uint32_t *colIdxs;
    const float *vals, *B;
    uint32_t N, ldb;
    float *C;
    uint32_t ldc;
};

template <int G, int VEC, class Acc>
static void launch_ell_wide(const EllArgs &a) {
    dim3 grid(ceil_div(a.M, 256 / G), ceil_div(a.N, G * VEC));
    hipLaunchKernelGGL((ell_wide<G, VEC, Acc>), grid, dim3(256), 0, a.stream, a.M, a.width, a.colIdxs, a.vals, a.B, a.N,
                       a.ldb, a.C, a.ldc);
}

template <int VEC, class Acc>
static void launch_ell_wide_g(const EllArgs &a, int g) {
    switch (g) {
        case 8: launch_ell_wide<8, VEC, Acc>(a); break;
        case 16: launch_ell_wide<16, VEC, Acc>(a); break;
        case 32: launch_ell_wide<32, VEC, Acc>(a); break;
        default: launch_ell_wide<64, VEC, Acc>(a); break;
    }
}

template <class Acc>
static void launch_ell_v(const EllArgs &a, int vec) {
    if (static_cast<uint64_t>(a.K) * a.ldb * 4u > 0x7FFFFFFFull) {  // B too large for 32-bit buffer offsets
        const int g = pick_group(a.N, vec);
        if (vec == 4) launch_ell_wide_g<4, Acc>(a, g);
        else if (vec == 2) launch_ell_wide_g<2, Acc>(a, g);
        else launch_ell_wide_g<1, Acc>(a, g);
        return;
    }
    const RowGatherArgs ga{a.stream, a.M, a.K, a.colIdxs, a.vals, a.B, a.N, a.ldb, a.C, a.ldc};
    // more than one round of waves: the persistent row-walking launch (row_stream.hpp)
    if (try_row_stream(ga, a.width, true, std::is_same_v<Acc, AccFast> ? 2 : 1, vec, -1)) return;
    launch_row_gather_auto<Acc>(ga, EllRows{a.width}, vec);
}

}  // namespace mispmm

using namespace mispmm;

extern "C" int mispmm_ell_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t width, const uint32_t *colIdxs,
                              const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc,
                              int kernel, int acc_mode) {
    if (kernel < 0 || kernel > MISPMM_ELL_NUM_KERNELS) return fail(MISPMM_ERR_INVALID_ARG, "ell: unknown kernel id %d", kernel);
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "ell: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (width != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "ell: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    const EllArgs a{as_stream(stream), M, K, width, colIdxs, vals, B, N, ldb, C, ldc};
    const int vec = pick_vec(B, ldb, C, ldc, N);
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_ell_v<AccRefF32>(a, vec);
    else launch_ell_v<AccFast>(a, vec);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}
