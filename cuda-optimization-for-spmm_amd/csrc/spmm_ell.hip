// ELL x dense SpMM for gfx950, row-major ELL (colIdxs/vals are [M x width], pad 0xFFFFFFFF).
//
// Replaces /root/reference/src/spmm/ell/spmm_ell_k1.cu and spmm_ell_k2.cu, which scatter a
// COLUMN-major ELL with one atomicAdd per (slot, output column).  Here a row group of G lanes
// owns one C row: lane i fetches slot i of the row (one coalesced load per array), the B-row byte
// offsets are broadcast by lane shuffle and all `width` B reads of a row go out back to back as
// bounds-checked buffer loads (padding slots are dropped by the range check, not branched
// around).  Products are summed in slot order = ascending column order, the order in which the
// reference's spmmELLCpu reaches a given C row (spmm_ell.cpp:16-29), so REFERENCE mode is
// bit-identical to it.  Roofline: HBM; algorithmic bytes = M*width*8 + K*N*4 + M*N*4.
#include "spmm_common.hpp"

namespace mispmm {

template <int G, int VEC, class Acc>
__global__ __launch_bounds__(256) void ell_k1(uint32_t M, uint32_t width, const uint32_t *__restrict__ colIdxs,
                                              const float *__restrict__ vals, const float *__restrict__ B,
                                              uint32_t b_bytes, uint32_t N, uint32_t ldb, float *__restrict__ C,
                                              uint32_t ldc) {
    constexpr int GROUPS = 256 / G;
    constexpr int U = (VEC == 4) ? 8 : 16;
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t lane = threadIdx.x % G;
    const uint32_t row = blockIdx.x * GROUPS + threadIdx.x / G;
    const uint32_t col0 = blockIdx.y * (G * VEC) + lane * VEC;
    const bool row_ok = row < M;
    const bool col_ok = col0 < N;
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;
    const size_t row_base = static_cast<size_t>(row_ok ? row : 0) * width;
    const uint32_t row_cnt = row_ok ? width : 0;

    for (uint32_t base = 0; base < row_cnt; base += G) {
        const uint32_t cnt = min(static_cast<uint32_t>(G), row_cnt - base);
        const size_t mine = row_base + base + min(lane, cnt - 1);
        const uint32_t my_col = colIdxs[mine];
        const float my_val = vals[mine];
        // padding (0xFFFFFFFF) becomes a dropped load with a zero coefficient
        const uint32_t my_off = (my_col == 0xFFFFFFFFu) ? kDropLoad : my_col * (ldb * 4u);
        const float my_a = (my_col == 0xFFFFFFFFu) ? 0.f : my_val;
        for (uint32_t j = 0; j < cnt; j += U) {
            vec_t bv[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t src = (j + u) & (G - 1);
                const uint32_t off = __shfl(my_off, src, G);
                const float a = __shfl(my_a, src, G);
                const bool live = (j + u < cnt) && (off != kDropLoad);
                av[u] = live ? a : 0.f;
                bv[u] = buffer_load_vec<VEC>(rsrc, live ? off + lane_off : kDropLoad, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) Acc::mac(acc[v], av[u], vec_get<VEC>(bv[u], v));
            }
        }
    }
    if (row_ok && col_ok) {
        vec_t out;
#pragma unroll
        for (int v = 0; v < VEC; ++v) vec_set<VEC>(out, v, Acc::finish(acc[v]));
        store_vec<VEC>(C + static_cast<size_t>(row) * ldc + col0, out);
    }
}

// 64-bit-address fallback for a B of 4 GiB or more: padding / tail slots are skipped by predicate.
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
static void launch_ell(const EllArgs &a, bool wide) {
    dim3 grid(ceil_div(a.M, 256 / G), ceil_div(a.N, G * VEC));
    if (wide) {
        hipLaunchKernelGGL((ell_wide<G, VEC, Acc>), grid, dim3(256), 0, a.stream, a.M, a.width, a.colIdxs, a.vals, a.B,
                           a.N, a.ldb, a.C, a.ldc);
    } else {
        const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
        hipLaunchKernelGGL((ell_k1<G, VEC, Acc>), grid, dim3(256), 0, a.stream, a.M, a.width, a.colIdxs, a.vals, a.B,
                           b_bytes, a.N, a.ldb, a.C, a.ldc);
    }
}

template <int VEC, class Acc>
static void launch_ell_g(const EllArgs &a, int g, bool wide) {
    switch (g) {
        case 8: launch_ell<8, VEC, Acc>(a, wide); break;
        case 16: launch_ell<16, VEC, Acc>(a, wide); break;
        case 32: launch_ell<32, VEC, Acc>(a, wide); break;
        default: launch_ell<64, VEC, Acc>(a, wide); break;
    }
}

template <class Acc>
static void launch_ell_v(const EllArgs &a, int vec) {
    const bool wide = static_cast<uint64_t>(a.K) * a.ldb * 4u > 0x7FFFFFFFull;
    const int g = pick_group(a.N, vec);
    if (vec == 4) launch_ell_g<4, Acc>(a, g, wide);
    else if (vec == 2) launch_ell_g<2, Acc>(a, g, wide);
    else launch_ell_g<1, Acc>(a, g, wide);
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
