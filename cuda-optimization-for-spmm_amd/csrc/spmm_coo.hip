// COO x dense SpMM for gfx950.  Entries sorted row-major (the order the reference's converter
// writes, utils/python_utils/convert_mtx.py:172-190).
//
// Replaces /root/reference/src/spmm/coo/spmm_coo_k1.cu (one thread per non-zero, atomicAdd per
// output element).  A sorted COO is a CSR whose row pointers were never materialised: a first
// tiny kernel writes them into a caller-provided workspace (boundary detection over rowIdxs),
// then the rows are processed exactly like CSR rows -- a group of G lanes per C row, B reads as
// dropped-or-live buffer loads, fp32 products summed in storage order (REFERENCE mode = the
// rounding sequence of spmmCOOCpu, spmm_coo.cpp:16-24).  No atomics, deterministic.
// Without a workspace each row group finds its range by binary search instead.
#include "row_gather.hpp"
#include "csr_split.hpp"
#include "csr_hybrid.hpp"

namespace mispmm {

// rowPtrs[r] = first entry index whose row >= r, for r in [0, M]
__global__ __launch_bounds__(256) void coo_row_bounds(uint32_t M, uint32_t nnz, const uint32_t *__restrict__ rowIdxs,
                                                      uint32_t *__restrict__ rowPtrs) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > nnz) return;
    // entry i opens every row r with rowIdxs[i-1] < r <= rowIdxs[i]  (virtual rows -1 and M at the ends)
    const uint32_t first = (i == 0) ? 0u : rowIdxs[i - 1] + 1u;
    const uint32_t last = (i == nnz) ? M : min(rowIdxs[i], M);
    for (uint32_t r = first; r <= last; ++r) rowPtrs[r] = i;
}

__device__ __forceinline__ uint32_t lower_bound_row(const uint32_t *__restrict__ rowIdxs, uint32_t nnz, uint32_t row) {
    uint32_t lo = 0, hi = nnz;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (rowIdxs[mid] < row) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int G, int VEC, class Acc, bool WIDE>
__global__ __launch_bounds__(256) void coo_k1(uint32_t M, uint32_t nnz, const uint32_t *__restrict__ rowIdxs,
                                              const uint32_t *__restrict__ rowPtrs,  // may be null
                                              const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                              const float *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                              float *__restrict__ C, uint32_t ldc) {
    constexpr int GROUPS = 256 / G;
    constexpr int U = (VEC == 4) ? 8 : 16;
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t lane = threadIdx.x % G;
    const uint32_t row = blockIdx.x * GROUPS + threadIdx.x / G;
    const uint32_t col0 = blockIdx.y * (G * VEC) + lane * VEC;
    const bool row_ok = row < M;
    const bool col_ok = col0 < N;
    uint32_t start = 0, end = 0;
    if (row_ok) {
        if (rowPtrs) {
            start = rowPtrs[row];
            end = rowPtrs[row + 1];
        } else {
            start = lower_bound_row(rowIdxs, nnz, row);
            end = lower_bound_row(rowIdxs, nnz, row + 1);
        }
    }
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;
    const float *bcol = B + (col_ok ? col0 : 0);

    for (uint32_t base = start; base < end; base += G) {
        const uint32_t cnt = min(static_cast<uint32_t>(G), end - base);
        const uint32_t mine = base + min(lane, cnt - 1);
        const uint32_t my_col = colIdxs[mine];
        const uint32_t my_off = my_col * (ldb * 4u);
        const float my_val = vals[mine];
        for (uint32_t j = 0; j < cnt; j += U) {
            vec_t bv[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t src = (j + u) & (G - 1);
                const float a = __shfl(my_val, src, G);
                const bool live = j + u < cnt;
                if constexpr (WIDE) {
                    // 64-bit addresses: tail slots re-read the row's last entry and are skipped below
                    const uint32_t c = __shfl(my_col, live ? src : (cnt - 1) & (G - 1), G);
                    bv[u] = load_vec<VEC>(bcol + static_cast<size_t>(c) * ldb);
                    av[u] = a;
                } else {
                    const uint32_t off = __shfl(my_off, src, G);
                    bv[u] = buffer_load_vec<VEC>(rsrc, live ? off + lane_off : kDropLoad, 0);
                    av[u] = live ? a : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!WIDE || j + u < cnt) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) Acc::mac(acc[v], av[u], vec_get<VEC>(bv[u], v));
                }
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

struct CooArgs {
    hipStream_t stream;
    uint32_t M, K, nnz;
    const uint32_t *rowIdxs, *rowPtrs, *colIdxs;
    const float *vals, *B;
    uint32_t N, ldb;
    float *C;
    uint32_t ldc;
};

template <int G, int VEC, class Acc>
static void launch_coo(const CooArgs &a) {
    dim3 grid(ceil_div(a.M, 256 / G), ceil_div(a.N, G * VEC));
    const uint64_t bytes = static_cast<uint64_t>(a.K) * a.ldb * 4u;
    if (bytes > 0x7FFFFFFFull)
        hipLaunchKernelGGL((coo_k1<G, VEC, Acc, true>), grid, dim3(256), 0, a.stream, a.M, a.nnz, a.rowIdxs, a.rowPtrs,
                           a.colIdxs, a.vals, a.B, 0u, a.N, a.ldb, a.C, a.ldc);
    else
        hipLaunchKernelGGL((coo_k1<G, VEC, Acc, false>), grid, dim3(256), 0, a.stream, a.M, a.nnz, a.rowIdxs, a.rowPtrs,
                           a.colIdxs, a.vals, a.B, static_cast<uint32_t>(bytes), a.N, a.ldb, a.C, a.ldc);
}

template <int VEC, class Acc>
static void launch_coo_g(const CooArgs &a, int g) {
    switch (g) {
        case 8: launch_coo<8, VEC, Acc>(a); break;
        case 16: launch_coo<16, VEC, Acc>(a); break;
        case 32: launch_coo<32, VEC, Acc>(a); break;
        default: launch_coo<64, VEC, Acc>(a); break;
    }
}

template <class Acc>
static void launch_coo_v(const CooArgs &a, int vec) {
    const int g = pick_group(a.N, vec);
    if (vec == 4) launch_coo_g<4, Acc>(a, g);
    else if (vec == 2) launch_coo_g<2, Acc>(a, g);
    else launch_coo_g<1, Acc>(a, g);
}

// Rows with their boundaries materialised (prepared-bounds COO, the BSR non-zero list), fp32 arithmetic.  Long rows
// (24 entries per row or more on average) of 16-byte B vectors take the split kernel's shape: one wave per row x 32
// columns on the XCD column grid, 64 entries in flight -- summed in entry order in REFERENCE mode (fp32 product, fp32 add:
// no re-association), split in FAST mode.  GL7d25 K=128 through the CLI: 20.6 -> 10.1 us (COO kernel 2, BSR kernel 3).
template <class Acc>
static void launch_rows(hipStream_t st, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs, const uint32_t *colIdxs,
                        const float *vals, const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc, int vec,
                        const uint32_t *spans = nullptr) {
    static const int split_env = knob_int("MISPMM_SPLIT", 1);
    if (spans || (split_env != 0 && vec == 4 && M != 0 && nnz / M >= 24)) {
        SplitArgs sa{st, M, K, rowPtrs, colIdxs, vals, B, N, ldb, C, ldc};
        sa.spans = spans;
        sa.numSpans = spans ? M : 0u;
        launch_split<Acc>(sa);
        return;
    }
    const RowGatherArgs ga{st, M, K, colIdxs, vals, B, N, ldb, C, ldc, M ? nnz / M : 0u};
    launch_row_gather_auto<Acc>(ga, CsrRows{rowPtrs}, vec);
}

}  // namespace mispmm

using namespace mispmm;

extern "C" int mispmm_coo_row_bounds(mispmm_stream_t stream, uint32_t M, uint32_t nnz, const uint32_t *rowIdxs,
                                     uint32_t *rowPtrs_out) {
    if (!rowPtrs_out) return fail(MISPMM_ERR_INVALID_ARG, "coo_row_bounds: output is null");
    if (nnz != 0 && !rowIdxs) return fail(MISPMM_ERR_INVALID_ARG, "coo_row_bounds: rowIdxs is null");
    hipLaunchKernelGGL(coo_row_bounds, dim3(ceil_div(nnz + 1, 256)), dim3(256), 0, as_stream(stream), M, nnz, rowIdxs, rowPtrs_out);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

extern "C" int mispmm_coo_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowIdxs,
                              const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb,
                              float *C, uint32_t ldc, uint32_t *rowPtrs_workspace, int kernel, int acc_mode) {
    if (kernel < 0 || kernel > MISPMM_COO_NUM_KERNELS) return fail(MISPMM_ERR_INVALID_ARG, "coo: unknown kernel id %d", kernel);
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "coo: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (nnz != 0 && (!rowIdxs || !colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "coo: null index or value array");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (kernel == 2 && !rowPtrs_workspace) return fail(MISPMM_ERR_INVALID_ARG, "coo: kernel 2 needs the prepared row boundaries");
    hipStream_t st = as_stream(stream);
    if (rowPtrs_workspace && kernel != 2) {
        hipLaunchKernelGGL(coo_row_bounds, dim3(ceil_div(nnz + 1, 256)), dim3(256), 0, st, M, nnz, rowIdxs,
                           rowPtrs_workspace);
        MISPMM_LAUNCH_CHECK();
    }
    const CooArgs a{st, M, K, nnz, rowIdxs, rowPtrs_workspace, colIdxs, vals, B, N, ldb, C, ldc};
    const int vec = pick_vec(B, ldb, C, ldc, N);
    if (rowPtrs_workspace && static_cast<uint64_t>(K) * ldb * 4u <= 0x7FFFFFFFull) {
        // with its row bounds materialised a sorted COO is a CSR: same kernel, same order of sums
        if (acc_mode == MISPMM_ACC_REFERENCE) launch_rows<AccRefF32>(st, M, K, nnz, rowPtrs_workspace, colIdxs, vals, B, N, ldb, C, ldc, vec);
        else launch_rows<AccFast>(st, M, K, nnz, rowPtrs_workspace, colIdxs, vals, B, N, ldb, C, ldc, vec);
        MISPMM_LAUNCH_CHECK();
        return MISPMM_OK;
    }
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_coo_v<AccRefF32>(a, vec);
    else launch_coo_v<AccFast>(a, vec);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

// Rows with the fp32 arithmetic of COO / ELL / BSR on the split kernel's shape, walked longest first: spans = one
// (row, start, end, 0) per row from mispmm_csr_spans_by_length_host with share_len = 0xFFFFFFFF (a sum of this arithmetic
// cannot be dealt to several waves).  What the host layers call for a long-row COO, ELL or BSR list.
extern "C" int mispmm_rows_split_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *colIdxs,
                                     const float *vals, const uint32_t *spans, uint32_t numSpans, const float *B, uint32_t N,
                                     uint32_t ldb, float *C, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "rows_split: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!spans || !aligned16(spans)) return fail(MISPMM_ERR_INVALID_ARG, "rows_split: spans is null or not 16-byte aligned");
    if (numSpans != M) return fail(MISPMM_ERR_INVALID_ARG, "rows_split: %u spans for %u rows (one per row: share_len = 0xFFFFFFFF)", numSpans, M);
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "rows_split: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull) return fail(MISPMM_ERR_UNSUPPORTED, "rows_split: B of 2 GiB or more");
    if (pick_vec(B, ldb, C, ldc, N) != 4) return fail(MISPMM_ERR_UNSUPPORTED, "rows_split: B and C rows must be 16-byte vectors");
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_rows<AccRefF32>(as_stream(stream), M, K, nnz, nullptr, colIdxs, vals, B, N, ldb, C, ldc, 4, spans);
    else launch_rows<AccFast>(as_stream(stream), M, K, nnz, nullptr, colIdxs, vals, B, N, ldb, C, ldc, 4, spans);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

// The same list in one launch by two bodies (csr_hybrid.hpp): positions [0, numLongSpans) -- the long rows -- on the split
// kernel's shape, the others by the row-gather body (a lane group per row, one running fp32 sum per element: the same
// arithmetic in the same order).  MISPMM_ERR_UNSUPPORTED without a message where the shape has no such launch.
extern "C" int mispmm_rows_hybrid_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *colIdxs, const float *vals,
                                      const uint32_t *spans, uint32_t numSpans, uint32_t numLongSpans, const float *B, uint32_t N,
                                      uint32_t ldb, float *C, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "rows_hybrid: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!spans || !aligned16(spans)) return fail(MISPMM_ERR_INVALID_ARG, "rows_hybrid: spans is null or not 16-byte aligned");
    if (numSpans != M) return fail(MISPMM_ERR_INVALID_ARG, "rows_hybrid: %u spans for %u rows (one per row: share_len = 0xFFFFFFFF)", numSpans, M);
    if (numLongSpans > numSpans || (numLongSpans % 4u != 0 && numLongSpans != numSpans))
        return fail(MISPMM_ERR_INVALID_ARG, "rows_hybrid: %u long spans of %u: a multiple of 4 is needed", numLongSpans, numSpans);
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "rows_hybrid: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull || pick_vec(B, ldb, C, ldc, N) != 4) return MISPMM_ERR_UNSUPPORTED;
    const HybridArgs a{as_stream(stream), M, K, colIdxs, vals, B, N, ldb, C, ldc, spans, numSpans, numLongSpans};
    const bool taken = acc_mode == MISPMM_ACC_REFERENCE ? launch_hybrid<AccRefF32>(a) : launch_hybrid<AccFast>(a);
    if (!taken) return MISPMM_ERR_UNSUPPORTED;
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

// An ELL without its padding is the same thing once more: rows with their boundaries, fp32 product and fp32 add in list order.
extern "C" int mispmm_ell_compact_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                                      const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb,
                                      float *C, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "ell_compact: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!rowPtrs) return fail(MISPMM_ERR_INVALID_ARG, "ell_compact: rowPtrs is null");
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "ell_compact: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull)
        return fail(MISPMM_ERR_UNSUPPORTED, "ell_compact: B of 2 GiB or more: use mispmm_ell_f32");
    const int vec = pick_vec(B, ldb, C, ldc, N);
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_rows<AccRefF32>(as_stream(stream), M, K, nnz, rowPtrs, colIdxs, vals, B, N, ldb, C, ldc, vec);
    else launch_rows<AccFast>(as_stream(stream), M, K, nnz, rowPtrs, colIdxs, vals, B, N, ldb, C, ldc, vec);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

// The zero-skipping BSR path: the block entries that are not zero, listed per C row in the reference's order of
// addition (mispmm_bsr_nonzeros_host), are a CSR whose REFERENCE arithmetic is COO's (fp32 product, fp32 add), so
// it runs on the same instantiations as the prepared-bounds COO kernel above.
extern "C" int mispmm_bsr_nonzeros_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                                       const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb,
                                       float *C, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "bsr_nonzeros: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!rowPtrs) return fail(MISPMM_ERR_INVALID_ARG, "bsr_nonzeros: rowPtrs is null");
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "bsr_nonzeros: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull)
        return fail(MISPMM_ERR_UNSUPPORTED, "bsr_nonzeros: B of 2 GiB or more: use mispmm_bsr_f32");
    const int vec = pick_vec(B, ldb, C, ldc, N);
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_rows<AccRefF32>(as_stream(stream), M, K, nnz, rowPtrs, colIdxs, vals, B, N, ldb, C, ldc, vec);
    else launch_rows<AccFast>(as_stream(stream), M, K, nnz, rowPtrs, colIdxs, vals, B, N, ldb, C, ldc, vec);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}
