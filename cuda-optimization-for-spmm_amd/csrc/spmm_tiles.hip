// CSR x dense with the B rows of a GROUP of rows staged in LDS: the kernel shape BASELINE.json's north_star names for the
// row-parallel kernels ("coalesced HBM loads of the dense B panel staged into LDS tiles"; its ancestor is the shared-memory
// staging of /root/reference/src/spmm/csr/spmm_csr_k4.cu:26-62).  Rounds 1-3 measured the BOUND on what such a tile could save
// (rows of n4c6-b13 share B rows 1.3-1.5x after a greedy grouping) and did not build it; round 4 builds it to have the number.
//
// Upload time (mispmm_csr_tiles_host): rows are grouped greedily into tiles of <= 16 rows whose DISTINCT columns number <= 128;
// per tile the column list, per entry the position of its column in that list (one byte).  Kernel: one 256-thread workgroup
// per (tile, 64-column part): every 16-lane group fetches its plan row's (slot, value) entries, all groups together stage the
// tile's <= 128 B-row slices (256 B each) into 32 KiB of LDS -- each slice ONCE however many rows of the tile read it --, one
// barrier, then every lane group sums its row's products in storage order reading B from LDS (ds_read_b128: a 16-lane group
// reads one 256-byte slice, conflict-free) and stores its C row with a non-temporal store.  Same arithmetic, same order as
// the row-gather kernel: REFERENCE mode is bit-exact.  Rows of one width (<= 16 entries) only -- this is an experiment.
#include "row_gather.hpp"

namespace mispmm {

constexpr uint32_t kTileCols = 128;   // distinct columns per tile = the LDS budget (x 256 B = 32 KiB); a tile holds <= 16 rows (one per lane group)

// DMA: the slices go from the buffer load straight into LDS (buffer_load_dwordx4 ... lds: a wave's four lane groups stage four
// CONSECUTIVE list positions = 1 KiB contiguous, exactly what one LDS-DMA instruction writes) instead of through registers and
// ds_write_b128 -- measured beside the register form (profiles/r4/lds_tile_ab.log).
template <class Acc, int W, bool DMA>
__global__ __launch_bounds__(256, 3) void csr_lds_tile_kernel(
    const uint32_t *__restrict__ tileRowPtrs, const uint32_t *__restrict__ tileColPtrs, const uint32_t *__restrict__ tileCols,
    const uint8_t *__restrict__ slots, const float *__restrict__ vals, const float *__restrict__ B, uint32_t b_bytes, uint32_t width,
    uint32_t ldb, uint32_t numTiles, uint32_t tiling /* bits 0..7 log2 P, bits 8.. tiles per row part */, uint32_t cols_per_part,
    uint32_t N, const uint32_t *__restrict__ rowMap, float *__restrict__ C, uint32_t c_bytes, uint32_t ldc) {
    constexpr int G = 16, VEC = 4, JMAX = kTileCols / 16;
    __shared__ f32x4 tile[kTileCols * G];                     // 128 slices x 16 lanes x 16 B = 32 KiB
    const uint32_t lane = threadIdx.x % G, g = threadIdx.x / G;
    const uint32_t xcd = blockIdx.x & 7u, log2p = tiling & 0xFFu, chunk = tiling >> 8;
    const uint32_t p = xcd & ((1u << log2p) - 1u), q = xcd >> log2p;
    const uint32_t t = p * chunk + (blockIdx.x >> 3);
    if (t >= numTiles) return;
    const uint32_t col0 = q * cols_per_part + blockIdx.y * (G * VEC) + lane * VEC;
    const bool col_ok = col0 < min(N, (q + 1) * cols_per_part);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;
    const uint32_t ldb4 = ldb * 4u;
    const rsrc_t brs = make_rsrc(B, b_bytes);
    const uint32_t r0 = tileRowPtrs[t], nrows = tileRowPtrs[t + 1] - r0;
    const uint32_t d0 = tileColPtrs[t], D = tileColPtrs[t + 1] - d0;
    // this lane group's row: its entries as (slot, value), issued now and used behind the barrier
    const uint32_t prow = r0 + g;
    const bool row_ok = g < nrows;
    const size_t e = static_cast<size_t>(row_ok ? prow : r0) * width + min(lane, width - 1u);
    const uint32_t my_slot = slots[e];
    const float my_val = vals[e];
    const uint32_t orow = row_ok ? rowMap[prow] : 0u;
    // stage: list position j = g + 16 k of the tile goes through lane group g (all 16 lanes read the same column index)
    uint32_t cidx[JMAX];
#pragma unroll
    for (int k = 0; k < JMAX; ++k) {
        const uint32_t j = g + 16u * k;
        cidx[k] = tileCols[d0 + min(j, D - 1u)];
    }
    if constexpr (DMA) {
        using lds_ptr_t = __attribute__((address_space(3))) void *;
        const uint32_t wave_first = (threadIdx.x >> 6) * 4u;          // the wave's first list position of pass k = 0
#pragma unroll
        for (int k = 0; k < JMAX; ++k) {
            const uint32_t j = g + 16u * k;
            // LDS destination = base + lane-in-wave * 16: position wave_first + 16 k of the tile image, 1 KiB per wave instruction
            const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(tile)) + (wave_first + 16u * k) * (G * 16u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, reinterpret_cast<lds_ptr_t>(static_cast<uintptr_t>(lds_base)), 16,
                                                     j < D ? cidx[k] * ldb4 + lane_off : kDropLoad, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's slices have landed
    } else {
        f32x4 v[JMAX];
#pragma unroll
        for (int k = 0; k < JMAX; ++k) {
            const uint32_t j = g + 16u * k;
            v[k] = buffer_load_vec<VEC>(brs, j < D ? cidx[k] * ldb4 + lane_off : kDropLoad, 0);
        }
#pragma unroll
        for (int k = 0; k < JMAX; ++k) {
            const uint32_t j = g + 16u * k;
            if (j < D) tile[j * G + lane] = v[k];
        }
    }
    __syncthreads();
    typename Acc::T acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0;
    static_for<0, W>([&](auto s) {
        constexpr int S = decltype(s)::value;
        if (S < static_cast<int>(width)) {                    // wave-uniform: every row holds `width` entries
            const uint32_t slot = group_bcast<G, S>(my_slot);
            const float a = __builtin_bit_cast(float, group_bcast<G, S>(__builtin_bit_cast(uint32_t, my_val)));
            const f32x4 b = tile[slot * G + lane];
            if constexpr (std::is_same_v<Acc, AccRefWide>) {
                Acc::mac4(acc, a, b[0], b[1], b[2], b[3]);
            } else {
#pragma unroll
                for (int i = 0; i < VEC; ++i) Acc::mac(acc[i], a, b[i]);
            }
        }
    });
    f32x4 out;
#pragma unroll
    for (int i = 0; i < VEC; ++i) out[i] = Acc::finish(acc[i]);
    buffer_store_vec_c<VEC>(make_rsrc(C, c_bytes), (row_ok && col_ok) ? (orow * ldc + col0) * 4u : kDropLoad, out);
}

template <class Acc>
static void launch_tiles(hipStream_t st, uint32_t numTiles, const uint32_t *tileRowPtrs, const uint32_t *tileColPtrs, const uint32_t *tileCols,
                         const uint8_t *slots, const float *vals, const uint32_t *rowMap, uint32_t width, uint32_t K, const float *B, uint32_t N,
                         uint32_t ldb, float *C, uint32_t M, uint32_t ldc) {
    const XcdTiling t = xcd_tiling(N, 4, K);
    const uint32_t cols_per_part = N / t.q;
    const uint32_t chunk = ceil_div(numTiles, 1u << t.log2p);
    dim3 grid(8u * chunk, ceil_div(cols_per_part, 64u));
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(K) * ldb * 4u), c_bytes = static_cast<uint32_t>(static_cast<uint64_t>(M) * ldc * 4u);
    // staging through registers (ds_write_b128) or by LDS-DMA: MISPMM_TILE_DMA=0 / 1 in the tuning build (default: DMA, the faster)
    static const bool dma = knob_int("MISPMM_TILE_DMA", 1) != 0;
    note_kernel("csr_lds_tile<%s,%s> xcd %ux%u, %u tiles", acc_tag<Acc>(), dma ? "lds-dma" : "ds_write", 1u << t.log2p, t.q, numTiles);
#define MISPMM_TILE_LAUNCH(WW, DD)                                                                                                          \
    hipLaunchKernelGGL((csr_lds_tile_kernel<Acc, WW, DD>), grid, dim3(256), 0, st, tileRowPtrs, tileColPtrs, tileCols, slots, vals, B, b_bytes, \
                       width, ldb, numTiles, t.log2p | (chunk << 8), cols_per_part, N, rowMap, C, c_bytes, ldc)
#define MISPMM_TILE_PICK(DD)                     \
    do {                                         \
        if (width > 14) MISPMM_TILE_LAUNCH(16, DD);      \
        else if (width > 12) MISPMM_TILE_LAUNCH(14, DD); \
        else if (width > 8) MISPMM_TILE_LAUNCH(12, DD);  \
        else MISPMM_TILE_LAUNCH(8, DD);                  \
    } while (0)
    if (dma) MISPMM_TILE_PICK(true);
    else MISPMM_TILE_PICK(false);
#undef MISPMM_TILE_PICK
#undef MISPMM_TILE_LAUNCH
}

}  // namespace mispmm

using namespace mispmm;

extern "C" int mispmm_csr_lds_tile_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t rowNnz, uint32_t numTiles,
                                       const uint32_t *tileRowPtrs, const uint32_t *tileColPtrs, const uint32_t *tileCols, const uint8_t *slots,
                                       const float *vals, const uint32_t *rowMap, const float *B, uint32_t N, uint32_t ldb, float *C,
                                       uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_lds_tile: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0 || numTiles == 0) return MISPMM_OK;
    if (!tileRowPtrs || !tileColPtrs || !tileCols || !slots || !vals || !rowMap) return fail(MISPMM_ERR_INVALID_ARG, "csr_lds_tile: null pointer");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (rowNnz == 0 || rowNnz > 16) return fail(MISPMM_ERR_UNSUPPORTED, "csr_lds_tile: rows of one width of 1..16 entries (got %u)", rowNnz);
    const XcdTiling t = xcd_tiling(N, pick_vec(B, ldb, C, ldc, N), K);
    if (pick_vec(B, ldb, C, ldc, N) != 4 || N % t.q != 0 || (N / t.q) % 64 != 0 || static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull ||
        static_cast<uint64_t>(M) * ldc * 4u > 0x7FFFFFFFull || static_cast<uint64_t>(ceil_div(numTiles, 1u << t.log2p)) >= (1u << 24))
        return fail(MISPMM_ERR_UNSUPPORTED, "csr_lds_tile: 16-byte-aligned operands, column parts of whole 64-column groups, B and C below 2 GiB");
    if (acc_mode == MISPMM_ACC_REFERENCE)
        launch_tiles<AccRefWide>(as_stream(stream), numTiles, tileRowPtrs, tileColPtrs, tileCols, slots, vals, rowMap, rowNnz, K, B, N, ldb, C, M, ldc);
    else
        launch_tiles<AccFast>(as_stream(stream), numTiles, tileRowPtrs, tileColPtrs, tileCols, slots, vals, rowMap, rowNnz, K, B, N, ldb, C, M, ldc);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}
