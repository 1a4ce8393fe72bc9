// CSR x dense SpMM for gfx950.  C[M x N] = A * B, row-major B/C, fp32 values, u32 indices.
//
// The path this replaces: /root/reference/src/spmm/csr/spmm_csr_k{1,2,3,4}.cu (kernels and
// wrapper bodies).  Nothing here is derived from those kernels: the work split is by 64-lane
// wavefronts that own whole C rows (or 2^k rows when N is narrow), each lane owning VEC
// contiguous output columns so every B-row read is one coalesced 256 B..1 KiB segment; a row's
// products are summed in CSR storage order inside one lane, so there is no cross-lane reduction,
// no atomics, and the result can reproduce the reference CPU engine's rounding bit for bit
// (AccRefWide).  Kernels differ only in how the row's (col, val) pairs reach the lanes:
//   k1  vector loads + lane shuffle broadcast (any row length, rows share a wave when N <= 128)
//   k2  a workgroup's contiguous (col, val) range staged through LDS, read back as broadcasts
//   k3  one wave per row: one coalesced vector load per 64 (col, val) pairs, v_readlane moves each
//       pair to SGPRs, so the B row base is an SGPR pair and each B read a saddr-form global_load
//   k4  k3 with two rows interleaved per wave (half the waves, twice the loads in flight)
//   k5  (default) k1's row groups in 128-thread workgroups laid over the XCDs as a (row part x column
//       part) grid, one 16-read batch per short row, non-temporal C stores: row_gather.hpp, shared
//       with ELL and COO
// Roofline: HBM (arithmetic intensity ~1.3 flop/B); algorithmic bytes per launch =
//   nnz*8 + (M+1)*4 + K*N*4 + M*N*4   (SURVEY.md section 8(d)).
#include "row_gather.hpp"
#include "row_stream.hpp"
#include "csr_split.hpp"
#include "csr_hybrid.hpp"

namespace mispmm {

// Every kernel follows one rule learnt from the ISA: the B reads of a batch are issued back to
// back with no branch between them (unused slots are dropped by the buffer range check, see
// spmm_common.hpp), a sched_barrier keeps hipcc from interleaving consumption into the batch, and
// only the arithmetic is predicated.

// ---------------------------------------------------------------------------------- wide fallback
// 64-bit addressing, for a B that spans 2 GiB or more (buffer offsets are 32-bit, bit 31 = drop).  Same work
// split as k1; the unused slots of a batch re-read the row's last entry instead of being dropped.
template <int G, int VEC, class Acc>
__global__ __launch_bounds__(256) void csr_wide(uint32_t M, const uint32_t *__restrict__ rowPtrs,
                                                const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                                const float *__restrict__ B, uint32_t N, uint32_t ldb,
                                                float *__restrict__ C, uint32_t ldc, uint32_t xcd_chunk) {
    constexpr int GROUPS = 256 / G;
    constexpr int U = (VEC == 4) ? 4 : 8;
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t lane = threadIdx.x % G;
    const uint32_t row = xcd_block(blockIdx.x, xcd_chunk) * GROUPS + threadIdx.x / G;
    const uint32_t col0 = blockIdx.y * (G * VEC) + lane * VEC;
    const bool row_ok = row < M;
    const bool col_ok = col0 < N;
    uint32_t start = 0, end = 0;
    if (row_ok) {
        start = rowPtrs[row];
        end = rowPtrs[row + 1];
    }
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    const float *bcol = B + (col_ok ? col0 : 0);  // idle lanes re-read column 0: valid, never stored

    for (uint32_t base = start; base < end; base += G) {
        const uint32_t cnt = min(static_cast<uint32_t>(G), end - base);
        const uint32_t mine = base + min(lane, cnt - 1);
        const uint32_t my_col = colIdxs[mine];
        const float my_val = vals[mine];
        for (uint32_t j = 0; j < cnt; j += U) {
            vec_t bv[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t src = min(j + u, cnt - 1);
                const uint32_t c = __shfl(my_col, src, G);
                av[u] = __shfl(my_val, src, G);
                bv[u] = load_vec<VEC>(bcol + static_cast<size_t>(c) * ldb);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (j + u < cnt) {  // re-read slots hold real B values: they must be skipped, not zeroed
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

// ------------------------------------------------------------------------------------------ k1
// 256/G rows per workgroup, G lanes per row.  Lane i of a row group loads pair i of the current
// G-pair chunk and turns its column into a byte offset (one multiply per pair, spread
// over the lanes); the offsets are then broadcast with lane shuffles.
template <int G, int VEC, class Acc>
__global__ __launch_bounds__(256) void csr_k1(uint32_t M, const uint32_t *__restrict__ rowPtrs,
                                              const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                              const float *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                              float *__restrict__ C, uint32_t ldc, uint32_t xcd_chunk) {
    constexpr int GROUPS = 256 / G;
    constexpr int U = (VEC == 4) ? 8 : 16;  // B loads in flight per lane
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t lane = threadIdx.x % G;
    const uint32_t row = xcd_block(blockIdx.x, xcd_chunk) * GROUPS + threadIdx.x / G;
    const uint32_t col0 = blockIdx.y * (G * VEC) + lane * VEC;
    const bool row_ok = row < M;
    const bool col_ok = col0 < N;
    uint32_t start = 0, end = 0;
    if (row_ok) {
        start = rowPtrs[row];
        end = rowPtrs[row + 1];
    }
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;  // lanes past N never fetch

    for (uint32_t base = start; base < end; base += G) {
        const uint32_t cnt = min(static_cast<uint32_t>(G), end - base);
        const uint32_t mine = base + min(lane, cnt - 1);
        const uint32_t my_off = colIdxs[mine] * (ldb * 4u);
        const float my_val = vals[mine];
        for (uint32_t j = 0; j < cnt; j += U) {
            vec_t bv[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t src = (j + u) & (G - 1);
                const uint32_t off = __shfl(my_off, src, G);
                const float a = __shfl(my_val, src, G);
                av[u] = (j + u < cnt) ? a : 0.f;
                bv[u] = buffer_load_vec<VEC>(rsrc, (j + u < cnt) ? off + lane_off : kDropLoad, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {  // dropped slots contribute 0 * 0 (header comment)
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

// ------------------------------------------------------------------------------------------ k5
// k1's row groups with a 2-D XCD tiling and non-temporal C stores: row_gather.hpp (shared with ELL
// and COO).  The 8 XCDs form a P x Q grid (P * Q = 8): XCD x works on row part x % P and column
// part x / P.  What leaves the L2s is  (sum over row parts of distinct B rows) * N*4 + Q * bytes(A):
// fewer, larger row parts share more B rows, at the price of re-reading A once per column part and
// of narrower (N/Q-column) row segments.  P = 8, Q = 1 is k1's traffic.

// ------------------------------------------------------------------------------------------ k2
// One workgroup = 256/G consecutive rows.  Their (col, val) pairs are one contiguous range of the
// CSR arrays: all 256 threads copy it into LDS with coalesced loads (chunks of CHUNK pairs, the
// column already scaled to a byte offset), then each row group walks its own sub-range with
// broadcast LDS reads.
template <int G, int VEC, class Acc>
__global__ __launch_bounds__(256) void csr_k2(uint32_t M, const uint32_t *__restrict__ rowPtrs,
                                              const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                              const float *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                              float *__restrict__ C, uint32_t ldc, uint32_t xcd_chunk) {
    constexpr int GROUPS = 256 / G;
    constexpr int CHUNK = 1024;
    constexpr int U = (VEC == 4) ? 8 : 16;
    using vec_t = typename VecOf<VEC>::type;
    __shared__ uint2 stage[CHUNK];  // {B byte offset of the column, val bits}

    const uint32_t lane = threadIdx.x % G;
    const uint32_t row0 = xcd_block(blockIdx.x, xcd_chunk) * GROUPS;
    if (row0 >= M) return;  // padding block of the XCD-rounded grid (workgroup-uniform)
    const uint32_t row = row0 + threadIdx.x / G;
    const uint32_t col0 = blockIdx.y * (G * VEC) + lane * VEC;
    const bool row_ok = row < M;
    const bool col_ok = col0 < N;
    const uint32_t row_last = min(row0 + GROUPS, M);
    const uint32_t blk_start = rowPtrs[row0];  // row0 < M by grid construction
    const uint32_t blk_end = rowPtrs[row_last];
    uint32_t start = 0, end = 0;
    if (row_ok) {
        start = rowPtrs[row];
        end = rowPtrs[row + 1];
    }
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;

    for (uint32_t lo = blk_start; lo < blk_end; lo += CHUNK) {
        const uint32_t hi = min(lo + CHUNK, blk_end);
        if (lo != blk_start) __syncthreads();  // previous chunk fully consumed
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
            stage[i - lo] = make_uint2(colIdxs[i] * (ldb * 4u), __float_as_uint(vals[i]));
        }
        __syncthreads();
        const uint32_t s = max(start, lo), e = min(end, hi);
        for (uint32_t j = s; j < e; j += U) {
            vec_t bv[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint2 cv = stage[(j + u - lo) & (CHUNK - 1)];
                av[u] = (j + u < e) ? __uint_as_float(cv.y) : 0.f;
                bv[u] = buffer_load_vec<VEC>(rsrc, (j + u < e) ? cv.x + lane_off : kDropLoad, 0);
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

// ------------------------------------------------------------------------------------- k3 / k4
// Wave-per-row.  The wave's row is provably uniform (readfirstlane), so rowPtrs come by s_load.
// A chunk of up to 64 (col, val) pairs is fetched by ONE coalesced vector load per array (lane i
// holds pair i; the next chunk is prefetched while this one is consumed).  v_readlane moves each
// column to an SGPR, scaled to the byte offset of that B row, which rides in the buffer load's
// scalar offset: `buffer_load_dwordxN v, v_lane_off, s[rsrc], s_row_off offen`.  (gfx950 range-checks
// voffset + soffset against the descriptor, so the descriptor spans all of B.)  Lanes past N and
// the unused slots of the last batch carry kDropLoad and fetch nothing.  B reads go out 16 per lane per row; ROWS rows are interleaved per
// wave (k4 = 2) to halve the wave count and double the loads in flight.
// One batch: U B-row reads per row issued back to back, then U multiply-adds per row in storage
// order.  Slots at or past the row's count are dropped loads (zeros) times a zeroed coefficient:
// 0 * 0 = +0 added to an accumulator that started at +0 and therefore is never -0, an exact no-op
// in both accumulate modes.
template <int VEC, int ROWS, int U, class Acc>
__device__ __forceinline__ void wave_batch(rsrc_t rsrc, uint32_t lane_off, uint32_t ldb4, const uint32_t (&cur_col)[ROWS],
                                           const float (&cur_val)[ROWS], const uint32_t (&cnt)[ROWS], uint32_t j,
                                           typename Acc::T (&acc)[ROWS][VEC]) {
    typename VecOf<VEC>::type bv[ROWS][U];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const uint32_t c = __builtin_amdgcn_readlane(cur_col[r], (j + t) & 63);
            bv[r][t] = buffer_load_vec<VEC>(rsrc, (j + t < cnt[r]) ? lane_off : kDropLoad, c * ldb4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
        for (int t = 0; t < U; ++t) {
            const float a0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cur_val[r]), (j + t) & 63));
            const float a = (j + t < cnt[r]) ? a0 : 0.f;  // scalar select, wave-uniform
#pragma unroll
            for (int v = 0; v < VEC; ++v) Acc::mac(acc[r][v], a, vec_get<VEC>(bv[r][t], v));
        }
    }
}

template <int VEC, int ROWS, class Acc>
__global__ __launch_bounds__(256) void csr_k3(uint32_t M, const uint32_t *__restrict__ rowPtrs,
                                              const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                              const float *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                              float *__restrict__ C, uint32_t ldc, uint32_t xcd_chunk) {
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t row0 = (xcd_block(blockIdx.x, xcd_chunk) * 4 + wave) * ROWS;
    if (row0 >= M) return;  // wave-uniform exit
    const uint32_t col0 = blockIdx.y * (64 * VEC) + lane * VEC;
    const bool col_ok = col0 < N;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;  // lanes past N never fetch
    const uint32_t ldb4 = ldb * 4u;

    uint32_t pos[ROWS], end[ROWS];
    uint32_t cur_col[ROWS];
    float cur_val[ROWS];
    typename Acc::T acc[ROWS][VEC];
    bool more = false;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const bool ok = row0 + r < M;
        pos[r] = ok ? rowPtrs[row0 + r] : 0;
        end[r] = ok ? rowPtrs[row0 + r + 1] : 0;
        more |= pos[r] < end[r];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[r][v] = 0;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        cur_col[r] = 0;
        cur_val[r] = 0.f;
        if (pos[r] < end[r]) {  // wave-uniform; the clamp keeps every lane's index inside the row
            const uint32_t i = min(pos[r] + lane, end[r] - 1);
            cur_col[r] = colIdxs[i];
            cur_val[r] = vals[i];
        }
    }
    while (more) {
        uint32_t nxt_col[ROWS];
        float nxt_val[ROWS];
        uint32_t cnt[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {  // prefetch the next chunk of (col, val)
            nxt_col[r] = 0;
            nxt_val[r] = 0.f;
            if (pos[r] + 64 < end[r]) {
                const uint32_t i = min(pos[r] + 64 + lane, end[r] - 1);
                nxt_col[r] = colIdxs[i];
                nxt_val[r] = vals[i];
            }
            cnt[r] = min(64u, end[r] - pos[r]);
        }
        uint32_t maxcnt = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) maxcnt = max(maxcnt, cnt[r]);
        for (uint32_t j = 0; j < maxcnt;) {  // wave-uniform: whole batches sit under the branches
            const uint32_t rem = maxcnt - j;
            if (rem > 8) {
                wave_batch<VEC, ROWS, 16, Acc>(rsrc, lane_off, ldb4, cur_col, cur_val, cnt, j, acc);
                j += 16;
            } else if (rem > 4) {
                wave_batch<VEC, ROWS, 8, Acc>(rsrc, lane_off, ldb4, cur_col, cur_val, cnt, j, acc);
                j += 8;
            } else {
                wave_batch<VEC, ROWS, 4, Acc>(rsrc, lane_off, ldb4, cur_col, cur_val, cnt, j, acc);
                j += 4;
            }
        }
        more = false;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            pos[r] = min(pos[r] + 64, end[r]);
            cur_col[r] = nxt_col[r];
            cur_val[r] = nxt_val[r];
            more |= pos[r] < end[r];
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (row0 + r < M && col_ok) {
            vec_t out;
#pragma unroll
            for (int v = 0; v < VEC; ++v) vec_set<VEC>(out, v, Acc::finish(acc[r][v]));
            store_vec<VEC>(C + static_cast<size_t>(row0 + r) * ldc + col0, out);
        }
    }
}

// ------------------------------------------------------------------------------- csr_wave_deep
// Long rows.  A row's sum is sequential by contract, so a row of L entries takes (L / reads in flight) x memory
// latency however many waves the grid has; the lane-group kernels keep 8-16 reads in flight per row.  Here one WAVE
// owns one row x (64 * VEC) columns and keeps W = 32 B-row reads in flight, refilled block by block (8 slots).  The
// row's entries are staged once in a wave-private LDS strip as (byte offset of the B row, coefficient) pairs -- padding
// to a multiple of 8 is (kDropLoad, 0), an exact no-op -- and every slot reads its pair back as an LDS broadcast, so
// the body is a plain loop over blocks for any row length.  Chosen by the dispatcher when nnz >= 24 M.  GL7d25 (mean 29,
// longest 422 entries), REFERENCE / FAST us: lane-group kernel 29.0 / 25.6, the same with 16 reads in flight 23.0 / 18.2,
// this kernel 19.2 / 16.8.  Measured and no better: 64 reads in flight (24.3 / 21.1), one wave per 64 columns (19.0 /
// 17.5), the pairs of the next block read from LDS one step early (21.7 / 17.2), other XCD tilings (23-35).
template <int VEC, class Acc, int W = 32, int PHASE = 512>
__global__ __launch_bounds__(256) void csr_wave_deep(uint32_t M, const uint32_t *__restrict__ rowPtrs,
                                                     const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                                     const float *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                                     float *__restrict__ C, uint32_t ldc, uint32_t xcd_chunk) {
    using vec_t = typename VecOf<VEC>::type;
    using u2 = uint32_t __attribute__((ext_vector_type(2)));
    constexpr int NBLK = W / 8;
    __shared__ u2 strip[4][PHASE];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t row = xcd_block(blockIdx.x, xcd_chunk) * 4 + wave;
    if (row >= M) return;  // wave-uniform exit; no workgroup barrier below
    const uint32_t col0 = blockIdx.y * (64 * VEC) + lane * VEC;
    const bool col_ok = col0 < N;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;
    const uint32_t ldb4 = ldb * 4u;
    const uint32_t start = rowPtrs[row], end = rowPtrs[row + 1];
    u2 *e = strip[wave];
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;

    for (uint32_t base = start; base < end; base += PHASE) {
        const uint32_t n = min(static_cast<uint32_t>(PHASE), end - base);
        // padded to whole RINGS of NBLK blocks (W slots): every pass of the loop below then consumes and refills exactly
        // W slots with no branch in between, which is what lets the compiler count its vmcnt waits -- with a per-block
        // `if (blk < nblk)` it fell back to vmcnt(7..0) before every block, i.e. a full drain of the window per 8
        // entries (GL7d25: 19.7 us; the padding slots are dropped loads with a zero coefficient: exact no-ops)
        const uint32_t nblk = ((n + W - 1u) / W) * NBLK;
        for (uint32_t i = lane; i < nblk * 8u; i += 64) {
            u2 pair{kDropLoad, 0u};
            if (i < n) {
                pair[0] = colIdxs[base + i] * ldb4;
                pair[1] = __float_as_uint(vals[base + i]);
            }
            e[i] = pair;
        }
        // the strip is private to this wave and a wave's LDS operations complete in order; the wait keeps the
        // compiler from moving the reads below ahead of the writes above
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        vec_t bv[W];
        float av[W];
        auto issue_block = [&](uint32_t blk, auto ring_tag) {
            constexpr int R = decltype(ring_tag)::value;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const u2 pair = e[blk * 8u + t];  // same address in every lane: an LDS broadcast
                av[R * 8 + t] = __uint_as_float(pair[1]);
                bv[R * 8 + t] = buffer_load_vec<VEC>(rsrc, pair[0] + lane_off, 0);
            }
        };
        auto consume_block = [&](auto ring_tag) {
            constexpr int R = decltype(ring_tag)::value;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if constexpr (VEC == 4 && std::is_same_v<Acc, AccRefWide>) {
                    Acc::mac4(acc, av[R * 8 + t], vec_get<VEC>(bv[R * 8 + t], 0), vec_get<VEC>(bv[R * 8 + t], 1),
                              vec_get<VEC>(bv[R * 8 + t], 2), vec_get<VEC>(bv[R * 8 + t], 3));
                } else if constexpr (VEC == 2 && std::is_same_v<Acc, AccRefWide>) {
                    Acc::mac2(acc, av[R * 8 + t], vec_get<VEC>(bv[R * 8 + t], 0), vec_get<VEC>(bv[R * 8 + t], 1));
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) Acc::mac(acc[v], av[R * 8 + t], vec_get<VEC>(bv[R * 8 + t], v));
                }
            }
        };
        auto pin = [&] {
            // a refill reuses the registers just consumed: keep it behind the sums (see row_gather.hpp)
            if constexpr (VEC == 4) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
            else if constexpr (VEC == 2) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]) : : "memory");
            else asm volatile("" : "+v"(acc[0]) : : "memory");
        };
        static_for<0, NBLK>([&](auto r) { issue_block(decltype(r)::value, r); });  // nblk >= NBLK: the first ring
        uint32_t b0 = 0;
        for (; b0 + NBLK < nblk; b0 += NBLK) {  // steady state: W reads in flight, each block refilled as it is consumed
            static_for<0, NBLK>([&](auto r) {
                consume_block(r);
                pin();
                __builtin_amdgcn_sched_barrier(0);
                issue_block(b0 + NBLK + decltype(r)::value, r);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        static_for<0, NBLK>([&](auto r) { consume_block(r); });  // the last ring drains
        // the next phase overwrites the strip: every read of this phase has been issued and, LDS being in order per
        // wave, completes before those writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (col_ok) {
        vec_t out;
#pragma unroll
        for (int v = 0; v < VEC; ++v) vec_set<VEC>(out, v, Acc::finish(acc[v]));
        store_vec<VEC>(C + static_cast<size_t>(row) * ldc + col0, out);
    }
}

// ------------------------------------------------------------------------------------ dispatch
struct CsrArgs {
    hipStream_t stream;
    uint32_t M, K;
    const uint32_t *rowPtrs, *colIdxs;
    const float *vals, *B;
    uint32_t N, ldb;
    float *C;
    uint32_t ldc;
    uint32_t nnz = 0;
    const uint32_t *spans = nullptr;  // kernel 6 only: the span list (rows longest first, long rows as 4 chunks)
    uint32_t numSpans = 0;
};

template <int G, int VEC, class Acc>
static void launch_grouped(const CsrArgs &a, int kernel, bool wide) {
    const XcdGrid xg = xcd_grid(ceil_div(a.M, 256 / G));
    dim3 grid(xg.grid, ceil_div(a.N, G * VEC));
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
    note_kernel("%s<G%d,V%d,%s>", wide ? "csr_wide" : kernel == 1 ? "csr_k1" : "csr_k2", G, VEC, acc_tag<Acc>());
    if (wide)
        hipLaunchKernelGGL((csr_wide<G, VEC, Acc>), grid, dim3(256), 0, a.stream, a.M, a.rowPtrs, a.colIdxs, a.vals,
                           a.B, a.N, a.ldb, a.C, a.ldc, xg.chunk);
    else if (kernel == 1)
        hipLaunchKernelGGL((csr_k1<G, VEC, Acc>), grid, dim3(256), 0, a.stream, a.M, a.rowPtrs, a.colIdxs, a.vals, a.B,
                           b_bytes, a.N, a.ldb, a.C, a.ldc, xg.chunk);
    else
        hipLaunchKernelGGL((csr_k2<G, VEC, Acc>), grid, dim3(256), 0, a.stream, a.M, a.rowPtrs, a.colIdxs, a.vals, a.B,
                           b_bytes, a.N, a.ldb, a.C, a.ldc, xg.chunk);
}

template <int VEC, class Acc>
static void launch_grouped_g(const CsrArgs &a, int kernel, int g, bool wide) {
    switch (g) {
        case 8: launch_grouped<8, VEC, Acc>(a, kernel, wide); break;
        case 16: launch_grouped<16, VEC, Acc>(a, kernel, wide); break;
        case 32: launch_grouped<32, VEC, Acc>(a, kernel, wide); break;
        default: launch_grouped<64, VEC, Acc>(a, kernel, wide); break;
    }
}

template <int VEC, int ROWS, class Acc>
static void launch_wave(const CsrArgs &a) {
    const XcdGrid xg = xcd_grid(ceil_div(a.M, 4 * ROWS));
    dim3 grid(xg.grid, ceil_div(a.N, 64 * VEC));
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
    note_kernel("csr_k3<V%d,R%d,%s>", VEC, ROWS, acc_tag<Acc>());
    hipLaunchKernelGGL((csr_k3<VEC, ROWS, Acc>), grid, dim3(256), 0, a.stream, a.M, a.rowPtrs, a.colIdxs, a.vals, a.B,
                       b_bytes, a.N, a.ldb, a.C, a.ldc, xg.chunk);
}

template <int VEC, class Acc>
static void launch_wave_deep(const CsrArgs &a) {
    const XcdGrid xg = xcd_grid(ceil_div(a.M, 4u));
    dim3 grid(xg.grid, ceil_div(a.N, 64 * VEC));
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
    note_kernel("csr_wave_deep<V%d,%s>", VEC, acc_tag<Acc>());
    hipLaunchKernelGGL((csr_wave_deep<VEC, Acc>), grid, dim3(256), 0, a.stream, a.M, a.rowPtrs, a.colIdxs, a.vals, a.B,
                       b_bytes, a.N, a.ldb, a.C, a.ldc, xg.chunk);
}

// the split kernel (csr_split.hpp) on the arguments of a CSR call
template <class Acc>
static void launch_split(const CsrArgs &a) {
    SplitArgs sa{a.stream, a.M, a.K, a.rowPtrs, a.colIdxs, a.vals, a.B, a.N, a.ldb, a.C, a.ldc};
    sa.spans = a.spans;
    sa.numSpans = a.numSpans;
    mispmm::launch_split<Acc>(sa);
}

template <class Acc>
static void launch_csr(const CsrArgs &a, int kernel, int vec) {
    // buffer offsets are 32-bit and bit 31 marks a dropped load: a B of 2 GiB or more takes the
    // 64-bit-address kernel
    const bool wide = static_cast<uint64_t>(a.K) * a.ldb * 4u > 0x7FFFFFFFull;
    // kernel 5 on a matrix whose MEAN row is long (nnz >= 24 M): the deep wave-per-row kernel.  MISPMM_LONGROWS=0/1
    // forces the choice (measurement aid; results do not depend on it).
    static const int long_env = knob_int("MISPMM_LONGROWS", -1);
    const bool long_rows = long_env >= 0 ? long_env != 0 : (a.M != 0 && a.nnz / a.M >= 24);
    // kernel 6, and kernel 5 on long rows: the row split over the lane groups of a wave (csr_split.hpp); needs 16-byte
    // B rows.  GL7d25 (mean 29, longest 422 entries), rows in order, us at N = 64 / 128 / 256 / 512:
    //   REFERENCE, no element needs the ordered re-sum    7.6 /  9.9 / 13.7 / 26.3
    //   REFERENCE, every wave re-sums (B spread over 2^60) 15.1 / 18.6 / 24.7 / 42.6
    //   REFERENCE before (deep wave / lane group)        13.9 / 17.4 / 27.5 / 30.8
    //   FAST 5.7 / 7.0 / 10.0 / 21.4 (before: 15.2 at N = 128)
    // so REFERENCE mode keeps the lane-group kernel from N = 384 on, where the bad case costs more than the good one
    // gains.  (With the span list of mispmm_csr_split_f32: 4.7 / 6.9 / 11.2 / 24.2, worst case 10.3 / 10.6 / 18.0 / 37.0,
    // FAST 3.9 / 5.5 / 8.8 / 20.0.)  MISPMM_SPLIT=0 keeps kernel 5 off the split kernel altogether (measurement aid).
    static const int split_env = knob_int("MISPMM_SPLIT", 1);
    const bool split_pays = std::is_same_v<Acc, AccFast> || a.N < 384;
    if (!wide && vec == 4 && (kernel == 6 || (kernel == 5 && long_rows && split_env != 0 && split_pays))) {
        launch_split<Acc>(a);
        return;
    }
    if (kernel == 6) kernel = 5;  // shapes the split kernel does not take
    // long rows without 16-byte B rows: the deep wave-per-row kernel up to 256 output columns; from 384 on the lane-group
    // kernel with 16 reads in flight is faster again (GL7d25, REFERENCE us, deep wave / lane group: N = 128 17.2 / 22.9,
    // 256 26.2 / 29.1, 384 39.4 / 37.4, 512 44.5 / 30.2)
    if (kernel == 5 && !wide && long_rows && (long_env == 1 || a.N < 384)) {
        int v = vec;
        while (v > 1 && 64u * (v / 2) >= a.N) v /= 2;
        static const int vec_env = knob_int("MISPMM_LONGROWS_VEC", 0);
        if (vec_env > 0) v = min(v, vec_env);
        if (v == 4) launch_wave_deep<4, Acc>(a);
        else if (v == 2) launch_wave_deep<2, Acc>(a);
        else launch_wave_deep<1, Acc>(a);
        return;
    }
    if (kernel == 5 && !wide) {
        RowGatherArgs ga{a.stream, a.M, a.K, a.colIdxs, a.vals, a.B, a.N, a.ldb, a.C, a.ldc, a.M ? a.nnz / a.M : 0u};
        ga.row_len_guess = uniform_guess(a.M, a.nnz);
        launch_row_gather_auto<Acc>(ga, CsrRows{a.rowPtrs}, vec);
        return;
    }
    if (kernel == 1 || kernel == 2 || kernel == 5 || wide) {
        const int g = pick_group(a.N, vec);
        if (vec == 4) launch_grouped_g<4, Acc>(a, kernel, g, wide);
        else if (vec == 2) launch_grouped_g<2, Acc>(a, kernel, g, wide);
        else launch_grouped_g<1, Acc>(a, kernel, g, wide);
        return;
    }
    // wave per row: narrow the vector so a 64-lane wave is not mostly idle at small N
    while (vec > 1 && 64u * (vec / 2) >= a.N) vec /= 2;
    if (kernel == 3) {
        if (vec == 4) launch_wave<4, 1, Acc>(a);
        else if (vec == 2) launch_wave<2, 1, Acc>(a);
        else launch_wave<1, 1, Acc>(a);
    } else {
        if (vec == 4) launch_wave<4, 2, Acc>(a);
        else if (vec == 2) launch_wave<2, 2, Acc>(a);
        else launch_wave<1, 2, Acc>(a);
    }
}

}  // namespace mispmm

using namespace mispmm;

#ifdef MISPMM_STAMPS
// diagnostic build only: where the row-gather waves of THIS translation unit leave their stamps (8 x uint64 per wave)
extern "C" int mispmm_debug_set_stamps(void *device_buffer) {
    MISPMM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mispmm_stamp_buf), &device_buffer, sizeof(device_buffer)));
    return MISPMM_OK;
}
#endif

#ifdef MISPMM_TUNING
// measurement build only (libmispmm_tune.so): how often csr_split had to leave the split sum since the last call --
// out[0] waves that summed their row again in entry order (out[1] unused).  Synchronises the device.
extern "C" int mispmm_debug_split_stats(unsigned long long out[2]) {
    const unsigned long long zero[2] = {0, 0};
    MISPMM_HIP_TRY(hipDeviceSynchronize());
    MISPMM_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(mispmm_split_stats), sizeof(zero)));
    MISPMM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mispmm_split_stats), zero, sizeof(zero)));
    return MISPMM_OK;
}
#endif

extern "C" int mispmm_csr_uniform_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t rowNnz,
                                      const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb,
                                      float *C, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_uniform: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (rowNnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "csr_uniform: colIdxs or vals is null");
    if (static_cast<uint64_t>(M) * rowNnz > 0xFFFFFFFFull) return fail(MISPMM_ERR_INVALID_ARG, "csr_uniform: more than 2^32 entries");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull)
        return fail(MISPMM_ERR_UNSUPPORTED, "csr_uniform: B of 2 GiB or more: use mispmm_csr_f32");
    const RowGatherArgs ga{as_stream(stream), M, K, colIdxs, vals, B, N, ldb, C, ldc};
    const int vec = pick_vec(B, ldb, C, ldc, N);
    // more than one round of waves: the persistent row-walking launch (row_stream.hpp)
    if (try_row_stream(ga, rowNnz, false, acc_mode == MISPMM_ACC_REFERENCE ? 0 : 2, vec, -1)) {
        MISPMM_LAUNCH_CHECK();
        return MISPMM_OK;
    }
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_row_gather_auto<AccRefWide>(ga, UniformRows{rowNnz}, vec);
    else launch_row_gather_auto<AccFast>(ga, UniformRows{rowNnz}, vec);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

extern "C" int mispmm_csr_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                              const uint32_t *colIdxs, const float *vals, const float *B, uint32_t N, uint32_t ldb,
                              float *C, uint32_t ldc, int kernel, int acc_mode) {
    if (kernel < 0 || kernel > MISPMM_CSR_NUM_KERNELS) return fail(MISPMM_ERR_INVALID_ARG, "csr: unknown kernel id %d", kernel);
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "csr: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!rowPtrs) return fail(MISPMM_ERR_INVALID_ARG, "csr: rowPtrs is null");
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "csr: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (kernel == MISPMM_KERNEL_AUTO) kernel = 5;
    const CsrArgs a{as_stream(stream), M, K, rowPtrs, colIdxs, vals, B, N, ldb, C, ldc, nnz};
    const int vec = pick_vec(B, ldb, C, ldc, N);
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_csr<AccRefWide>(a, kernel, vec);
    else launch_csr<AccFast>(a, kernel, vec);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

// Kernel 6 walking a span list (mispmm_csr_spans_by_length_host, uploaded by the caller): rows longest first, long rows
// as 4 chunks on the 4 waves of one workgroup.
extern "C" int mispmm_csr_split_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                                    const uint32_t *colIdxs, const float *vals, const uint32_t *spans, uint32_t numSpans,
                                    const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_split: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!rowPtrs && !spans) return fail(MISPMM_ERR_INVALID_ARG, "csr_split: rowPtrs and spans are both null");
    if (!aligned16(spans)) return fail(MISPMM_ERR_INVALID_ARG, "csr_split: spans must be 16-byte aligned");
    if (spans && (numSpans < M || (numSpans - M) % 3u != 0))
        return fail(MISPMM_ERR_INVALID_ARG, "csr_split: %u spans cannot describe %u rows (M + 3 per shared row)", numSpans, M);
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "csr_split: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull) return fail(MISPMM_ERR_UNSUPPORTED, "csr_split: B of 2 GiB or more: use mispmm_csr_f32");
    if (pick_vec(B, ldb, C, ldc, N) != 4)
        return fail(MISPMM_ERR_UNSUPPORTED, "csr_split: B and C rows must be 16-byte vectors (N, ldb, ldc multiples of 4, aligned): use mispmm_csr_f32");
    CsrArgs a{as_stream(stream), M, K, rowPtrs, colIdxs, vals, B, N, ldb, C, ldc, nnz};
    a.spans = spans;
    a.numSpans = numSpans;
    if (acc_mode == MISPMM_ACC_REFERENCE) launch_split<AccRefWide>(a);
    else launch_split<AccFast>(a);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

// The span list in one launch by two bodies (csr_hybrid.hpp): positions [0, numLongSpans) -- the long rows, a multiple of 4
// that covers every chunk group (mispmm_csr_spans_long_count_host) -- through the split body, the others through the
// row-gather body.  MISPMM_ERR_UNSUPPORTED (no message) where the shape has no such launch: take mispmm_csr_split_f32.
extern "C" int mispmm_csr_hybrid_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *colIdxs, const float *vals,
                                     const uint32_t *spans, uint32_t numSpans, uint32_t numLongSpans, const float *B, uint32_t N,
                                     uint32_t ldb, float *C, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_hybrid: unknown accumulate mode %d", acc_mode);
    if (M == 0 || N == 0) return MISPMM_OK;
    if (!spans || !aligned16(spans)) return fail(MISPMM_ERR_INVALID_ARG, "csr_hybrid: spans must be a 16-byte aligned device array");
    if (numSpans < M || (numSpans - M) % 3u != 0)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_hybrid: %u spans cannot describe %u rows (M + 3 per shared row)", numSpans, M);
    if (numLongSpans > numSpans || (numLongSpans % 4u != 0 && numLongSpans != numSpans) || numLongSpans < ((numSpans - M) / 3u) * 4u)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_hybrid: %u long spans: a multiple of 4 that covers the %u chunk groups is needed", numLongSpans,
                    (numSpans - M) / 3u);
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "csr_hybrid: colIdxs or vals is null");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull || pick_vec(B, ldb, C, ldc, N) != 4) return MISPMM_ERR_UNSUPPORTED;
    const HybridArgs a{as_stream(stream), M, K, colIdxs, vals, B, N, ldb, C, ldc, spans, numSpans, numLongSpans};
    const bool taken = acc_mode == MISPMM_ACC_REFERENCE ? launch_hybrid<AccRefWide>(a) : launch_hybrid<AccFast>(a);
    if (!taken) return MISPMM_ERR_UNSUPPORTED;
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

#ifdef MISPMM_STAMPS
// diagnostic build only: where the waves of the split body leave their stamps (8 x uint64 per wave, 4 waves per workgroup)
extern "C" int mispmm_debug_set_stamps_split(void *device_buffer) {
    MISPMM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mispmm_split_stamp_buf), &device_buffer, sizeof(device_buffer)));
    return MISPMM_OK;
}
#endif

// Several dense operands, one launch.  Only the row-gather kernel has a batched form; shapes it does not take
// (a B of 2 GiB or more, rows too long for it, operands that are not 16-byte vectors) go out as one launch each.
extern "C" int mispmm_csr_batch_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                                    const uint32_t *colIdxs, const float *vals, uint32_t uniformRowNnz, uint32_t batch,
                                    const float *const *B_list_host, uint32_t N, uint32_t ldb, float *const *C_list_host,
                                    uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_batch: unknown accumulate mode %d", acc_mode);
    if (batch == 0 || M == 0 || N == 0) return MISPMM_OK;
    if (!B_list_host || !C_list_host) return fail(MISPMM_ERR_INVALID_ARG, "csr_batch: null operand list");
    if (!rowPtrs && uniformRowNnz == 0) return fail(MISPMM_ERR_INVALID_ARG, "csr_batch: rowPtrs is null");
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "csr_batch: colIdxs or vals is null");
    int vec = 4;
    for (uint32_t i = 0; i < batch; ++i) {
        if (int s = check_dense_args(B_list_host[i], N, ldb, C_list_host[i], ldc)) return s;
        vec = std::min(vec, pick_vec(B_list_host[i], ldb, C_list_host[i], ldc, N));
    }
    const bool long_rows = uniformRowNnz == 0 && nnz / M >= 24;
    const bool fits = static_cast<uint64_t>(K) * ldb * 4u <= 0x7FFFFFFFull && static_cast<uint64_t>(M) * ldc * 4u <= 0x7FFFFFFFull;
    const uint32_t cpp = N / xcd_tiling(N, vec, K).q;
    if (vec != 4 || !fits || long_rows || (cpp % 32 != 0)) {  // no batched kernel for this shape: one launch per operand
        for (uint32_t i = 0; i < batch; ++i) {
            const int st = uniformRowNnz ? mispmm_csr_uniform_f32(stream, M, K, uniformRowNnz, colIdxs, vals, B_list_host[i], N, ldb,
                                                                 C_list_host[i], ldc, acc_mode)
                                         : mispmm_csr_f32(stream, M, K, nnz, rowPtrs, colIdxs, vals, B_list_host[i], N, ldb,
                                                          C_list_host[i], ldc, MISPMM_KERNEL_AUTO, acc_mode);
            if (st != MISPMM_OK) return st;
        }
        return MISPMM_OK;
    }
    for (uint32_t first = 0; first < batch; first += kMaxBatch) {
        RowGatherArgs ga{as_stream(stream), M, K, colIdxs, vals, nullptr, N, ldb, nullptr, ldc, M ? nnz / M : 0u};
        ga.batch = std::min(kMaxBatch, batch - first);
        ga.B_list = B_list_host + first;
        ga.C_list = C_list_host + first;
        ga.row_len_guess = uniform_guess(M, nnz);
        if (uniformRowNnz) {
            if (acc_mode == MISPMM_ACC_REFERENCE) launch_row_gather_auto<AccRefWide>(ga, UniformRows{uniformRowNnz}, vec);
            else launch_row_gather_auto<AccFast>(ga, UniformRows{uniformRowNnz}, vec);
        } else {
            if (acc_mode == MISPMM_ACC_REFERENCE) launch_row_gather_auto<AccRefWide>(ga, CsrRows{rowPtrs}, vec);
            else launch_row_gather_auto<AccFast>(ga, CsrRows{rowPtrs}, vec);
        }
        MISPMM_LAUNCH_CHECK();
    }
    return MISPMM_OK;
}

// A CSR stored in a PLAN order (rows permuted once at upload, e.g. by mispmm_csr_cluster_rows_host so that rows which
// read the same B rows run close together in time and on the same XCD): array row i produces row rowMap[i] of C.  Always the
// row-gather kernel (the only one that scatters its rows); one launch per 16 operands where the batched form exists.
extern "C" int mispmm_csr_plan_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                                   const uint32_t *colIdxs, const float *vals, uint32_t uniformRowNnz, const uint32_t *rowMap,
                                   uint32_t batch, const float *const *B_list_host, uint32_t N, uint32_t ldb,
                                   float *const *C_list_host, uint32_t ldc, int acc_mode) {
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "csr_plan: unknown accumulate mode %d", acc_mode);
    if (batch == 0 || M == 0 || N == 0) return MISPMM_OK;
    if (!B_list_host || !C_list_host) return fail(MISPMM_ERR_INVALID_ARG, "csr_plan: null operand list");
    if (!rowPtrs && uniformRowNnz == 0) return fail(MISPMM_ERR_INVALID_ARG, "csr_plan: rowPtrs is null");
    if (nnz != 0 && (!colIdxs || !vals)) return fail(MISPMM_ERR_INVALID_ARG, "csr_plan: colIdxs or vals is null");
    int vec = 4;
    for (uint32_t i = 0; i < batch; ++i) {
        if (int s = check_dense_args(B_list_host[i], N, ldb, C_list_host[i], ldc)) return s;
        vec = std::min(vec, pick_vec(B_list_host[i], ldb, C_list_host[i], ldc, N));
    }
    if (static_cast<uint64_t>(K) * ldb * 4u > 0x7FFFFFFFull)
        return fail(MISPMM_ERR_UNSUPPORTED, "csr_plan: B of 2 GiB or more: multiply from the unpermuted arrays (mispmm_csr_f32)");
    if (rowMap && !row_gather_supports_map(M, K, N, ldc, vec, uniformRowNnz ? uniformRowNnz : nnz / M))
        return fail(MISPMM_ERR_UNSUPPORTED, "csr_plan: the row-mapped kernel takes 16-byte-aligned operands with N a multiple of 32 "
                                            "(per XCD column part), C below 2 GiB and short rows: multiply from the unpermuted arrays");
    const bool batched_ok = vec == 4 && static_cast<uint64_t>(M) * ldc * 4u <= 0x7FFFFFFFull && (N / xcd_tiling(N, vec, K).q) % 32 == 0 &&
                            !(uniformRowNnz == 0 && nnz / M >= 24);
    auto launch = [&](RowGatherArgs &ga) {
        ga.rowMap = rowMap;
        ga.row_len_guess = uniform_guess(M, nnz);
        row_gather_declined() = false;
        if (uniformRowNnz && try_row_stream(ga, uniformRowNnz, false, acc_mode == MISPMM_ACC_REFERENCE ? 0 : 2, vec, -1)) return true;
        if (uniformRowNnz) {
            if (acc_mode == MISPMM_ACC_REFERENCE) launch_row_gather_auto<AccRefWide>(ga, UniformRows{uniformRowNnz}, vec);
            else launch_row_gather_auto<AccFast>(ga, UniformRows{uniformRowNnz}, vec);
        } else {
            if (acc_mode == MISPMM_ACC_REFERENCE) launch_row_gather_auto<AccRefWide>(ga, CsrRows{rowPtrs}, vec);
            else launch_row_gather_auto<AccFast>(ga, CsrRows{rowPtrs}, vec);
        }
        return !row_gather_declined();
    };
    const char *declined = "csr_plan: the row-gather body chosen for this shape has no row-mapped form: multiply from the unpermuted arrays";
    if (batch > 1 && batched_ok) {
        for (uint32_t first = 0; first < batch; first += kMaxBatch) {
            RowGatherArgs ga{as_stream(stream), M, K, colIdxs, vals, nullptr, N, ldb, nullptr, ldc, M ? nnz / M : 0u};
            ga.batch = std::min(kMaxBatch, batch - first);
            ga.B_list = B_list_host + first;
            ga.C_list = C_list_host + first;
            if (!launch(ga)) return fail(MISPMM_ERR_UNSUPPORTED, "%s", declined);
            MISPMM_LAUNCH_CHECK();
        }
        return MISPMM_OK;
    }
    for (uint32_t i = 0; i < batch; ++i) {
        RowGatherArgs ga{as_stream(stream), M, K, colIdxs, vals, B_list_host[i], N, ldb, C_list_host[i], ldc, M ? nnz / M : 0u};
        if (!launch(ga)) return fail(MISPMM_ERR_UNSUPPORTED, "%s", declined);   // before anything was launched for this operand
        MISPMM_LAUNCH_CHECK();
    }
    return MISPMM_OK;
}

// ---- autotune: plan order or storage order, measured once per (matrix, N) -------------------------------------------------
extern "C" int mispmm_autotune_pick(const float *times_us, uint32_t n, float min_gain) {
    if (!times_us || n == 0) return -1;
    auto usable = [](float t) { return t > 0.f && t < 3.0e38f && t == t; };
    int best = usable(times_us[0]) ? 0 : -1;
    for (uint32_t i = 1; i < n; ++i) {
        if (!usable(times_us[i])) continue;
        if (best < 0) best = static_cast<int>(i);
        else if (best == 0 ? times_us[i] < times_us[0] * (1.f - min_gain) : times_us[i] < times_us[best]) best = static_cast<int>(i);
    }
    return best < 0 ? 0 : best;
}

extern "C" int mispmm_csr_autotune_plan_f32(mispmm_stream_t stream, uint32_t M, uint32_t K, uint32_t nnz, const uint32_t *rowPtrs,
                                            const uint32_t *colIdxs, const float *vals, uint32_t uniformRowNnz, const uint32_t *planRowPtrs,
                                            const uint32_t *planColIdxs, const float *planVals, const uint32_t *planRowMap, uint32_t N,
                                            int acc_mode, uint32_t launches, int *use_plan_out, float *times_us_out) {
    if (!use_plan_out) return fail(MISPMM_ERR_INVALID_ARG, "autotune: use_plan_out is null");
    *use_plan_out = 0;
    const float inf = __builtin_huge_valf();
    float times[2] = {inf, inf};
    if (times_us_out) times_us_out[0] = times_us_out[1] = inf;
    if (M == 0 || N == 0 || nnz == 0 || !planRowMap || !planColIdxs || !planVals) return MISPMM_OK;   // nothing to choose from
    if (launches == 0) launches = 48;
    // Timed with plain launches on the CALLER's stream.  (A version that replayed each candidate from a graph on a stream of its
    // own discriminated kernels of a few microseconds better -- launched one by one those run at the host's launch rate -- but
    // left the process slower: every later launch of the K = 512 product took 13.55 instead of 12.98 us, same kernel, same
    // operands, profiles/r4/autotune_side_effect.log.  The plan order only ever wins on products of 10 us and more, where
    // plain launches time true; on shorter ones both candidates read the launch rate, tie, and the default is kept.)
    hipStream_t st = as_stream(stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(MISPMM_ERR_INVALID_ARG, "autotune: the stream is being captured");
    float *B = nullptr, *C = nullptr;
    const size_t b_elems = static_cast<size_t>(K) * N, c_elems = static_cast<size_t>(M) * N;
    if (hipMalloc(&B, b_elems * sizeof(float)) != hipSuccess)
        return fail(MISPMM_ERR_ALLOC, "autotune: no memory for the scratch B (%zu bytes)", b_elems * sizeof(float));
    if (hipMalloc(&C, c_elems * sizeof(float)) != hipSuccess) {
        (void)hipFree(B);
        return fail(MISPMM_ERR_ALLOC, "autotune: no memory for the scratch C (%zu bytes)", c_elems * sizeof(float));
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int status = MISPMM_OK;
    auto cleanup = [&] {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        (void)hipFree(B);
        (void)hipFree(C);
    };
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(B), 0x3F800000, b_elems, st) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess) {
        cleanup();
        return fail(MISPMM_ERR_HIP, "autotune: scratch setup failed");
    }
    const float *bl[1] = {B};
    float *cl[1] = {C};
    auto run = [&](int which) -> int {
        if (which == 1)
            return mispmm_csr_plan_f32(stream, M, K, nnz, planRowPtrs, planColIdxs, planVals, uniformRowNnz, planRowMap, 1, bl, N, N, cl, N, acc_mode);
        if (uniformRowNnz && static_cast<uint64_t>(K) * N * 4u <= 0x7FFFFFFFull)
            return mispmm_csr_uniform_f32(stream, M, K, uniformRowNnz, colIdxs, vals, B, N, N, C, N, acc_mode);
        return mispmm_csr_f32(stream, M, K, nnz, rowPtrs, colIdxs, vals, B, N, N, C, N, MISPMM_KERNEL_AUTO, acc_mode);
    };
    for (int which = 0; which < 2 && status == MISPMM_OK; ++which) {
        int rs = MISPMM_OK;
        for (int i = 0; i < 3 && rs == MISPMM_OK; ++i) rs = run(which);               // warm: code, caches, clocks
        if (rs == MISPMM_ERR_UNSUPPORTED) continue;                                   // this candidate does not take the shape
        if (rs != MISPMM_OK) { status = rs; break; }
        float best = inf;
        for (int round = 0; round < 3; ++round) {
            (void)hipEventRecord(e0, st);
            for (uint32_t i = 0; i < launches && rs == MISPMM_OK; ++i) rs = run(which);
            (void)hipEventRecord(e1, st);
            if (rs != MISPMM_OK || hipEventSynchronize(e1) != hipSuccess) { status = rs != MISPMM_OK ? rs : MISPMM_ERR_HIP; break; }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) best = std::min(best, ms * 1e3f / static_cast<float>(launches));
        }
        times[which] = best;
    }
    (void)hipStreamSynchronize(st);
    cleanup();
    if (status != MISPMM_OK) return status;
    if (times_us_out) {
        times_us_out[0] = times[0];
        times_us_out[1] = times[1];
    }
    *use_plan_out = mispmm_autotune_pick(times, 2, 0.02f) == 1;
    return MISPMM_OK;
}

