// csr_hybrid: a long-row CSR in ONE launch by two bodies -- the rows of more than a threshold of entries through the split
// body (csr_split.hpp: a wave per row x 32 columns, the longest rows as 4 chunks on the waves of a workgroup), every
// other row through the rolling row-gather body (row_gather.hpp: a lane group per row, one running sum per element).
//
// Why: on GL7d25 (2798 rows, mean 29 entries, longest 422) the split kernel alone takes 7.0 us REFERENCE /
// 5.5 us FAST, and tools/probe/hybrid_longrows_probe.py shows where: its 437 rows of more than 32 entries ALONE take
// 4.9 / 3.3 us (the critical path of the longest rows), the 2361 short rows alone 3.7 / 3.4 us through the row-gather
// kernel -- the split kernel spends the difference on walking 2361 short rows with a wave per row and 32 columns.  Two
// launches would pay the 1.3 us launch boundary twice (9.4 us one after the other) and two parallel kernel nodes in a
// graph cost 13-23 us per step (cross-queue synchronisation), so the two bodies share one grid: workgroups
// [0, nSplitBlocks) run the split body -- they are dispatched first, so the longest rows start first -- and the
// workgroups behind them the row-gather body on the tail of the SAME span list (rows by decreasing length: the short
// rows are its tail), whose entries (row, start, end, .) name a row's entries and its row of C in one 16-byte read.
// Results: the row-gather body sums a row in entry order (the reference's order), the split body as csr_split.hpp says;
// REFERENCE mode stays bit-exact, FAST within its bound.
#pragma once
#include "csr_split.hpp"
#include "row_gather.hpp"

namespace mispmm {

template <class Acc, int NB, int G>
__global__ __launch_bounds__(256) void csr_hybrid(uint32_t nSplitBlocks, uint32_t nLong, const uint32_t *__restrict__ colIdxs,
                                                  const float *__restrict__ vals, const float *__restrict__ B, uint32_t b_bytes,
                                                  uint32_t N, uint32_t ldb, float *__restrict__ C, uint32_t ldc, uint32_t tile_q,
                                                  uint32_t wgs_per_part, const uint32_t *__restrict__ spans, uint32_t nShort,
                                                  uint32_t rb_chunk, uint32_t tiling, uint32_t cols_per_part, uint32_t c_bytes, uint32_t align_q) {
    if (blockIdx.x < nSplitBlocks) {  // workgroup-uniform
        csr_split_body<Acc, 4, NB>(blockIdx.x, 0u, nLong, nullptr, colIdxs, vals, B, b_bytes, N, ldb, C, ldc, tile_q, wgs_per_part, spans);
    } else {
        // nSplitBlocks is a multiple of 8: the workgroup keeps the XCD (blockIdx.x % 8) a grid of its own would give it.
        // align_q != 0: the row-gather body is handed the grid position that puts THIS XCD on the column part the split body
        // reads here (split: column part = xcd % q; row-gather: column part = xcd' >> log2p, row part = xcd' & (P - 1)),
        // so that an XCD's L2 only ever sees one 128-byte column slice of B from both bodies
        uint32_t bx = blockIdx.x - nSplitBlocks;
        if (align_q != 0) {
            const uint32_t x = bx & 7u, log2p = tiling & 0xFFu;
            bx = (bx & ~7u) | ((x % align_q) << log2p) | (x / align_q);
        }
        row_gather_body<G, 4, Acc, true, SpanRows, 256, 8, true, 16, false, false>(bx, 0u, 0u, nShort, rb_chunk, tiling,
                                                                                 cols_per_part, N, ldb, SpanRows{spans + static_cast<size_t>(nLong) * 4u},
                                                                                 colIdxs, vals, b_bytes, B, C, c_bytes, ldc, nullptr, BatchArg<false>{}
#ifdef MISPMM_STAMPS
                                                                                 , 0u
#endif
        );
    }
}

struct HybridArgs {
    hipStream_t stream;
    uint32_t M, K;  // rows of C, rows of B
    const uint32_t *colIdxs;
    const float *vals, *B;
    uint32_t N, ldb;
    float *C;
    uint32_t ldc;
    const uint32_t *spans;
    uint32_t numSpans, numLong;  // positions [0, numLong) of the span list go to the split body
};

// false: this shape has no two-body launch (the caller takes the split kernel).  Needs 16-byte vectors and B, C below 2 GiB.
template <class Acc>
inline bool launch_hybrid(const HybridArgs &a) {
    if (a.numLong == 0 || a.numLong >= a.numSpans || a.numLong % 4u != 0) return false;
    const SplitTiling st = split_tiling(a.numLong, a.N, 4, true);
    if (st.grid_y != 1) return false;  // more than 256 columns
    // the row-gather body on the split body's XCD grid (P x Q, column parts of 32 columns, the same column part per XCD):
    // with its own grid (64-column parts at N = 128) the two bodies pull different slices of B through every L2 -- on GL7d25
    // 5.4 + 2.7 MB against 4 MiB -- and the launch took as long as the two bodies one after the other (FAST: 5.67 us against
    // 5.52 for the split kernel alone; aligned: see profiles/r3/hybrid_longrows.log).  MISPMM_HYBRID_ALIGN=0 restores it.
    static const bool align = knob_int("MISPMM_HYBRID_ALIGN", 1) != 0;
    XcdTiling gt = xcd_tiling(a.N, 4, a.K);
    uint32_t align_q = 0;
    if (align && a.N % st.q == 0 && a.N / st.q == 32u) {
        gt.q = st.q;
        gt.log2p = st.p == 8 ? 3u : st.p == 4 ? 2u : st.p == 2 ? 1u : 0u;
        align_q = st.q;
    }
    const uint32_t cpp = a.N / gt.q;
    const int g = cpp == 64 ? 16 : cpp == 32 ? 8 : 0;  // one lane group of whole vectors per column part
    const uint64_t c_bytes = static_cast<uint64_t>(a.M) * a.ldc * 4u;
    if (g == 0 || !gt.sc1 || c_bytes > 0x7FFFFFFFull) return false;
    const uint32_t nShort = a.numSpans - a.numLong;
    const uint32_t rb = ceil_div(nShort, 256u / static_cast<uint32_t>(g));
    const uint32_t rb_chunk = ceil_div(rb, 1u << gt.log2p);
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
    const dim3 grid(st.grid_x + 8u * rb_chunk);
    note_kernel("csr_hybrid<%s,R8> split %u spans xcd %ux%u + row_gather<G%d,spans> %u rows xcd %ux%u%s", acc_tag<Acc>(), a.numLong, st.p, st.q, g,
                nShort, 1u << gt.log2p, gt.q, align_q ? " aligned" : "");
#define MISPMM_HYBRID_ARGS st.grid_x, a.numLong, a.colIdxs, a.vals, a.B, b_bytes, a.N, a.ldb, a.C, a.ldc, st.q, st.rows_per_part, a.spans, nShort, \
                           rb_chunk, gt.log2p, cpp, static_cast<uint32_t>(c_bytes), align_q
    if (g == 16) hipLaunchKernelGGL((csr_hybrid<Acc, 2, 16>), grid, dim3(256), 0, a.stream, MISPMM_HYBRID_ARGS);
    else hipLaunchKernelGGL((csr_hybrid<Acc, 2, 8>), grid, dim3(256), 0, a.stream, MISPMM_HYBRID_ARGS);
#undef MISPMM_HYBRID_ARGS
    return true;
}

}  // namespace mispmm
