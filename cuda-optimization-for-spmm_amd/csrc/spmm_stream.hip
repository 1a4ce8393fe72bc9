// Host side of the persistent row-walking kernel (row_stream.hpp): which shapes have an instance and how many workgroups the
// chip holds at once.  MEASURED AND NOT ADOPTED (profiles/r4/stream_ab.log): equal to the one-row-per-lane-group launch of
// row_gather.hpp where a lane group has one row to walk, 4-8 % slower where it has two (n4c6-b13 x N = 512).  The kernels are
// therefore compiled into the tuning build only (libmispmm_tune.so, MISPMM_STREAM=1 takes them wherever an instance exists:
// tests/_row_stream_cases.py keeps them bit-exact); the production library never selects them and does not carry them.
#include "row_stream.hpp"

#ifndef MISPMM_TUNING
namespace mispmm {
bool try_row_stream(const RowGatherArgs &, uint32_t, bool, int, int, int) { return false; }
}  // namespace mispmm
#else

namespace mispmm {

namespace {

constexpr uint32_t kCUs = 256, kCUsPerXcd = 32;

template <int G, class Acc, bool PADDED, int W, bool MAPPED>
void launch_stream(const RowGatherArgs &a, uint32_t width, const XcdTiling &t) {
    auto kern = row_stream_kernel<G, Acc, PADDED, W, MAPPED>;
    // workgroups of 128 threads one CU holds at once (registers decide; asked once per instance)
    static const int per_cu = [&] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 128, 0) != hipSuccess || n <= 0) n = 2 * MISPMM_STREAM_WAVES;
        static const int cap = knob_int("MISPMM_STREAM_WG_PER_CU", 0);   // measurement aid: fewer workgroups per CU
        return cap > 0 ? std::min(n, cap) : n;
    }();
    constexpr uint32_t GROUPS = 128 / G;
    const uint32_t cols_per_part = a.N / t.q;
    const uint32_t col_groups = ceil_div(cols_per_part, G * 4u);
    const uint32_t rows_per_part = ceil_div(ceil_div(a.M, GROUPS), 1u << t.log2p) * GROUPS;   // whole workgroups per row part
    // per XCD: as many workgroups as its 32 CUs hold at once, shared by the column groups of the part (grid.y)
    uint32_t nwg = std::max(1u, kCUsPerXcd * static_cast<uint32_t>(per_cu) / col_groups);
    // ... but no more than give every lane group the same number of rows: with R row blocks for at most `nwg` workgroups the
    // walk takes ceil(R / nwg) rows per lane group whatever the count, so use the FEWEST workgroups that still take that many
    // (6304 rows on 4096 lane groups: 2 rows for some, 1 for the rest, and the launch as long as 2 rows; on 3152: 2 each)
    const uint32_t row_blocks = ceil_div(rows_per_part, GROUPS);
    nwg = ceil_div(row_blocks, ceil_div(row_blocks, nwg));
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
    const uint32_t c_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.M) * a.ldc * 4u);
    note_kernel("row_stream<G%d,%s,%s,W%d> xcd %ux%u, %u workgroups per XCD x %u%s", G, acc_tag<Acc>(), PADDED ? "ell" : "uniform", W, 1u << t.log2p,
                t.q, nwg, col_groups, MAPPED ? " plan-order" : "");
    hipLaunchKernelGGL(kern, dim3(8u * nwg, col_groups), dim3(128), 0, a.stream, a.colIdxs, a.vals, a.B, b_bytes, width, a.ldb, a.M, rows_per_part,
                       t.log2p | (nwg << 8), cols_per_part, a.N, a.C, c_bytes, a.ldc, a.rowMap);
}

template <int G, class Acc, bool PADDED, bool MAPPED>
bool launch_stream_w(const RowGatherArgs &a, uint32_t width, const XcdTiling &t) {
    if (width > 14) launch_stream<G, Acc, PADDED, 16, MAPPED>(a, width, t);
    else if (width > 12) launch_stream<G, Acc, PADDED, 14, MAPPED>(a, width, t);
    else if (width > 10) launch_stream<G, Acc, PADDED, 12, MAPPED>(a, width, t);
    else launch_stream<G, Acc, PADDED, 10, MAPPED>(a, width, t);
    return true;
}

template <int G>
bool launch_stream_g(const RowGatherArgs &a, uint32_t width, bool padded, int acc_kind, const XcdTiling &t) {
    if (padded) {   // ELL: the reference's fp32 product, fp32 add (spmm_ell.cpp:25), or the fused form
        if (a.rowMap) return false;
        return acc_kind == 2 ? launch_stream_w<G, AccFast, true, false>(a, width, t) : launch_stream_w<G, AccRefF32, true, false>(a, width, t);
    }
    if (acc_kind == 1) return false;
    if (a.rowMap) return acc_kind == 2 ? launch_stream_w<G, AccFast, false, true>(a, width, t) : launch_stream_w<G, AccRefWide, false, true>(a, width, t);
    return acc_kind == 2 ? launch_stream_w<G, AccFast, false, false>(a, width, t) : launch_stream_w<G, AccRefWide, false, false>(a, width, t);
}

}  // namespace

bool try_row_stream(const RowGatherArgs &a, uint32_t width, bool padded, int acc_kind, int vec, int force) {
    static const int knob = knob_int("MISPMM_STREAM", 0);    // 1: take the persistent launch wherever an instance exists
    if (force < 0) force = knob;
    if (force <= 0 || a.batch != 0 || vec != 4 || width < 9 || width > 16 || a.M == 0) return false;
    if (static_cast<uint64_t>(a.K) * a.ldb * 4u > 0x7FFFFFFFull || static_cast<uint64_t>(a.M) * a.ldc * 4u > 0x7FFFFFFFull) return false;
    if (static_cast<uint64_t>(a.M) * width * 4u > 0x7FFFFFFFull) return false;               // (col, val) through buffer descriptors
    if (static_cast<uint64_t>(a.M) + 3ull * kCUs * 160u > 0xFFFFFFFFull) return false;       // row + 2 * stride stays in 32 bits
    const XcdTiling t = xcd_tiling(a.N, vec, a.K);
    if (a.N % t.q != 0) return false;
    const uint32_t cpp = a.N / t.q;
    const int g = cpp % 64 == 0 ? 16 : cpp % 32 == 0 ? 8 : 0;
    if (g == 0) return false;
    return g == 16 ? launch_stream_g<16>(a, width, padded, acc_kind, t) : launch_stream_g<8>(a, width, padded, acc_kind, t);
}

}  // namespace mispmm

#endif  // MISPMM_TUNING
