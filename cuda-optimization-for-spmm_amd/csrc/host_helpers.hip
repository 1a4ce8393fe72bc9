// Host-only entry points of the C ABI (no device work): format conversion and row sharding.
#include <algorithm>
#include <cstring>
#include <vector>

#include "mispmm_internal.hpp"

using namespace mispmm;

extern "C" int mispmm_ell_colmajor_to_rowmajor_host(uint32_t numRows, uint32_t numCols, uint32_t maxColNnz,
                                                    const uint32_t *rowIdxs_host, const float *vals_host,
                                                    uint32_t *width_out, uint32_t *colIdxs_out_host,
                                                    float *vals_out_host) {
    if (!width_out) return fail(MISPMM_ERR_INVALID_ARG, "ell convert: width_out is null");
    const size_t slots = static_cast<size_t>(numCols) * maxColNnz;
    if (slots != 0 && (!rowIdxs_host || !vals_host)) return fail(MISPMM_ERR_INVALID_ARG, "ell convert: null input");
    std::vector<uint32_t> fill(numRows, 0);
    for (size_t s = 0; s < slots; ++s) {
        const uint32_t r = rowIdxs_host[s];
        if (static_cast<int32_t>(r) < 0) continue;  // padding, as `if (row >= 0)` in spmm_ell.cpp:21
        if (r >= numRows) return fail(MISPMM_ERR_INVALID_ARG, "ell convert: row index %u out of range", r);
        ++fill[r];
    }
    uint32_t width = 0;
    for (uint32_t r = 0; r < numRows; ++r) width = std::max(width, fill[r]);
    *width_out = width;
    if (!colIdxs_out_host && !vals_out_host) return MISPMM_OK;  // size query
    if (!colIdxs_out_host || !vals_out_host) return fail(MISPMM_ERR_INVALID_ARG, "ell convert: one output is null");
    const size_t out_slots = static_cast<size_t>(numRows) * width;
    std::fill(colIdxs_out_host, colIdxs_out_host + out_slots, 0xFFFFFFFFu);
    std::fill(vals_out_host, vals_out_host + out_slots, 0.f);
    std::fill(fill.begin(), fill.end(), 0u);
    // column-major walk, column then slot: each row receives its entries in the order the
    // reference's CPU loop adds them
    for (uint32_t c = 0; c < numCols; ++c) {
        for (uint32_t s = 0; s < maxColNnz; ++s) {
            const size_t i = static_cast<size_t>(c) * maxColNnz + s;
            const uint32_t r = rowIdxs_host[i];
            if (static_cast<int32_t>(r) < 0) continue;
            const size_t o = static_cast<size_t>(r) * width + fill[r]++;
            colIdxs_out_host[o] = c;
            vals_out_host[o] = vals_host[i];
        }
    }
    return MISPMM_OK;
}

// The occupied slots of a row-major ELL as a row-pointer list, slot order kept.
extern "C" int mispmm_ell_compact_host(uint32_t M, uint32_t width, const uint32_t *rmColIdxs_host, const float *rmVals_host,
                                       uint32_t *nnz_out, uint32_t *rowPtrs_out_host, uint32_t *colIdxs_out_host,
                                       float *vals_out_host) {
    if (!nnz_out) return fail(MISPMM_ERR_INVALID_ARG, "ell compact: nnz_out is null");
    const size_t slots = static_cast<size_t>(M) * width;
    if (slots != 0 && (!rmColIdxs_host || !rmVals_host)) return fail(MISPMM_ERR_INVALID_ARG, "ell compact: null input");
    uint64_t count = 0;
    for (size_t i = 0; i < slots; ++i) count += rmColIdxs_host[i] != 0xFFFFFFFFu;
    if (count > 0xFFFFFFFFull) return fail(MISPMM_ERR_INVALID_ARG, "ell compact: more than 2^32 entries");
    *nnz_out = static_cast<uint32_t>(count);
    if (!rowPtrs_out_host && !colIdxs_out_host && !vals_out_host) return MISPMM_OK;  // size query
    if (!rowPtrs_out_host || (count != 0 && (!colIdxs_out_host || !vals_out_host)))
        return fail(MISPMM_ERR_INVALID_ARG, "ell compact: null output");
    uint32_t n = 0;
    for (uint32_t r = 0; r < M; ++r) {
        rowPtrs_out_host[r] = n;
        for (uint32_t s = 0; s < width; ++s) {
            const size_t i = static_cast<size_t>(r) * width + s;
            if (rmColIdxs_host[i] == 0xFFFFFFFFu) continue;
            colIdxs_out_host[n] = rmColIdxs_host[i];
            vals_out_host[n] = rmVals_host[i];
            ++n;
        }
    }
    rowPtrs_out_host[M] = n;
    return MISPMM_OK;
}

extern "C" int mispmm_shard_rows_by_nnz_host(uint32_t M, const uint32_t *rowPtrs_host, uint32_t parts,
                                             uint32_t *bounds_out_host) {
    if (parts == 0) return fail(MISPMM_ERR_INVALID_ARG, "shard: parts must be >= 1");
    if (!bounds_out_host || (M != 0 && !rowPtrs_host)) return fail(MISPMM_ERR_INVALID_ARG, "shard: null pointer");
    bounds_out_host[0] = 0;
    const uint64_t total = M ? rowPtrs_host[M] : 0;
    for (uint32_t p = 1; p < parts; ++p) {
        uint32_t b;
        if (total == 0) {
            b = static_cast<uint32_t>(static_cast<uint64_t>(M) * p / parts);  // no non-zeros: equal row counts
        } else {
            // first row boundary whose prefix reaches p/parts of the non-zeros
            const uint32_t target = static_cast<uint32_t>((total * p + parts / 2) / parts);
            b = static_cast<uint32_t>(std::lower_bound(rowPtrs_host, rowPtrs_host + M + 1, target) - rowPtrs_host);
        }
        bounds_out_host[p] = std::max(bounds_out_host[p - 1], std::min(b, M));
    }
    bounds_out_host[parts] = M;
    return MISPMM_OK;
}

// The span list mispmm_csr_split_f32 walks: rows longest first (ties in row order: a counting sort by length, stable).
// A row of more than share_len entries becomes 4 chunk spans (row, start, end, 1) in one aligned group of 4 positions --
// the 4 waves of one workgroup; chunk lengths are multiples of 8 (one step of the kernel) -- and these groups come first;
// the other rows follow as (row, start, end, 0).
extern "C" int mispmm_csr_spans_by_length_host(uint32_t M, const uint32_t *rowPtrs_host, uint32_t share_len, uint32_t *count_out,
                                               uint32_t *spans_out_host) {
    if (!count_out) return fail(MISPMM_ERR_INVALID_ARG, "csr spans: count_out is null");
    *count_out = 0;
    if (M == 0) return MISPMM_OK;
    if (!rowPtrs_host) return fail(MISPMM_ERR_INVALID_ARG, "csr spans: rowPtrs is null");
    if (share_len == 0) share_len = 128;
    uint32_t longest = 0;
    uint64_t shared_rows = 0;
    for (uint32_t r = 0; r < M; ++r) {
        if (rowPtrs_host[r + 1] < rowPtrs_host[r]) return fail(MISPMM_ERR_INVALID_ARG, "csr spans: row pointers decrease at row %u", r);
        const uint32_t len = rowPtrs_host[r + 1] - rowPtrs_host[r];
        longest = std::max(longest, len);
        shared_rows += len > share_len;
    }
    const uint64_t count = static_cast<uint64_t>(M) + 3u * shared_rows;
    if (count > 0xFFFFFFFFull) return fail(MISPMM_ERR_INVALID_ARG, "csr spans: more than 2^32 spans");
    *count_out = static_cast<uint32_t>(count);
    if (!spans_out_host) return MISPMM_OK;  // size query
    // rank of every row in the descending order: first position of every length, then a stable fill
    std::vector<uint32_t> first(static_cast<size_t>(longest) + 2, 0);
    for (uint32_t r = 0; r < M; ++r) ++first[longest - (rowPtrs_host[r + 1] - rowPtrs_host[r]) + 1];
    for (size_t k = 1; k < first.size(); ++k) first[k] += first[k - 1];
    // the shared rows are exactly the first `shared_rows` ranks (they are the longest): rank k < shared_rows -> group k,
    // any other rank k -> position 4 * shared_rows + (k - shared_rows)
    for (uint32_t r = 0; r < M; ++r) {
        const uint32_t s = rowPtrs_host[r], e = rowPtrs_host[r + 1], len = e - s;
        const uint64_t rank = first[longest - len]++;
        if (len > share_len) {
            const uint32_t chunk = ((len + 3u) / 4u + 7u) & ~7u;
            for (uint32_t k = 0; k < 4; ++k) {
                uint32_t *span = spans_out_host + (rank * 4u + k) * 4u;
                span[0] = r;
                span[1] = std::min(e, s + k * chunk);
                span[2] = std::min(e, s + (k + 1u) * chunk);
                span[3] = 1;
            }
        } else {
            uint32_t *span = spans_out_host + (4u * shared_rows + (rank - shared_rows)) * 4u;
            span[0] = r;
            span[1] = s;
            span[2] = e;
            span[3] = 0;
        }
    }
    return MISPMM_OK;
}

// Where a span list (mispmm_csr_spans_by_length_host: rows by decreasing length) divides into long rows for the split body
// and short rows for the row-gather body of mispmm_csr_hybrid_f32: the first position whose row holds at most `threshold`
// entries, rounded up to a multiple of 4 (positions are dealt to workgroups of 4 waves; the chunk groups of the longest
// rows come first and are always covered).  numSpans itself when no row is that short.
extern "C" int mispmm_csr_spans_long_count_host(uint32_t numSpans, const uint32_t *spans_host, uint32_t threshold, uint32_t *numLong_out) {
    if (!numLong_out) return fail(MISPMM_ERR_INVALID_ARG, "csr spans long count: numLong_out is null");
    *numLong_out = 0;
    if (numSpans == 0) return MISPMM_OK;
    if (!spans_host) return fail(MISPMM_ERR_INVALID_ARG, "csr spans long count: spans is null");
    uint32_t p = 0;
    while (p < numSpans) {
        const uint32_t *span = spans_host + static_cast<size_t>(p) * 4u;
        if (span[2] < span[1]) return fail(MISPMM_ERR_INVALID_ARG, "csr spans long count: span %u ends before it starts", p);
        if (span[3] == 0 && span[2] - span[1] <= threshold) break;
        ++p;
    }
    const uint64_t rounded = (static_cast<uint64_t>(p) + 3u) & ~3ull;
    *numLong_out = rounded >= numSpans ? numSpans : static_cast<uint32_t>(rounded);
    return MISPMM_OK;
}

extern "C" int mispmm_coo_sort_by_row_host(uint32_t M, uint32_t nnz, const uint32_t *rowIdxs_host,
                                           const uint32_t *colIdxs_host, const float *vals_host,
                                           uint32_t *rowIdxs_out_host, uint32_t *colIdxs_out_host,
                                           float *vals_out_host, int *was_sorted) {
    if (nnz != 0 && (!rowIdxs_host || !colIdxs_host || !vals_host))
        return fail(MISPMM_ERR_INVALID_ARG, "coo sort: null input");
    bool sorted = true;
    for (uint32_t i = 0; i < nnz; ++i) {
        if (rowIdxs_host[i] >= M) return fail(MISPMM_ERR_INVALID_ARG, "coo sort: row index %u out of range", rowIdxs_host[i]);
        if (i && rowIdxs_host[i] < rowIdxs_host[i - 1]) sorted = false;
    }
    if (was_sorted) *was_sorted = sorted ? 1 : 0;
    if (!rowIdxs_out_host && !colIdxs_out_host && !vals_out_host) return MISPMM_OK;  // query only
    if (!rowIdxs_out_host || !colIdxs_out_host || !vals_out_host)
        return fail(MISPMM_ERR_INVALID_ARG, "coo sort: an output is null");
    // counting sort by row: stable, so every row keeps its entries in storage order -- the order in which
    // the reference's spmmCOOCpu adds them into that row of C (spmm_coo.cpp:16-24)
    std::vector<uint32_t> start(static_cast<size_t>(M) + 1, 0);
    for (uint32_t i = 0; i < nnz; ++i) ++start[rowIdxs_host[i] + 1];
    for (uint32_t r = 0; r < M; ++r) start[r + 1] += start[r];
    for (uint32_t i = 0; i < nnz; ++i) {
        const uint32_t o = start[rowIdxs_host[i]]++;
        rowIdxs_out_host[o] = rowIdxs_host[i];
        colIdxs_out_host[o] = colIdxs_host[i];
        vals_out_host[o] = vals_host[i];
    }
    return MISPMM_OK;
}

extern "C" int mispmm_bsr_nonzeros_host(uint32_t numBlockRows, uint32_t bR, uint32_t bC, uint32_t numBlocks,
                                        const uint32_t *blockRowPtrs_host, const uint32_t *blockColIdxs_host,
                                        const float *blocks_host, uint32_t *nnz_out, uint32_t *rowPtrs_out_host,
                                        uint32_t *colIdxs_out_host, float *vals_out_host) {
    if (!nnz_out) return fail(MISPMM_ERR_INVALID_ARG, "bsr nonzeros: nnz_out is null");
    if (bR == 0 || bC == 0) return fail(MISPMM_ERR_INVALID_ARG, "bsr nonzeros: zero block dimension");
    if (numBlockRows != 0 && !blockRowPtrs_host) return fail(MISPMM_ERR_INVALID_ARG, "bsr nonzeros: blockRowPtrs is null");
    if (numBlocks != 0 && (!blockColIdxs_host || !blocks_host)) return fail(MISPMM_ERR_INVALID_ARG, "bsr nonzeros: null block arrays");
    const bool fill = rowPtrs_out_host && colIdxs_out_host && vals_out_host;
    if (!fill && (rowPtrs_out_host || colIdxs_out_host || vals_out_host))
        return fail(MISPMM_ERR_INVALID_ARG, "bsr nonzeros: give all three outputs or none (size query)");
    // row by row, blocks in storage order, columns ascending inside a block: the order in which spmmBSRCpu
    // (spmm_bsr.cpp:17-38) adds terms into one row of C
    uint64_t n = 0;
    for (uint32_t R = 0; R < numBlockRows; ++R) {
        for (uint32_t i = 0; i < bR; ++i) {
            if (fill) rowPtrs_out_host[static_cast<size_t>(R) * bR + i] = static_cast<uint32_t>(n);
            for (uint32_t b = blockRowPtrs_host[R]; b < blockRowPtrs_host[R + 1]; ++b) {
                if (b >= numBlocks) return fail(MISPMM_ERR_INVALID_ARG, "bsr nonzeros: block index %u out of range", b);
                const float *row = blocks_host + (static_cast<size_t>(b) * bR + i) * bC;
                for (uint32_t j = 0; j < bC; ++j) {
                    if (row[j] == 0.f) continue;  // +0 and -0 alike: the term 0 * b cannot change a finite sum
                    if (fill) {
                        colIdxs_out_host[n] = blockColIdxs_host[b] * bC + j;
                        vals_out_host[n] = row[j];
                    }
                    ++n;
                }
            }
        }
    }
    if (n > 0xFFFFFFFFull) return fail(MISPMM_ERR_UNSUPPORTED, "bsr nonzeros: more than 2^32 entries");
    if (fill) rowPtrs_out_host[static_cast<size_t>(numBlockRows) * bR] = static_cast<uint32_t>(n);
    *nnz_out = static_cast<uint32_t>(n);
    return MISPMM_OK;
}

namespace {
// fp32 -> bf16 bit pattern, round to nearest even; a NaN stays a (quiet) NaN
inline uint16_t bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) return static_cast<uint16_t>((u >> 16) | 0x0040u);
    return static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
}  // namespace

namespace {
// columns of block row R that hold a value which is still non-zero as bf16, as (block, column-in-block) codes
// b * bC + j in the order the blocks store them (= ascending when block columns ascend): the k order of the MFMA steps
int occupied_columns(uint32_t R, uint32_t bC, uint32_t numBlocks, const uint32_t *blockRowPtrs_host, const float *blocks_host,
                     std::vector<uint64_t> &cols) {
    cols.clear();
    for (uint32_t b = blockRowPtrs_host[R]; b < blockRowPtrs_host[R + 1]; ++b) {
        if (b >= numBlocks) return fail(MISPMM_ERR_INVALID_ARG, "bsr compact: block index %u out of range", b);
        const float *blk = blocks_host + static_cast<size_t>(b) * 16u * bC;
        for (uint32_t j = 0; j < bC; ++j) {
            bool any = false;
            for (uint32_t i = 0; i < 16 && !any; ++i) any = (bf16_rne(blk[static_cast<size_t>(i) * bC + j]) & 0x7FFFu) != 0;
            if (any) cols.push_back(static_cast<uint64_t>(b) * bC + j);
        }
    }
    return MISPMM_OK;
}

// step `s` of a block row's occupied-column list -> 32 B-row indices (0xFFFFFFFF = padding: a dropped read) and the
// [16 rows][32 k] bf16 tile of the values gathered to those columns (zero under padding); s past the list = all padding
void fill_step(const std::vector<uint64_t> &cols, uint32_t s, uint32_t bC, const uint32_t *blockColIdxs_host, const float *blocks_host,
               uint32_t *cdst, uint16_t *tdst) {
    for (uint32_t k = 0; k < 32; ++k) {
        const size_t e = static_cast<size_t>(s) * 32 + k;
        if (e < cols.size()) {
            const uint32_t b = static_cast<uint32_t>(cols[e] / bC), j = static_cast<uint32_t>(cols[e] % bC);
            cdst[k] = blockColIdxs_host[b] * bC + j;
            const float *blk = blocks_host + static_cast<size_t>(b) * 16u * bC;
            for (uint32_t i = 0; i < 16; ++i) tdst[i * 32 + k] = bf16_rne(blk[static_cast<size_t>(i) * bC + j]);
        } else {
            cdst[k] = 0xFFFFFFFFu;
            for (uint32_t i = 0; i < 16; ++i) tdst[i * 32 + k] = 0;
        }
    }
}
}  // namespace

extern "C" int mispmm_bsr_compact_bf16_host(uint32_t numBlockRows, uint32_t bR, uint32_t bC, uint32_t numBlocks,
                                            const uint32_t *blockRowPtrs_host, const uint32_t *blockColIdxs_host,
                                            const float *blocks_host, uint32_t *nSteps_out, uint32_t *stepPtrs_out_host,
                                            uint32_t *cols_out_host, uint16_t *tiles_out_host) {
    if (!nSteps_out) return fail(MISPMM_ERR_INVALID_ARG, "bsr compact: nSteps_out is null");
    if (bR != 16 || bC == 0) return fail(MISPMM_ERR_UNSUPPORTED, "bsr compact: block rows of 16 only (got %u x %u blocks)", bR, bC);
    if (numBlockRows != 0 && !blockRowPtrs_host) return fail(MISPMM_ERR_INVALID_ARG, "bsr compact: blockRowPtrs is null");
    if (numBlocks != 0 && (!blockColIdxs_host || !blocks_host)) return fail(MISPMM_ERR_INVALID_ARG, "bsr compact: null block arrays");
    const bool fill = stepPtrs_out_host && cols_out_host && tiles_out_host;
    if (!fill && (stepPtrs_out_host || cols_out_host || tiles_out_host))
        return fail(MISPMM_ERR_INVALID_ARG, "bsr compact: give all three outputs or none (size query)");
    uint64_t steps = 0;
    std::vector<uint64_t> cols;
    for (uint32_t R = 0; R < numBlockRows; ++R) {
        if (int st = occupied_columns(R, bC, numBlocks, blockRowPtrs_host, blocks_host, cols)) return st;
        const uint32_t nsteps = static_cast<uint32_t>((cols.size() + 31) / 32);
        if (fill) {
            stepPtrs_out_host[R] = static_cast<uint32_t>(steps);
            for (uint32_t s = 0; s < nsteps; ++s)
                fill_step(cols, s, bC, blockColIdxs_host, blocks_host, cols_out_host + (steps + s) * 32, tiles_out_host + (steps + s) * 512);
        }
        steps += nsteps;
    }
    if (steps > 0x03FFFFFFull) return fail(MISPMM_ERR_UNSUPPORTED, "bsr compact: too many K steps");
    if (fill) stepPtrs_out_host[numBlockRows] = static_cast<uint32_t>(steps);
    *nSteps_out = static_cast<uint32_t>(steps);
    return MISPMM_OK;
}

extern "C" int mispmm_bsr_compact_slots_bf16_host(uint32_t numBlockRows, uint32_t bR, uint32_t bC, uint32_t numBlocks,
                                                  const uint32_t *blockRowPtrs_host, const uint32_t *blockColIdxs_host,
                                                  const float *blocks_host, uint32_t *nSteps_out, uint32_t *nUsedSteps_out,
                                                  uint32_t *extraPtrs_out_host, uint32_t *cols_out_host, uint16_t *tiles_out_host) {
    constexpr uint32_t kSlots = 4;  // = kBsrSlots of bsr_slots.hpp
    if (!nSteps_out) return fail(MISPMM_ERR_INVALID_ARG, "bsr compact slots: nSteps_out is null");
    if (bR != 16 || bC == 0) return fail(MISPMM_ERR_UNSUPPORTED, "bsr compact slots: block rows of 16 only (got %u x %u blocks)", bR, bC);
    if (numBlockRows != 0 && !blockRowPtrs_host) return fail(MISPMM_ERR_INVALID_ARG, "bsr compact slots: blockRowPtrs is null");
    if (numBlocks != 0 && (!blockColIdxs_host || !blocks_host)) return fail(MISPMM_ERR_INVALID_ARG, "bsr compact slots: null block arrays");
    const bool fill = extraPtrs_out_host && cols_out_host && tiles_out_host;
    if (!fill && (extraPtrs_out_host || cols_out_host || tiles_out_host))
        return fail(MISPMM_ERR_INVALID_ARG, "bsr compact slots: give all three outputs or none (size query)");
    const uint64_t fixed = static_cast<uint64_t>(numBlockRows) * kSlots;
    uint64_t extra = 0, used = 0;
    std::vector<uint64_t> cols;
    for (uint32_t R = 0; R < numBlockRows; ++R) {
        if (int st = occupied_columns(R, bC, numBlocks, blockRowPtrs_host, blocks_host, cols)) return st;
        const uint32_t nsteps = static_cast<uint32_t>((cols.size() + 31) / 32);
        if (fill) {
            extraPtrs_out_host[R] = static_cast<uint32_t>(extra);
            for (uint32_t s = 0; s < std::max(nsteps, kSlots); ++s) {
                const uint64_t at = s < kSlots ? static_cast<uint64_t>(R) * kSlots + s : fixed + extra + (s - kSlots);
                fill_step(cols, s, bC, blockColIdxs_host, blocks_host, cols_out_host + at * 32, tiles_out_host + at * 512);
            }
        }
        used += nsteps;
        if (nsteps > kSlots) extra += nsteps - kSlots;
    }
    if (fixed + extra > 0x03FFFFFFull) return fail(MISPMM_ERR_UNSUPPORTED, "bsr compact slots: too many K steps");
    if (fill) extraPtrs_out_host[numBlockRows] = static_cast<uint32_t>(extra);
    *nSteps_out = static_cast<uint32_t>(fixed + extra);
    if (nUsedSteps_out) *nUsedSteps_out = static_cast<uint32_t>(used);
    return MISPMM_OK;
}

// ---- row clustering (the plan order of mispmm_csr_plan_f32) ----------------------------------------------------------
// Greedy growth of `parts` row clusters of equal size: a cluster starts from the first unassigned row and keeps taking
// the unassigned row that shares the most columns with what the cluster already holds (lazy max-heap keyed by that
// count).  Rows that share B rows then run on one XCD (the row-gather kernel gives an XCD a contiguous range of array
// rows) and close together in time, so a B row fetched for one of them is still in that L2 for the others.  Columns of
// more than 64 entries are ignored when counting (they are in every cluster anyway and would make the walk quadratic).
// order_out[i] = the original row stored at position i; *_distinct_out = sum over the parts of the distinct columns a part
// touches, before (contiguous parts in storage order) and after: the smaller, the fewer B rows an L2 must fetch.
extern "C" int mispmm_csr_cluster_rows_host(uint32_t M, uint32_t K, const uint32_t *rowPtrs_host, const uint32_t *colIdxs_host,
                                            uint32_t parts, uint32_t *order_out_host, uint64_t *natural_distinct_out,
                                            uint64_t *clustered_distinct_out) {
    if (!rowPtrs_host || !order_out_host) return fail(MISPMM_ERR_INVALID_ARG, "cluster_rows: null pointer");
    if (parts == 0) return fail(MISPMM_ERR_INVALID_ARG, "cluster_rows: parts must be positive");
    for (uint32_t r = 0; r < M; ++r)
        if (rowPtrs_host[r + 1] < rowPtrs_host[r]) return fail(MISPMM_ERR_INVALID_ARG, "cluster_rows: rowPtrs decrease at row %u", r);
    const uint64_t nnz = rowPtrs_host[M];
    if (nnz != 0 && !colIdxs_host) return fail(MISPMM_ERR_INVALID_ARG, "cluster_rows: colIdxs is null");
    for (uint64_t i = 0; i < nnz; ++i)
        if (colIdxs_host[i] >= K) return fail(MISPMM_ERR_INVALID_ARG, "cluster_rows: column index %u out of range", colIdxs_host[i]);
    // transpose structure: the rows of every column
    std::vector<uint32_t> colPtr(static_cast<size_t>(K) + 1, 0), colRows(nnz);
    for (uint64_t i = 0; i < nnz; ++i) ++colPtr[colIdxs_host[i] + 1];
    for (uint32_t c = 0; c < K; ++c) colPtr[c + 1] += colPtr[c];
    {
        std::vector<uint32_t> fill(colPtr.begin(), colPtr.end() - 1);
        for (uint32_t r = 0; r < M; ++r)
            for (uint32_t i = rowPtrs_host[r]; i < rowPtrs_host[r + 1]; ++i) colRows[fill[colIdxs_host[i]]++] = r;
    }
    constexpr uint32_t kMaxColumnDegree = 64;
    const uint32_t cap = (M + parts - 1) / parts;
    std::vector<uint8_t> assigned(M, 0);
    std::vector<uint32_t> gain(M, 0), stamp(K, 0xFFFFFFFFu), touched;
    std::vector<std::pair<uint32_t, uint32_t>> heap;  // (gain, row), max-heap; stale entries are skipped when popped
    uint32_t next_seed = 0, placed = 0;
    for (uint32_t p = 0; p < parts && placed < M; ++p) {
        heap.clear();
        for (uint32_t r : touched) gain[r] = 0;
        touched.clear();
        uint32_t size = 0;
        while (size < cap && placed < M) {
            uint32_t r = 0xFFFFFFFFu;
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const auto top = heap.back();
                heap.pop_back();
                if (!assigned[top.second] && gain[top.second] == top.first) {
                    r = top.second;
                    break;
                }
            }
            if (r == 0xFFFFFFFFu) {  // nothing shares a column with the cluster (or it is empty): next row in storage order
                while (next_seed < M && assigned[next_seed]) ++next_seed;
                r = next_seed;
            }
            assigned[r] = 1;
            order_out_host[placed++] = r;
            ++size;
            for (uint32_t i = rowPtrs_host[r]; i < rowPtrs_host[r + 1]; ++i) {
                const uint32_t c = colIdxs_host[i];
                if (stamp[c] == p) continue;  // the cluster already holds this column
                stamp[c] = p;
                if (colPtr[c + 1] - colPtr[c] > kMaxColumnDegree) continue;
                for (uint32_t j = colPtr[c]; j < colPtr[c + 1]; ++j) {
                    const uint32_t r2 = colRows[j];
                    if (assigned[r2]) continue;
                    if (gain[r2]++ == 0) touched.push_back(r2);
                    heap.emplace_back(gain[r2], r2);
                    std::push_heap(heap.begin(), heap.end());
                }
            }
        }
    }
    // the figure of merit, before and after
    auto distinct = [&](auto row_at) {
        std::vector<uint32_t> seen(K, 0xFFFFFFFFu);
        uint64_t total = 0;
        for (uint32_t p = 0; p < parts; ++p)
            for (uint32_t i = p * cap; i < std::min<uint64_t>(M, static_cast<uint64_t>(p + 1) * cap); ++i) {
                const uint32_t r = row_at(i);
                for (uint32_t e = rowPtrs_host[r]; e < rowPtrs_host[r + 1]; ++e)
                    if (seen[colIdxs_host[e]] != p) {
                        seen[colIdxs_host[e]] = p;
                        ++total;
                    }
            }
        return total;
    };
    if (natural_distinct_out) *natural_distinct_out = distinct([](uint32_t i) { return i; });
    if (clustered_distinct_out) *clustered_distinct_out = distinct([&](uint32_t i) { return order_out_host[i]; });
    return MISPMM_OK;
}

// The arrays of the CSR whose row i is row order[i] of the input (order = a permutation of 0 .. M-1): what
// mispmm_csr_plan_f32 multiplies from, with rowMap = order.
extern "C" int mispmm_csr_permute_rows_host(uint32_t M, const uint32_t *rowPtrs_host, const uint32_t *colIdxs_host, const float *vals_host,
                                            const uint32_t *order_host, uint32_t *rowPtrs_out_host, uint32_t *colIdxs_out_host,
                                            float *vals_out_host) {
    if (!rowPtrs_host || !order_host || !rowPtrs_out_host) return fail(MISPMM_ERR_INVALID_ARG, "permute_rows: null pointer");
    for (uint32_t r = 0; r < M; ++r)
        if (rowPtrs_host[r + 1] < rowPtrs_host[r]) return fail(MISPMM_ERR_INVALID_ARG, "permute_rows: rowPtrs decrease at row %u", r);
    const uint64_t nnz = rowPtrs_host[M];
    if (nnz != 0 && (!colIdxs_host || !vals_host || !colIdxs_out_host || !vals_out_host))
        return fail(MISPMM_ERR_INVALID_ARG, "permute_rows: null entry arrays");
    std::vector<uint8_t> seen(M, 0);
    uint64_t at = 0;
    for (uint32_t i = 0; i < M; ++i) {
        const uint32_t r = order_host[i];
        if (r >= M || seen[r]) return fail(MISPMM_ERR_INVALID_ARG, "permute_rows: order is not a permutation (entry %u = %u)", i, r);
        seen[r] = 1;
        rowPtrs_out_host[i] = static_cast<uint32_t>(at);
        const uint32_t len = rowPtrs_host[r + 1] - rowPtrs_host[r];
        if (len) {
            std::memcpy(colIdxs_out_host + at, colIdxs_host + rowPtrs_host[r], static_cast<size_t>(len) * sizeof(uint32_t));
            std::memcpy(vals_out_host + at, vals_host + rowPtrs_host[r], static_cast<size_t>(len) * sizeof(float));
        }
        at += len;
    }
    rowPtrs_out_host[M] = static_cast<uint32_t>(at);
    return MISPMM_OK;
}

// ---- LDS tiles (mispmm_csr_lds_tile_f32): groups of rows that share B rows, with the list of the distinct columns each group
// reads.  Greedy growth like the clustering above, but a tile also stops growing when its column list would exceed `maxCols`
// (the LDS budget of the kernel): a tile keeps taking the unassigned row that shares the most columns with it -- which is the
// row that adds the fewest new ones.  Rows of a tile sit at consecutive plan positions; an entry's `slot` is the position of
// its column in its tile's list (< maxCols <= 256: one byte).  Outputs NULL = size query (*numTiles_out, *numListed_out).
extern "C" int mispmm_csr_tiles_host(uint32_t M, uint32_t K, const uint32_t *rowPtrs_host, const uint32_t *colIdxs_host, uint32_t maxRows,
                                     uint32_t maxCols, uint32_t *numTiles_out, uint32_t *numListed_out, uint32_t *tileRowPtrs_out_host,
                                     uint32_t *tileColPtrs_out_host, uint32_t *tileCols_out_host, uint32_t *order_out_host,
                                     uint8_t *slots_out_host) {
    if (!rowPtrs_host || !numTiles_out || !numListed_out) return fail(MISPMM_ERR_INVALID_ARG, "csr_tiles: null pointer");
    if (maxRows == 0 || maxRows > 16 || maxCols == 0 || maxCols > 256) return fail(MISPMM_ERR_INVALID_ARG, "csr_tiles: 1..16 rows and 1..256 columns per tile");
    for (uint32_t r = 0; r < M; ++r) {
        if (rowPtrs_host[r + 1] < rowPtrs_host[r]) return fail(MISPMM_ERR_INVALID_ARG, "csr_tiles: rowPtrs decrease at row %u", r);
        if (rowPtrs_host[r + 1] - rowPtrs_host[r] > maxCols) return fail(MISPMM_ERR_UNSUPPORTED, "csr_tiles: row %u alone exceeds a tile's %u columns", r, maxCols);
    }
    const uint64_t nnz = rowPtrs_host[M];
    if (nnz != 0 && !colIdxs_host) return fail(MISPMM_ERR_INVALID_ARG, "csr_tiles: colIdxs is null");
    for (uint64_t i = 0; i < nnz; ++i)
        if (colIdxs_host[i] >= K) return fail(MISPMM_ERR_INVALID_ARG, "csr_tiles: column index %u out of range", colIdxs_host[i]);
    const bool fill = tileRowPtrs_out_host && tileColPtrs_out_host && tileCols_out_host && order_out_host && slots_out_host;
    std::vector<uint32_t> colPtr(static_cast<size_t>(K) + 1, 0), colRows(nnz);
    for (uint64_t i = 0; i < nnz; ++i) ++colPtr[colIdxs_host[i] + 1];
    for (uint32_t c = 0; c < K; ++c) colPtr[c + 1] += colPtr[c];
    {
        std::vector<uint32_t> at(colPtr.begin(), colPtr.end() - 1);
        for (uint32_t r = 0; r < M; ++r)
            for (uint32_t i = rowPtrs_host[r]; i < rowPtrs_host[r + 1]; ++i) colRows[at[colIdxs_host[i]]++] = r;
    }
    constexpr uint32_t kMaxColumnDegree = 64;
    std::vector<uint8_t> assigned(M, 0);
    std::vector<uint32_t> gain(M, 0), slotOf(K, 0xFFFFFFFFu), touched, listed;
    std::vector<std::pair<uint32_t, uint32_t>> heap;
    uint32_t next_seed = 0, placed = 0, tiles = 0;
    uint64_t listedTotal = 0, entryAt = 0;
    if (fill) tileRowPtrs_out_host[0] = tileColPtrs_out_host[0] = 0;
    while (placed < M) {
        heap.clear();
        for (uint32_t r : touched) gain[r] = 0;
        touched.clear();
        for (uint32_t c : listed) slotOf[c] = 0xFFFFFFFFu;
        listed.clear();
        uint32_t rows = 0;
        while (rows < maxRows && placed < M) {
            uint32_t r = 0xFFFFFFFFu;
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const auto top = heap.back();
                heap.pop_back();
                if (!assigned[top.second] && gain[top.second] == top.first) {
                    r = top.second;
                    break;
                }
            }
            if (r == 0xFFFFFFFFu) {
                if (rows != 0) break;  // nothing shares a column with the tile any more: close it (a tile of unrelated rows buys nothing)
                while (next_seed < M && assigned[next_seed]) ++next_seed;
                r = next_seed;
            }
            uint32_t fresh = 0;        // the columns row r would add to the list
            for (uint32_t i = rowPtrs_host[r]; i < rowPtrs_host[r + 1]; ++i) fresh += slotOf[colIdxs_host[i]] == 0xFFFFFFFFu;
            if (listed.size() + fresh > maxCols) break;                // (the seed always fits: its row is at most maxCols long)
            assigned[r] = 1;
            if (fill) order_out_host[placed] = r;
            ++placed;
            ++rows;
            for (uint32_t i = rowPtrs_host[r]; i < rowPtrs_host[r + 1]; ++i) {
                const uint32_t c = colIdxs_host[i];
                if (slotOf[c] == 0xFFFFFFFFu) {
                    slotOf[c] = static_cast<uint32_t>(listed.size());
                    listed.push_back(c);
                    if (colPtr[c + 1] - colPtr[c] <= kMaxColumnDegree)
                        for (uint32_t j = colPtr[c]; j < colPtr[c + 1]; ++j) {
                            const uint32_t r2 = colRows[j];
                            if (assigned[r2]) continue;
                            if (gain[r2]++ == 0) touched.push_back(r2);
                            heap.emplace_back(gain[r2], r2);
                            std::push_heap(heap.begin(), heap.end());
                        }
                }
                if (fill) slots_out_host[entryAt] = static_cast<uint8_t>(slotOf[c]);
                ++entryAt;
            }
        }
        if (fill) {
            std::memcpy(tileCols_out_host + listedTotal, listed.data(), listed.size() * sizeof(uint32_t));
            tileRowPtrs_out_host[tiles + 1] = placed;
            tileColPtrs_out_host[tiles + 1] = static_cast<uint32_t>(listedTotal + listed.size());
        }
        listedTotal += listed.size();
        ++tiles;
    }
    if (listedTotal > 0xFFFFFFFFull) return fail(MISPMM_ERR_UNSUPPORTED, "csr_tiles: column lists exceed 2^32 entries");
    *numTiles_out = tiles;
    *numListed_out = static_cast<uint32_t>(listedTotal);
    return MISPMM_OK;
}
