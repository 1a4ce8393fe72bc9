// Host-only entry points of the C ABI (no device work): format conversion and row sharding.
#include <algorithm>
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
