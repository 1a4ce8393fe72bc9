// The row-group gather kernel shared by CSR (kernel 5), ELL and COO: G lanes own one C row, lane i
// fetches entry i of the row's current G-entry chunk, B-row byte offsets are broadcast by lane
// shuffle, up to 16 B-row reads per lane are in flight as dropped-or-live buffer loads, products
// are summed in storage order, and C leaves through non-temporal buffer stores (spmm_common.hpp).  Workgroups are laid
// over the 8 XCDs as a P x Q grid of (row part, column part) so each XCD's L2 holds 1/Q of the
// width of the B rows that 1/P of the matrix rows touch (see k5 in spmm_csr.hip for the
// traffic model).  `Rows` says where a row's entries live:
//   CsrRows  entries [rowPtrs[r], rowPtrs[r+1])            (also a COO whose row bounds were built)
//   UniformRows  CSR with a constant row length: entries [r * width, (r + 1) * width), no row pointers
//   EllRows  entries [r * width, (r + 1) * width), column 0xFFFFFFFF = padding (dropped)
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "spmm_common.hpp"

namespace mispmm {

#ifdef MISPMM_STAMPS
// Diagnostic build only (tools/stamp_headline.py): every wave leaves s_memrealtime stamps (100 MHz) in a side buffer.
static __device__ unsigned long long *mispmm_stamp_buf = nullptr;
constexpr uint32_t kStampLaunches = 32;  // the side buffer keeps the records of the last 32 launches (8 x uint64 per wave)
#endif

struct CsrRows {
    const uint32_t *rowPtrs;
    static constexpr bool kPadded = false;
    __device__ __forceinline__ void extent(uint32_t row, size_t &base, uint32_t &len) const {
        const uint32_t s = rowPtrs[row];
        base = s;
        len = rowPtrs[row + 1] - s;
    }
};

// CSR whose rows all hold exactly `width` entries (rowPtrs[r] == r * width): the row pointer array is
// never read, one dependent memory hop less per wave (measured 4.25 -> 4.08 us on the headline).
struct UniformRows {
    uint32_t width;
    static constexpr bool kPadded = false;
    __device__ __forceinline__ void extent(uint32_t row, size_t &base, uint32_t &len) const {
        base = static_cast<size_t>(row) * width;
        len = width;
    }
};

struct EllRows {
    uint32_t width;
    static constexpr bool kPadded = true;
    __device__ __forceinline__ void extent(uint32_t row, size_t &base, uint32_t &len) const {
        base = static_cast<size_t>(row) * width;
        len = width;
    }
};

// Rows named by a span list (row, start, end, info) as mispmm_csr_spans_by_length_host builds it: position i of the list is
// the i-th row of this launch, its entries are colIdxs / vals [start, end) and its result goes to row `row` of C -- one
// 16-byte read instead of two row pointers and a row map (the short rows of a long-row matrix, csr_hybrid in csr_split.hpp)
struct SpanRows {
    const uint32_t *spans;
    static constexpr bool kPadded = false;
};

// Batched launch (mispmm_csr_batch_f32): blockIdx.z picks one of up to kMaxBatch dense operand / result pairs, all
// multiplied by the same A in ONE launch -- what separates two dependent launches on this chip (~1.2 us of idle
// between the last wave of one and the first of the next: profiles/r2/stamps_default.log) is paid once per batch.
constexpr uint32_t kMaxBatch = 16;
struct BatchPtrs {
    const float *b[kMaxBatch];
    float *c[kMaxBatch];
};
template <bool BATCHED> struct BatchArg {};
template <> struct BatchArg<true> {
    BatchPtrs p;
};

template <int G, int VEC, class Acc, bool CBUF, class Rows, int BLOCK = 256, int UMAX = 16, bool ROLL = false, int SLOTS = 16,
          bool BATCHED = false, bool MAPPED = false>
#ifndef MISPMM_X_WAVES
#define MISPMM_X_WAVES 5  // experiment builds: another register budget for the 8-reads-in-flight rolling body
#endif
// the kernel's body with the workgroup's grid position as arguments (bx, by, bz = blockIdx of a launch of its own; csr_hybrid
// runs it on the workgroups behind those of the split body)
__device__ __forceinline__ void row_gather_body(const uint32_t bx, const uint32_t by, const uint32_t bz,
    // the first 13 dwords are preloaded into SGPRs at wave launch (-amdgpu-kernarg-preload-count): they
    // are exactly what the wave needs to find its row and issue its first loads, so no wave starts
    // with a kernarg fetch in front of the row-pointer fetch
    // tiling: bits 0..7 log2 of the row parts P of the XCD grid; bits 8..31 (CsrRows only) a GUESSED row length w, given
    // when nnz == M * w: the wave fetches its first (col, val) entries from r * w while the row pointers are still on
    // their way and keeps them if the pointers confirm the guess (see the rolling body)
    uint32_t M, uint32_t rb_chunk, uint32_t tiling, uint32_t cols_per_part, uint32_t N, uint32_t ldb, Rows rows,
    const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals, uint32_t b_bytes,
    const float *__restrict__ B_one, float *__restrict__ C_one, uint32_t c_bytes, uint32_t ldc,
    // rows stored in a plan order (mispmm_csr_plan_f32): array row i is row rowMap[i] of C; NULL = identity
    const uint32_t *__restrict__ rowMap, BatchArg<BATCHED> batch
#ifdef MISPMM_STAMPS
    , uint32_t stamp_launch  // which of the last kStampLaunches launches this is: each keeps its own stamp records
#endif
    ) {
#ifdef MISPMM_STAMPS
    unsigned long long stamp[5];
    stamp[0] = wall_clock64();
#endif
#ifdef MISPMM_X_PRIO
    // experiment (tools/r3_headline_variants.sh): a wave that has just started outranks its SIMD's older waves (which are
    // busy with the sums of the gather phase) until its (col, val) reads are on their way
    __builtin_amdgcn_s_setprio(3);
#endif
    const float *__restrict__ B = B_one;
    float *__restrict__ C = C_one;
    if constexpr (BATCHED) {
        B = batch.p.b[bz];
        C = batch.p.c[bz];
    }
    constexpr int GROUPS = BLOCK / G;
    constexpr int U = G < UMAX ? G : UMAX;  // B reads in flight per lane; a row of <= U entries is ONE batch
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t lane = threadIdx.x % G;
    const uint32_t xcd = bx & 7u, slot = bx >> 3;
    const uint32_t log2p = tiling & 0xFFu;
    const uint32_t p = xcd & ((1u << log2p) - 1u), q = xcd >> log2p;
    const uint32_t row = (p * rb_chunk + slot) * GROUPS + threadIdx.x / G;
    const bool row_ok = row < M;
    const uint32_t col0 = q * cols_per_part + by * (G * VEC) + lane * VEC;
    const bool col_ok = col0 < min(N, (q + 1) * cols_per_part);
    size_t row_base = 0;
    uint32_t row_len = 0;
    // CSR: the two row pointers are read unconditionally (row clamped) and only RESOLVED into (row_base, row_len) where a
    // path first needs them -- inside an `if (row_ok)` block hipcc waits for them on the spot, which would put the bet
    // on uniform rows (rolling body) behind the very hop it is meant to overlap
    uint32_t ptr_lo = 0, ptr_hi = 0;
    // where this group's row goes in C (a row map is read behind the row's first (col, val) fetch: load_out_row below)
    uint32_t out_row = row;
    if constexpr (std::is_same_v<Rows, CsrRows>) {
        const uint32_t rr = min(row, M - 1u);
        ptr_lo = rows.rowPtrs[rr];
        ptr_hi = rows.rowPtrs[rr + 1];
    } else if constexpr (std::is_same_v<Rows, SpanRows>) {
        if (row_ok) {
            using u4 = uint32_t __attribute__((ext_vector_type(4)));
            const u4 span = *reinterpret_cast<const u4 *>(rows.spans + static_cast<size_t>(row) * 4u);
            out_row = span[0];
            row_base = span[1];
            row_len = span[2] - span[1];
        }
    } else {
        if (row_ok) rows.extent(row, row_base, row_len);
    }
    auto resolve_extent = [&] {
        if constexpr (std::is_same_v<Rows, CsrRows>) {
            row_base = ptr_lo;
            row_len = row_ok ? ptr_hi - ptr_lo : 0u;
        }
    };
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    // (MAPPED is a template parameter on purpose: as a run-time test of the pointer it cost every launch of the unmapped
    // kernel 0.2 us on the headline -- a kernarg wait and a branch in front of the first B reads)
    auto load_out_row = [&] {
        if constexpr (MAPPED) {
            if (row_ok) out_row = rowMap[row];
        }
    };
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;  // lanes past the column part never fetch
    const uint32_t ldb4 = ldb * 4u;

    // entry `mine` of the row as (B-row byte offset, coefficient); padding = dropped load, zero coefficient
    auto fetch = [&](size_t mine, uint32_t &off, float &val) {
        const uint32_t col = colIdxs[mine];
        val = vals[mine];
        off = col * ldb4;
        if constexpr (Rows::kPadded) {
            if (col == 0xFFFFFFFFu) {
                off = kDropLoad;
                val = 0.f;
            }
        }
    };
    // NB slots j .. j+NB-1 of the current chunk: broadcast (offset, coefficient) from the owning lanes and
    // issue the B reads back to back.  Slots at or past `cnt` (and padding) are dropped loads with a zero
    // coefficient: they add 0 * 0 = +0 to a sum that started at +0 and is never -0 -- an exact no-op.
    auto issue = [&](auto nb_tag, vec_t *bv, float *av, uint32_t my_off, float my_val, uint32_t j, uint32_t cnt) {
        constexpr int NB = decltype(nb_tag)::value;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const uint32_t src = (j + u) & (G - 1);
            const uint32_t off = __shfl(my_off, src, G);
            const float a = __shfl(my_val, src, G);
            bool live = j + u < cnt;
            if constexpr (Rows::kPadded) live = live && off != kDropLoad;
            av[u] = live ? a : 0.f;
            bv[u] = buffer_load_vec<VEC>(rsrc, live ? off + lane_off : kDropLoad, 0);
        }
    };
    auto consume = [&](auto nb_tag, const vec_t *bv, const float *av) {
        constexpr int NB = decltype(nb_tag)::value;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) Acc::mac(acc[v], av[u], vec_get<VEC>(bv[u], v));
        }
    };
    using FullBatch = std::integral_constant<int, U>;

    if constexpr (ROLL && G <= 16) {
        // Rolling window: a row is walked in super-chunks of SC = 16 slots (each lane holds E = SC / G entries of
        // it), U reads are kept in flight, and as soon as slot u has been consumed slot u + U is issued into the
        // registers it freed -- the memory pipe never drains between batches of one row.
        // SLOTS < 16 (G = 16 only): rows known to hold at most SLOTS entries -- the uniform-row entry point picks it
        // from the row length, so a 14-entry row issues 14 reads and no dropped ones
        constexpr int SC = SLOTS, E = (SC + G - 1) / G;
        static_assert(SLOTS == 16 || (SLOTS > U && SLOTS < 16) || (G == 16 && SLOTS == 32), "unsupported slot count");
        uint32_t nxt_off[E];
        float nxt_val[E];
        auto fetch_super = [&](uint32_t base) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint32_t idx = base + e * G + lane;
                fetch(row_base + min(idx, row_len - 1), nxt_off[e], nxt_val[e]);
                // entries past the row end become dropped loads with a zero coefficient here, once per lane, so the
                // per-slot code below needs no liveness test: a broadcast kDropLoad offset stays out of range after
                // the lane's column offset is added (in a column-masked lane the sum wraps to 0: a harmless in-range
                // read whose lane never stores)
                if (idx >= row_len) {
                    nxt_off[e] = kDropLoad;
                    nxt_val[e] = 0.f;
                }
            }
        };
        if constexpr (std::is_same_v<Rows, CsrRows>) {
            // A CSR with nnz == M * w may well have w entries in EVERY row (simplicial boundary matrices, regular graphs,
            // ELL-shaped exports: the headline matrix through the general entry point).  Bet on it: fetch the first entries
            // from r * w right behind the row-pointer reads instead of behind their result -- one dependent memory hop less
            // (n4c6-b13 without the uniform-row hint: 3.67 -> see profiles/r3) -- and fetch again only if some row of the
            // wave turns out different.  The guessed positions are in range because M * w == nnz; results never depend on
            // the guess.
            const uint32_t guess = tiling >> 8;
            if (guess != 0) {
                row_base = static_cast<size_t>(row) * guess;
                row_len = row_ok ? guess : 0u;
                if (row_len != 0) fetch_super(0);
                const size_t bet_base = row_base;
                const uint32_t bet_len = row_len;
                resolve_extent();  // the first use of the row pointers: behind the issue of the guessed fetch
                const bool hit = __all(!row_ok || (row_base == bet_base && row_len == bet_len));
                if (!hit && row_len != 0) fetch_super(0);
            } else {
                resolve_extent();
                if (row_len != 0) fetch_super(0);
            }
        } else {
            if (row_len != 0) fetch_super(0);
        }
        load_out_row();
#ifdef MISPMM_X_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef MISPMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[1] = wall_clock64();
#endif
        for (uint32_t base = 0; base < row_len; base += SC) {
            uint32_t my_off[E];
            float my_val[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                my_off[e] = nxt_off[e];
                my_val[e] = nxt_val[e];
            }
            if (base + SC < row_len) fetch_super(base + SC);
            vec_t bv[U];
            float av[U];
            auto issue_one = [&](auto slot_tag, vec_t &b, float &a) {
                constexpr int S = decltype(slot_tag)::value;
                const uint32_t off = group_bcast<G, S % G>(my_off[S / G]);
                const float c = __builtin_bit_cast(float, group_bcast<G, S % G>(__builtin_bit_cast(uint32_t, my_val[S / G])));
                a = c;
                b = buffer_load_vec<VEC, MISPMM_B_LOAD_AUX>(rsrc, off + lane_off, 0);
            };
            auto consume_one = [&](const vec_t &b, float a) {
                if constexpr (VEC == 4 && std::is_same_v<Acc, AccRefWide>) {
                    Acc::mac4(acc, a, vec_get<VEC>(b, 0), vec_get<VEC>(b, 1), vec_get<VEC>(b, 2), vec_get<VEC>(b, 3));
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) Acc::mac(acc[v], a, vec_get<VEC>(b, v));
                }
            };
            // is any live entry of this wave in a slot at or past `from`?  (entries past a row's end and ELL padding
            // already carry kDropLoad, so trailing padding counts as absent)
            auto live_from = [&](auto from_tag) {
                constexpr int FROM = decltype(from_tag)::value;
                bool live = false;
                static_for<0, E>([&](auto e_tag) {
                    constexpr int E0 = decltype(e_tag)::value * G;  // first slot held in register e
                    if constexpr (E0 + G > FROM) {
                        const bool in_range = (E0 >= FROM) || (lane >= static_cast<uint32_t>(FROM > E0 ? FROM - E0 : 0));
                        live = live || (in_range && my_off[decltype(e_tag)::value] != kDropLoad);
                    }
                });
                return __any(live);
            };
            // rolling body over SCX slots: U reads in flight, slot s + U issued as soon as slot s has been summed
            auto roll = [&](auto scx_tag) {
                constexpr int SCX = decltype(scx_tag)::value;
                static_for<0, U>([&](auto s) { issue_one(s, bv[decltype(s)::value], av[decltype(s)::value]); });
                __builtin_amdgcn_sched_barrier(0);
                // (skipping single slots behind wave-uniform branches was measured slower, 3.74 -> 4.35 us on the
                // headline; the body stays straight-line and only its length is chosen per super-chunk)
                static_for<0, SCX>([&](auto s) {
                    constexpr int S = decltype(s)::value;
                    consume_one(bv[S % U], av[S % U]);
                    if constexpr (S + U < SCX) {
                        // the refill may not be hoisted above the consume it waits for: an empty asm that "rewrites"
                        // the sums and clobbers memory sits between them (sched_barrier alone does not order the
                        // side-effect-free multiply-adds at instruction selection)
                        if constexpr (VEC == 4) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
                        else if constexpr (VEC == 2) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]) : : "memory");
                        else asm volatile("" : "+v"(acc[0]) : : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        issue_one(std::integral_constant<int, S + U>{}, bv[S % U], av[S % U]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            };
            if constexpr (Rows::kPadded) {
                if (!live_from(std::integral_constant<int, 0>{})) continue;  // a super-chunk of padding only
            }
            if (!live_from(std::integral_constant<int, U>{})) {
                static_for<0, U>([&](auto s) { issue_one(s, bv[decltype(s)::value], av[decltype(s)::value]); });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, U>([&](auto s) { consume_one(bv[decltype(s)::value], av[decltype(s)::value]); });
            } else if constexpr (SC == 16) {
                // ragged rows: the body is as long as the longest row of the wave needs, in steps of two slots (each
                // dropped slot still costs its broadcasts, its load issue and its sums: the general entry point ran
                // 3.88 us on the 14-entry rows of n4c6-b13 with the 16-slot body against 3.55 us behind the uniform hint)
#ifdef MISPMM_X_FEWBODIES
                // experiment: two body lengths instead of four (code size of the general kernel vs dropped slots)
                if (!live_from(std::integral_constant<int, 14>{})) roll(std::integral_constant<int, 14>{});
                else roll(std::integral_constant<int, 16>{});
#else
                if (!live_from(std::integral_constant<int, 10>{})) roll(std::integral_constant<int, 10>{});
                else if (!live_from(std::integral_constant<int, 12>{})) roll(std::integral_constant<int, 12>{});
                else if (!live_from(std::integral_constant<int, 14>{})) roll(std::integral_constant<int, 14>{});
                else roll(std::integral_constant<int, 16>{});
#endif
            } else {
                roll(std::integral_constant<int, SC>{});
            }
        }
    } else if (resolve_extent(), !__any(row_len > static_cast<uint32_t>(U))) {
        // Every row of this wave fits one batch: one (col, val) fetch, U B reads in flight, one pass of
        // multiply-adds.  Wave-uniform branch.
        load_out_row();
        if (row_len != 0) {
            uint32_t my_off;
            float my_val;
            fetch(row_base + min(lane, row_len - 1), my_off, my_val);
            vec_t bv[U];
            float av[U];
            issue(FullBatch{}, bv, av, my_off, my_val, 0, row_len);
            __builtin_amdgcn_sched_barrier(0);
            consume(FullBatch{}, bv, av);
        }
    } else {
        // Long rows: chunks of G entries with the next chunk's (col, val) prefetched; inside a chunk full
        // batches of U reads, each consumed before the next is issued.  (A two-stage half-batch pipeline
        // was measured slower here, GL7d25: 33.7 vs 28.5 us.  One row's sum is sequential by contract, so
        // a 422-entry row is bounded by U reads in flight per lane)
        uint32_t nxt_off = 0;
        float nxt_val = 0.f;
        if (row_len != 0) fetch(row_base + min(lane, row_len - 1), nxt_off, nxt_val);
        load_out_row();
        for (uint32_t base = 0; base < row_len; base += G) {
            const uint32_t cnt = min(static_cast<uint32_t>(G), row_len - base);
            const uint32_t my_off = nxt_off;
            const float my_val = nxt_val;
            if (base + G < row_len) fetch(row_base + min(base + G + lane, row_len - 1), nxt_off, nxt_val);
            for (uint32_t j = 0; j < cnt; j += U) {
                vec_t bv[U];
                float av[U];
                issue(FullBatch{}, bv, av, my_off, my_val, j, cnt);
                __builtin_amdgcn_sched_barrier(0);
                consume(FullBatch{}, bv, av);
            }
        }
    }
#ifdef MISPMM_STAMPS
    if constexpr (VEC == 4) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
    stamp[2] = wall_clock64();
#endif
    if (row_ok && col_ok) {
        vec_t out;
#pragma unroll
        for (int v = 0; v < VEC; ++v) vec_set<VEC>(out, v, Acc::finish(acc[v]));
        if constexpr (CBUF) {  // buffer stores with the library's C store policy (non-temporal); else plain global stores
            buffer_store_vec_c<VEC>(make_rsrc(C, c_bytes), (out_row * ldc + col0) * 4u, out);
        } else {
            store_vec<VEC>(C + static_cast<size_t>(out_row) * ldc + col0, out);
        }
    }
#ifdef MISPMM_STAMPS
    stamp[3] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[4] = wall_clock64();
    if (mispmm_stamp_buf && (threadIdx.x & 63) == 0) {
        const uint32_t wpb = blockDim.x / 64;
        const size_t waves_per_launch = static_cast<size_t>(gridDim.x) * gridDim.y * wpb;
        unsigned long long *o = mispmm_stamp_buf + (stamp_launch * waves_per_launch +
            (static_cast<size_t>(by) * gridDim.x + bx) * wpb + (threadIdx.x >> 6)) * 8;
#pragma unroll
        for (int i = 0; i < 5; ++i) o[i] = stamp[i];
        // where the wave ran: HW_REG_HW_ID (id 4: simd [5:4], cu [11:8], sh [12], se [15:13]) and HW_REG_XCC_ID (id 20)
        o[5] = static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 4));
        o[6] = static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20));
    }
#endif
}

template <int G, int VEC, class Acc, bool CBUF, class Rows, int BLOCK = 256, int UMAX = 16, bool ROLL = false, int SLOTS = 16,
          bool BATCHED = false, bool MAPPED = false>
__global__ __launch_bounds__(BLOCK, ROLL ? (UMAX > 8 ? 3 : MISPMM_X_WAVES) : 1) void row_gather_kernel(
    uint32_t M, uint32_t rb_chunk, uint32_t tiling, uint32_t cols_per_part, uint32_t N, uint32_t ldb, Rows rows,
    const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals, uint32_t b_bytes,
    const float *__restrict__ B_one, float *__restrict__ C_one, uint32_t c_bytes, uint32_t ldc,
    const uint32_t *__restrict__ rowMap, BatchArg<BATCHED> batch
#ifdef MISPMM_STAMPS
    , uint32_t stamp_launch
#endif
    ) {
    row_gather_body<G, VEC, Acc, CBUF, Rows, BLOCK, UMAX, ROLL, SLOTS, BATCHED, MAPPED>(blockIdx.x, blockIdx.y, blockIdx.z, M, rb_chunk, tiling,
                                                                                    cols_per_part, N, ldb, rows, colIdxs, vals, b_bytes, B_one,
                                                                                    C_one, c_bytes, ldc, rowMap, batch
#ifdef MISPMM_STAMPS
                                                                                    , stamp_launch
#endif
    );
}

// ---- host side -------------------------------------------------------------------------------
template <class Rows> constexpr const char *rows_tag() {
    if constexpr (std::is_same_v<Rows, CsrRows>) return "csr";
    else if constexpr (std::is_same_v<Rows, UniformRows>) return "uniform";
    else if constexpr (std::is_same_v<Rows, SpanRows>) return "spans";
    else return "ell";
}

struct RowGatherArgs {
    hipStream_t stream;
    uint32_t M, K;
    const uint32_t *colIdxs;
    const float *vals, *B;
    uint32_t N, ldb;
    float *C;
    uint32_t ldc;
    uint32_t mean_row_len = 0;  // nnz / M when the caller knows it (CSR, COO); 0 = unknown
    const uint32_t *rowMap = nullptr;  // rows stored in a plan order: array row i is row rowMap[i] of C (NULL = identity)
    uint32_t row_len_guess = 0;        // CsrRows: nnz / M when nnz == M * (nnz / M), else 0 (the kernel's bet on uniform rows)
    uint32_t batch = 0;         // > 0: B / C are ignored, product i uses B_list[i] / C_list[i] (host arrays), i < batch <= kMaxBatch
    const float *const *B_list = nullptr;
    float *const *C_list = nullptr;
};

// set by a launch helper that declined to launch (a row-mapped launch routed to a body without a mapped form): the entry
// point that asked for the launch reads and clears it and returns MISPMM_ERR_UNSUPPORTED instead of a wrong product
inline bool &row_gather_declined() {
    static thread_local bool declined = false;
    return declined;
}

// the row length the general CSR kernel bets on: nnz / M when the entries divide evenly over the rows, else 0 (no bet)
inline uint32_t uniform_guess(uint32_t M, uint64_t nnz) {
    return (M != 0 && nnz != 0 && nnz % M == 0) ? static_cast<uint32_t>(nnz / M) : 0u;
}

// P x Q XCD grid and the C store flavour (`sc1`, name kept from round 2: C through buffer stores with the policy of
// spmm_common.hpp -- non-temporal -- instead of plain global stores).  MISPMM_CSR_TILING="P,Q" and MISPMM_STORE_SC1=0/1 override
// the defaults (measurement aid; results never depend on them).
struct XcdTiling {
    uint32_t log2p, q;
    bool sc1;
};

inline XcdTiling xcd_tiling(uint32_t N, int vec, uint32_t K) {
    // environment overrides are read once per process
    struct Env {
        int log2p = -1;
        uint32_t q = 0;
        int sc1 = -1;
        Env() {
            if (const char *e = knob_str("MISPMM_CSR_TILING")) {
                unsigned pp = 8, qq = 1;
                if (sscanf(e, "%u,%u", &pp, &qq) == 2 && pp * qq == 8 && (pp == 1 || pp == 2 || pp == 4 || pp == 8)) {
                    log2p = pp == 1 ? 0 : pp == 2 ? 1 : pp == 4 ? 2 : 3;
                    q = qq;
                }
            }
            if (const char *e = knob_str("MISPMM_STORE_SC1")) sc1 = e[0] != '0';
        }
    };
    static const Env env;
    // default: column parts of 64 columns (16 lanes x 4) -- Q = N / 64 up to 8, P = 8 / Q row parts.
    // Measured on n4c6-b13: N = 128 -> 4 x 2 (4.25 us vs 4.41 us for 8 x 1); N = 512 -> 1 x 8 (15.5 us vs
    // 16.0 / 17.0 / 18.8 us for 2 x 4 / 4 x 2 / 8 x 1; two passes of 32-column parts: 15.3 us).  Narrow or odd N keeps 8 x 1.
    XcdTiling t{3u, 1u, true};
    if (vec >= 2 && N >= 128 && N % 64 == 0) {
        uint32_t q = 2;
        while (q < 8 && N / (q * 2) >= 64 && N % (q * 2 * 32) == 0) q *= 2;  // parts of >= 64 columns, whole 32-column groups
        // N = 256: with 2 x 4 an XCD reads a 64-column (256-byte) slice of the B rows half of the matrix touches; when
        // that slice cannot stay in its 4 MiB L2 (K x 256 B > 4 MiB) eight 32-column parts win -- every XCD sees all
        // rows but fetches only its own 128-byte slice of each B row (compulsory fills only): in-process A/B at
        // N = 256 (profiles/r3/tiling_ab.log) n4c6-b13 (K = 25 605) 6.16 us against 7.63 for 2 x 4, ACTIVSg10K
        // (K = 20 000) 12.0 / 12.7; where B is small 2 x 4 is as good or better (delaunay_n12 3.44 / 3.33,
        // ch7-6-b5 3.61 / 3.56, g7jac010 with K = 2 880: 5.19 / 4.39).  At N = 128 the same move loses (3.55 vs 3.39 us).
        if (q == 4 && N == 256 && static_cast<uint64_t>(K) * 256u > (4ull << 20)) q = 8;
        t.q = q;
        t.log2p = q == 2 ? 2u : q == 4 ? 1u : 0u;
    }
    if (env.log2p >= 0) {
        t.log2p = static_cast<uint32_t>(env.log2p);
        t.q = env.q;
    }
    // a column part must hold at least one 8-lane group of whole vectors
    while (t.q > 1 && (N % t.q != 0 || (N / t.q) % (8u * vec) != 0)) {
        t.q >>= 1;
        ++t.log2p;
    }
    if (env.sc1 >= 0) t.sc1 = env.sc1 != 0;
    return t;
}

template <int G, int VEC, class Acc, class Rows, int BLOCK, int UMAX = 16, bool ROLL = false, int SLOTS = 16>
void launch_row_gather_b(const RowGatherArgs &a, const Rows &rows, const XcdTiling &t) {
    const uint32_t cols_per_part = a.N / t.q;
    const uint32_t rb = ceil_div(a.M, BLOCK / G);
    const uint32_t rb_chunk = ceil_div(rb, 1u << t.log2p);
    dim3 grid(8u * rb_chunk, ceil_div(cols_per_part, G * VEC));
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
    const uint64_t c_bytes = static_cast<uint64_t>(a.M) * a.ldc * 4u;
    static const bool guess_on = knob_int("MISPMM_ROW_GUESS", 1) != 0;  // MISPMM_ROW_GUESS=0: no bet on uniform rows (measurement aid)
    const uint32_t tiling = t.log2p | ((std::is_same_v<Rows, CsrRows> && ROLL && guess_on && a.row_len_guess < (1u << 24)) ? a.row_len_guess << 8 : 0u);
    note_kernel("row_gather<G%d,V%d,%s,%s,B%d,U%d,%s,S%d> xcd %ux%u%s%s", G, VEC, acc_tag<Acc>(), rows_tag<Rows>(), BLOCK, UMAX,
                ROLL ? "roll" : "batch", SLOTS, 1u << t.log2p, t.q, a.batch ? " batched" : "", a.rowMap ? " plan-order" : "");
#ifdef MISPMM_STAMPS
    static uint32_t stamp_counter = 0;
    const uint32_t stamp_launch = stamp_counter++ % kStampLaunches;
#define MISPMM_STAMP_ARG , stamp_launch
#else
#define MISPMM_STAMP_ARG
#endif
    if constexpr (ROLL && BLOCK == 128 && UMAX == 8 && G <= 16) {
        if (a.batch > 0) {  // one launch for the whole batch (callers checked sc1 / c_bytes / batch <= kMaxBatch)
            BatchArg<true> arg;
            for (uint32_t i = 0; i < kMaxBatch; ++i) {
                arg.p.b[i] = a.B_list[i < a.batch ? i : 0];
                arg.p.c[i] = a.C_list[i < a.batch ? i : 0];
            }
            grid.z = a.batch;
            if (a.rowMap)
                hipLaunchKernelGGL((row_gather_kernel<G, VEC, Acc, true, Rows, BLOCK, UMAX, ROLL, SLOTS, true, true>), grid, dim3(BLOCK), 0, a.stream,
                                   a.M, rb_chunk, tiling, cols_per_part, a.N, a.ldb, rows, a.colIdxs, a.vals, b_bytes, nullptr, nullptr,
                                   static_cast<uint32_t>(c_bytes), a.ldc, a.rowMap, arg MISPMM_STAMP_ARG);
            else
                hipLaunchKernelGGL((row_gather_kernel<G, VEC, Acc, true, Rows, BLOCK, UMAX, ROLL, SLOTS, true>), grid, dim3(BLOCK), 0, a.stream,
                                   a.M, rb_chunk, tiling, cols_per_part, a.N, a.ldb, rows, a.colIdxs, a.vals, b_bytes, nullptr, nullptr,
                                   static_cast<uint32_t>(c_bytes), a.ldc, nullptr, arg MISPMM_STAMP_ARG);
            return;
        }
    }
    if (a.batch > 0) {  // this instantiation has no batched form (measurement overrides can route here): one launch each
        for (uint32_t i = 0; i < a.batch; ++i) {
            RowGatherArgs one = a;
            one.batch = 0;
            one.B = a.B_list[i];
            one.C = a.C_list[i];
            launch_row_gather_b<G, VEC, Acc, Rows, BLOCK, UMAX, ROLL, SLOTS>(one, rows, t);
        }
        return;
    }
    if constexpr (ROLL && BLOCK == 128 && UMAX == 8 && G <= 16 && VEC == 4) {
        if (a.rowMap) {  // plan order: the mapped form of the main body only (row_gather_supports_map)
            hipLaunchKernelGGL((row_gather_kernel<G, VEC, Acc, true, Rows, BLOCK, UMAX, ROLL, SLOTS, false, true>), grid, dim3(BLOCK), 0, a.stream,
                               a.M, rb_chunk, tiling, cols_per_part, a.N, a.ldb, rows, a.colIdxs, a.vals, b_bytes, a.B, a.C,
                               static_cast<uint32_t>(c_bytes), a.ldc, a.rowMap, BatchArg<false>{} MISPMM_STAMP_ARG);
            return;
        }
    }
    if (a.rowMap) {
        // Unreachable in the production library (row_gather_supports_map admits only shapes whose default body has a mapped
        // form); in the tuning build row_gather_supports_map also declines when a knob forces another body.  Should a new
        // dispatch path ever get here: never multiply unmapped -- nothing is launched and the caller reports UNSUPPORTED.
        row_gather_declined() = true;
        return;
    }
    if (t.sc1 && c_bytes <= 0x7FFFFFFFull)
        hipLaunchKernelGGL((row_gather_kernel<G, VEC, Acc, true, Rows, BLOCK, UMAX, ROLL, SLOTS>), grid, dim3(BLOCK), 0, a.stream, a.M, rb_chunk,
                           tiling, cols_per_part, a.N, a.ldb, rows, a.colIdxs, a.vals, b_bytes, a.B, a.C,
                           static_cast<uint32_t>(c_bytes), a.ldc, a.rowMap, BatchArg<false>{} MISPMM_STAMP_ARG);
    else
        hipLaunchKernelGGL((row_gather_kernel<G, VEC, Acc, false, Rows, BLOCK, UMAX, ROLL, SLOTS>), grid, dim3(BLOCK), 0, a.stream, a.M, rb_chunk,
                           tiling, cols_per_part, a.N, a.ldb, rows, a.colIdxs, a.vals, b_bytes, a.B, a.C, 0u, a.ldc, a.rowMap, BatchArg<false>{} MISPMM_STAMP_ARG);
#undef MISPMM_STAMP_ARG
}

template <int G, int VEC, class Acc, class Rows>
void launch_row_gather(const RowGatherArgs &a, const Rows &rows, const XcdTiling &t) {
    // 128-thread workgroups: with 256 threads a CU holds 3.08 workgroups on the headline, so some CUs
    // carry 4 and finish late; halving the granule measured 4.49 -> 4.24 us (REFERENCE) and 4.37 -> 4.02 us
    // (FAST); 64 threads gave 4.45 / 4.10.  MISPMM_BLOCK=64|128|256 overrides (measurement aid).
    static const int block = knob_int("MISPMM_BLOCK", 128);
    // 8 reads in flight per lane, not 16: 74 instead of 120 VGPRs lets 6 waves per SIMD stay resident, which
    // beats finishing a 14-entry row in one batch (same box, headline: 4.25 -> 4.21 us REFERENCE, 4.43 -> 3.78 us
    // FAST; K = 256: 8.15 -> 7.59 us; K = 512: 15.5 -> 14.7 us).  MISPMM_UMAX=16 restores the deep batch.
    static const int umax = knob_int("MISPMM_UMAX", 8);
    // Rolling refill (G <= 16): measured on n4c6-b13, REFERENCE / FAST us: N = 128 4.06 -> 4.00 / 3.69 -> 3.70,
    // N = 256 7.01 -> 6.79 / 6.65 -> 6.23, N = 512 14.38 -> 13.96 / 14.05 -> 13.82; 12 reads in flight instead of 8
    // changed nothing.  MISPMM_ROLL=0 restores the batch-at-a-time body.
    static const int roll = knob_int("MISPMM_ROLL", 1);
    if constexpr ((G == 16 || G == 8) && VEC == 4 && (std::is_same_v<Rows, UniformRows> || std::is_same_v<Rows, EllRows>)) {
        // rows of 9..14 slots, all the same length (uniform CSR, or ELL of that width): no dead slots
        // (MISPMM_SLOTS=0 keeps the generic 16)
        static const bool slots = knob_int("MISPMM_SLOTS", 1) != 0;
        if (roll && slots) {
            if (rows.width > 8 && rows.width <= 10) return launch_row_gather_b<G, VEC, Acc, Rows, 128, 8, true, 10>(a, rows, t);
            if (rows.width > 10 && rows.width <= 12) return launch_row_gather_b<G, VEC, Acc, Rows, 128, 8, true, 12>(a, rows, t);
            // (4 / 5 / 6 / 7 reads in flight instead of 8 on the headline: 3.85 / 3.74 / 3.72 / 3.67 us against 3.60-3.73)
#ifdef MISPMM_TUNING
            if constexpr (G == 16 && std::is_same_v<Rows, UniformRows>) {  // workgroup size of the headline's kernel (measurement aid)
                if (rows.width > 12 && rows.width <= 14 && block == 64) return launch_row_gather_b<G, VEC, Acc, Rows, 64, 8, true, 14>(a, rows, t);
                if (rows.width > 12 && rows.width <= 14 && block == 256) return launch_row_gather_b<G, VEC, Acc, Rows, 256, 8, true, 14>(a, rows, t);
            }
#endif
            if (rows.width > 12 && rows.width <= 14) return launch_row_gather_b<G, VEC, Acc, Rows, 128, 8, true, 14>(a, rows, t);
        }
    }
    // (A kernel specialised on the bet of the general entry point -- CsrRows with exactly nnz / M slots per super-chunk, i.e. the
    // uniform-row kernel's instruction stream behind the pointer reads -- was built and measured no faster than the ragged
    // five-body kernel with the bet: 3.73-3.81 vs 3.72-3.77 us on one box, profiles/r3/general_entry.log; not kept.)
    if constexpr (G == 16 && VEC == 4 && std::is_same_v<Rows, CsrRows>) {
        // Long rows: a row's sum is sequential, so its time is (entries / reads in flight) x latency.  Matrices whose
        // MEAN row already fills the 16-slot window (nnz >= 24 M; GL7d25: mean 29, longest 422) take 16 reads in
        // flight over 32-slot super-chunks: 116 instead of 74 VGPRs, which such short grids do not miss.
        // MISPMM_DEEP=0/1 forces the choice (measurement aid).
        static const int deep_env = knob_int("MISPMM_DEEP", -1);
        const bool deep = deep_env >= 0 ? deep_env != 0 : a.mean_row_len >= 24;
        if (roll && deep) return launch_row_gather_b<G, VEC, Acc, Rows, 128, 16, true, 32>(a, rows, t);
    }
    if constexpr (G <= 16) {
        if (roll) return launch_row_gather_b<G, VEC, Acc, Rows, 128, 8, true>(a, rows, t);
    }
    if (block == 64) launch_row_gather_b<G, VEC, Acc, Rows, 64, 8>(a, rows, t);
    else if (block == 256) launch_row_gather_b<G, VEC, Acc, Rows, 256, 8>(a, rows, t);
    else if (umax == 16) launch_row_gather_b<G, VEC, Acc, Rows, 128, 16>(a, rows, t);
    else launch_row_gather_b<G, VEC, Acc, Rows, 128, 8>(a, rows, t);
}

// Which shapes have a row-mapped (plan order) form: the main rolling body -- 16-byte vectors, column parts of whole
// 32-column groups, C below 2 GiB (buffer stores), short rows.  mispmm_csr_plan_f32 declines everything else.
inline bool row_gather_supports_map(uint32_t M, uint32_t K, uint32_t N, uint32_t ldc, int vec, uint32_t mean_row_len) {
    if (vec != 4 || static_cast<uint64_t>(M) * ldc * 4u > 0x7FFFFFFFull || mean_row_len >= 24) return false;
    // tuning build: a measurement knob that forces a body without a mapped form (batch-at-a-time, another workgroup size,
    // 16 reads in flight, wide lane groups, the deep body) declines the plan as a shape would -- the callers then multiply
    // from the unpermuted arrays (the production build folds all of these to their defaults)
    if (knob_int("MISPMM_ROLL", 1) == 0 || knob_int("MISPMM_BLOCK", 128) != 128 || knob_int("MISPMM_UMAX", 8) != 8 ||
        knob_int("MISPMM_GROUP", 0) > 16 || knob_int("MISPMM_DEEP", -1) == 1)
        return false;
    const XcdTiling t = xcd_tiling(N, vec, K);
    return t.sc1 && (N / t.q) % 32 == 0;
}

// needs K * ldb * 4 <= 0x7FFFFFFF (buffer offsets; bit 31 marks dropped loads): callers check
template <class Acc, class Rows>
void launch_row_gather_auto(const RowGatherArgs &a, const Rows &rows, int vec) {
    const XcdTiling t = xcd_tiling(a.N, vec, a.K);
    static const int group_env = knob_int("MISPMM_GROUP", 0);
    // 16 or 8 lanes per row (64 or 32 columns at VEC = 4) whenever they tile the column part exactly: several
    // sub-parts per XCD part (grid.y) instead of one wide, partly idle lane group -- and the rolling body, which
    // exists for G <= 16 only (N = 96 / 192 / 384 ran 3.98 / 6.31 / 12.9 us with 32- and 64-lane groups)
    const uint32_t cpp = a.N / t.q;
    const int g = group_env ? group_env
                  : (vec == 4 && cpp % 64 == 0) ? 16
                  : (vec == 4 && cpp % 32 == 0) ? 8
                                                : pick_group(cpp, vec);
#define MISPMM_RG_CASE(GG, VV)                                     \
    if (g == GG && vec == VV) {                                    \
        launch_row_gather<GG, VV, Acc, Rows>(a, rows, t);          \
        return;                                                    \
    }
    MISPMM_RG_CASE(8, 4) MISPMM_RG_CASE(16, 4) MISPMM_RG_CASE(32, 4) MISPMM_RG_CASE(64, 4)
    MISPMM_RG_CASE(8, 2) MISPMM_RG_CASE(16, 2) MISPMM_RG_CASE(32, 2) MISPMM_RG_CASE(64, 2)
    MISPMM_RG_CASE(8, 1) MISPMM_RG_CASE(16, 1) MISPMM_RG_CASE(32, 1) MISPMM_RG_CASE(64, 1)
#undef MISPMM_RG_CASE
}

}  // namespace mispmm
