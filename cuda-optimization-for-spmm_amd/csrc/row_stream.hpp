// The PERSISTENT form of the row-gather kernel (row_gather.hpp) for rows of one width (a CSR with a constant row length,
// an ELL): a launch of exactly as many workgroups as the chip holds at once, every lane group WALKS a strided run of rows and
// keeps its B-row reads rolling ACROSS the row boundaries -- while the last reads of row i are being summed the first reads of
// row i + 1 are already in flight, the (col, val) entries of row i + 2 are on their way, and row i's C leaves through a
// non-temporal store nobody waits for.
//
// Why it was built (round 4, profiles/r4/stamps_headline_k128_k512_streamed.log; profiles/r3 for the resident stamps): at N = 512 the one-row-per-lane-group launch is 12 608 waves for 5 120 wave
// slots, and a wave spends 1.0 us waiting for its 56 (col, val) entries, 2.1 us with B reads in flight and 0.5 us draining its
// store -- half of a resident wave's life has nothing in flight, and with B streamed from HBM (reads of ~3 us) even less.  The
// north_star's kernel clause asks for exactly this shape ("one-wavefront-per-row segmented reduction" walking rows; the
// ancestor is the warp-per-row loop of reference/src/spmm/csr/spmm_csr_k3.cu:9-56).
// What came of it (profiles/r4/stream_ab.log): equal to the one-row-per-lane-group launch where a lane group has one row,
// 4-8 % slower where it has two -- these launches are bound by what an XCD pulls over its fabric link, not by a wave's dead time.
// Compiled into the tuning build only (spmm_stream.hip).
//
// Arithmetic is that of row_gather.hpp: one lane owns its C elements and sums a row's products in storage order with the
// reference's rounding sequence, so REFERENCE mode stays bit-exact; a row past the end of a lane group's run and the slots past
// a row's width are dropped loads (offset bit 31: zeros, no traffic) with a zero coefficient -- exact no-ops.
#pragma once
#include "row_gather.hpp"

namespace mispmm {

#ifndef MISPMM_STREAM_WAVES
#define MISPMM_STREAM_WAVES 5  // register budget: waves per SIMD the allocator must leave room for
#endif

// G lanes own one C row x (G * 4) columns; W = slots per row the body is unrolled for (width <= W), U = W / 2 reads in flight
// per lane: slot s + U is issued into the registers slot s has just freed, slots W .. W + U - 1 are slots 0 .. U - 1 of the NEXT row.
template <int G, class Acc, bool PADDED, int W, bool MAPPED>
__global__ __launch_bounds__(128, MISPMM_STREAM_WAVES) void row_stream_kernel(
    // the first 13 dwords arrive in SGPRs at wave launch (-amdgpu-kernarg-preload-count): everything the wave needs to find its
    // rows and issue its first (col, val) and B reads
    const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals, const float *__restrict__ B, uint32_t b_bytes, uint32_t width,
    // tiling: bits 0..7 log2 of the row parts P of the XCD grid, bits 8..31 the workgroups per XCD (gridDim.x / 8: read from the
    // implicit arguments it would be a scalar memory hop in front of everything)
    uint32_t ldb, uint32_t M, uint32_t rows_per_part, uint32_t tiling, uint32_t cols_per_part, uint32_t N,
    float *__restrict__ C, uint32_t c_bytes, uint32_t ldc, const uint32_t *__restrict__ rowMap) {
    constexpr int VEC = 4, BLOCK = 128, GROUPS = BLOCK / G, U = W / 2, E = (W + G - 1) / G;
    static_assert(W % 2 == 0 && W <= 16 && (G == 8 || G == 16), "unsupported shape");
    using vec_t = f32x4;
    const uint32_t lane = threadIdx.x % G, grp = threadIdx.x / G;
    // the XCD grid of row_gather.hpp: workgroup b runs on XCD b % 8 = (row part p, column part q)
    const uint32_t xcd = blockIdx.x & 7u, wg = blockIdx.x >> 3, nwg = tiling >> 8, log2p = tiling & 0xFFu;
    const uint32_t a_bytes = M * width * 4u;                   // < 2 GiB (the launcher checks)
    const uint32_t p = xcd & ((1u << log2p) - 1u), q = xcd >> log2p;
    const uint32_t row_begin = p * rows_per_part;
    const uint32_t row_end = min(M, row_begin + rows_per_part);
    const uint32_t col0 = q * cols_per_part + blockIdx.y * (G * VEC) + lane * VEC;
    const bool col_ok = col0 < min(N, (q + 1) * cols_per_part);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;  // lanes past the column part never fetch
    const uint32_t ldb4 = ldb * 4u;
    const uint32_t stride = nwg * GROUPS;                      // rows between two rows of one lane group
    const uint32_t wg_row = row_begin + wg * GROUPS;
    if (wg_row >= row_end) return;                             // the whole workgroup has no row (wave-uniform)
    const uint32_t iters = (row_end - wg_row + stride - 1u) / stride;
    uint32_t row = wg_row + grp;
    const rsrc_t brs = make_rsrc(B, b_bytes);
    const rsrc_t crs = make_rsrc(C, c_bytes);
    // (col, val) through buffer descriptors as well: the reads of a row past the end of a lane group's run are DROPPED (bit 31),
    // not clamped -- clamped, every lane group of the XCD read the part's last row at once, one hot line of one L2 channel in
    // front of every wave's B reads (vmcnt counts in order): the first build ran 16 % slower than the kernel it was to replace
    const rsrc_t colrs = make_rsrc(colIdxs, a_bytes);
    const rsrc_t valrs = make_rsrc(vals, a_bytes);

    // A row's entries as they come from memory; what is done with them (offsets, padding, rows past the end) waits until the row
    // moves up in the pipeline -- any arithmetic on a loaded value makes hipcc wait for the load where the arithmetic stands.
    struct Raw {
        uint32_t col[E];
        float val[E];
        uint32_t orow;
    };
    auto load_raw = [&](uint32_t r, Raw &raw) {
        const bool live_row = r < row_end;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const uint32_t idx = static_cast<uint32_t>(e * G) + lane;
            const uint32_t pos4 = (live_row && idx < width) ? (r * width + idx) * 4u : kDropLoad;   // a_bytes < 2 GiB: no overflow
            raw.col[e] = __builtin_amdgcn_raw_buffer_load_b32(colrs, pos4, 0, 0);
            raw.val[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(valrs, pos4, 0, 0));
        }
        if constexpr (MAPPED) raw.orow = live_row ? rowMap[r] : 0u;
    };
    // branch-free on purpose, and every loaded value is USED whatever the predicates say: written as `drop ? constant : f(loaded)`
    // hipcc sinks the load into a conditional block at the point of use -- behind the whole row, with a full drain in front of it
    auto resolve = [&](uint32_t r, const Raw &raw, uint32_t (&off)[E], float (&val)[E], uint32_t &orow) {
        const bool live_row = r < row_end;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            uint32_t dropbit = (!live_row || static_cast<uint32_t>(e * G) + lane >= width) ? kDropLoad : 0u;
            if constexpr (PADDED) dropbit |= raw.col[e] == 0xFFFFFFFFu ? kDropLoad : 0u;
            off[e] = (raw.col[e] * ldb4) | dropbit;            // bit 31 set: out of range whatever the rest says
            val[e] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, raw.val[e]) & ((dropbit >> 31) - 1u));
        }
        if constexpr (MAPPED) orow = raw.orow;
        else orow = r;
    };
    // The ring: U reads in flight, entry i = (B values, coefficient).  The coefficients live in register PAIRS (entries 2k and
    // 2k + 1): the reference arithmetic multiplies with v_pk_mul_f32, whose broadcast operand is a 64-bit register pair of which
    // only one half is read -- given a lone register hipcc pairs it with ANY neighbour, and when that neighbour is the
    // destination of a (col, val) read still in flight the wave waits for it (vmcnt(0): a full drain per row, seen in the ISA).
    // With both halves of every pair holding a coefficient the operand never overlaps a register a load is writing.
    vec_t bv[U];
    f32x2 avp[(U + 1) / 2];
#pragma unroll
    for (int i = 0; i < (U + 1) / 2; ++i) avp[i] = f32x2{0.f, 0.f};
    auto issue = [&](auto slot_tag, auto ring_tag, const uint32_t (&off)[E], const float (&val)[E]) {
        constexpr int S = decltype(slot_tag)::value, R = decltype(ring_tag)::value;
        const uint32_t o = group_bcast<G, S % G>(off[S / G]);
        avp[R / 2][R % 2] = __builtin_bit_cast(float, group_bcast<G, S % G>(__builtin_bit_cast(uint32_t, val[S / G])));
        bv[R] = buffer_load_vec<VEC, MISPMM_B_LOAD_AUX>(brs, o + lane_off, 0);
    };
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    auto consume = [&](auto ring_tag) {
        constexpr int R = decltype(ring_tag)::value;
        const vec_t b = bv[R];
        if constexpr (std::is_same_v<Acc, AccRefWide>) {
            // fp32 products (two packed multiplies, IEEE rounding per element), widened, added to the double sums
            const f32x2 a2 = __builtin_shufflevector(avp[R / 2], avp[R / 2], R % 2, R % 2);
            const f32x2 p01 = a2 * f32x2{b[0], b[1]}, p23 = a2 * f32x2{b[2], b[3]};
            acc[0] += static_cast<double>(p01[0]);
            acc[1] += static_cast<double>(p01[1]);
            acc[2] += static_cast<double>(p23[0]);
            acc[3] += static_cast<double>(p23[1]);
        } else {
            const float a = avp[R / 2][R % 2];
#pragma unroll
            for (int v = 0; v < VEC; ++v) Acc::mac(acc[v], a, b[v]);
        }
    };

    // prologue: the first two rows' entries (one round trip for both), then the first U reads of row 0
    uint32_t off0[E], off1[E], orow0, orow1;
    float val0[E], val1[E];
    Raw raw, raw_b;
    load_raw(row, raw);
    load_raw(row + stride, raw_b);
    resolve(row, raw, off0, val0, orow0);
    resolve(row + stride, raw_b, off1, val1, orow1);
    // in ring order, pinned: the loop consumes entry 0 first, and hipcc counts its waits from the issue order it finds HERE --
    // left free it issued entry 0 sixth, and the merged count at the loop head (vmcnt(3)) drained the ring at every row start
    static_for<0, U>([&](auto s) {
        issue(s, s, off0, val0);
        __builtin_amdgcn_sched_barrier(0);
    });

    // One row of the walk.  NEXT: the row is followed by another one of this WAVE's run, whose first U reads are issued into the
    // ring as this row's last U entries free it.  The last row issues nothing behind its own reads: a dropped read is not free --
    // its broadcasts, its issue and its pass through the address unit cost what a live one's do (two dropped slots per row cost
    // the one-row kernel 9 %, row_gather.hpp; seven per run cost the first build of this kernel 0.6 us on the headline).
    auto walk_row = [&](auto next_tag) {
        constexpr bool NEXT = decltype(next_tag)::value;
        if constexpr (NEXT) load_raw(row + 2u * stride, raw);  // two rows ahead: it has a whole row's time to arrive
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, W>([&](auto s) {
            constexpr int S = decltype(s)::value;
            consume(std::integral_constant<int, S % U>{});
            if constexpr (S + U < W || NEXT) {
                // the refill may not be hoisted above the sums it waits for (see the rolling body of row_gather.hpp)
                asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (S + U < W) issue(std::integral_constant<int, S + U>{}, std::integral_constant<int, S % U>{}, off0, val0);
                else issue(std::integral_constant<int, S + U - W>{}, std::integral_constant<int, S % U>{}, off1, val1);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // row done: its C leaves through a non-temporal store; a row past the run's end (or a masked column) is an out-of-range,
        // i.e. dropped, store -- no branch
        vec_t out;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            out[v] = Acc::finish(acc[v]);
            acc[v] = 0;
        }
        const bool store_ok = row < row_end && col_ok;
        buffer_store_vec_c<VEC>(crs, store_ok ? (orow0 * ldc + col0) * 4u : kDropLoad, out);
        if constexpr (NEXT) {                                  // the pipeline moves up one row
#pragma unroll
            for (int e = 0; e < E; ++e) {
                off0[e] = off1[e];
                val0[e] = val1[e];
            }
            orow0 = orow1;
            resolve(row + 2u * stride, raw, off1, val1, orow1);
            row += stride;
        }
    };
    for (uint32_t it = 1; it < iters; ++it) walk_row(std::true_type{});
    walk_row(std::false_type{});
}

// ---- host side -------------------------------------------------------------------------------
// acc_kind: 0 = AccRefWide (CSR REFERENCE), 1 = AccRefF32 (ELL REFERENCE), 2 = AccFast.  Returns true if it launched.
// force: -1 = the library's rule (stream_pays), 0 = never, 1 = whenever the shape has an instance.
bool try_row_stream(const RowGatherArgs &a, uint32_t width, bool padded, int acc_kind, int vec, int force);

}  // namespace mispmm
