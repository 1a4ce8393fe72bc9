// CSR x dense for matrices with LONG or very uneven rows (kernel 6): a row's entries are dealt over the 8 lane groups of
// a wave that each keep a partial sum, and the partial sums are added in a fixed order at the end.
//
// Why: a row's sum is sequential by the reference's contract (spmm_csr.cpp:20-25: `acc += (float)(a * b)` in entry
// order into a double), so a kernel that keeps ONE running sum per output element needs (row length / reads in flight)
// memory round trips for its longest row -- GL7d25's 422-entry row alone is ~9 us on an otherwise idle GPU (DESIGN.md
// §5).  Splitting the row is a re-association of the sum.  In FAST mode that is allowed outright.  In REFERENCE mode it
// is allowed WHEN IT CANNOT CHANGE THE RESULT, which is decided per lane (4 output elements) from two numbers tracked
// alongside the sums:
//   every product p is an fp32 value, i.e. an integer multiple of ulp32(p) > 2^-24 |p|.  With u = 2^-24 min|p| over the
//   non-zero products of the row, every partial sum in ANY order is a multiple of u no larger than sum|p| <= L max|p|.
//   If L max|p| <= 2^29 min|p| every such partial sum is below 2^53 u and therefore exactly representable in fp64: no
//   addition rounds, in any order, so the split sum and the reference's sequential sum are the same double, and the
//   final rounding to fp32 sees the same value.  (2^28 is used: one bit of slack for evaluating the test in fp32.)
// Zero products are exact in any sum and are left out of min|p|.  A wave with an element that fails the test (a product
// 2^19.. times smaller than the largest of a 400-entry row, an Inf / NaN, a huge value) sums its 32 columns again IN
// ENTRY ORDER: the same pass with the same 64 entries in flight, but each pair of steps lists its 16 x 32 products in
// LDS and lanes 0..31 add their column's 16 terms in order -- bit-exact always, fast when the data allows.  (Measured on
// the way: a thread that walks the row on its own pays two dependent memory round trips per entry -- ONE such element
// of a 170-entry row took GL7d25 from 11.7 to 35 us; a wave re-reading the row one entry per step, 26 us.)
//
// Shape: one WAVE per row x 32 output columns: 8 lanes x 16 bytes read one 128-byte segment of a B row, a wave
// multiplies 8 entries per step, and the 32-column parts of a row go to DIFFERENT XCDs (P row parts x Q column parts
// over the 8 XCDs, Q = up to 8).  Each XCD's L2 then only ever sees its own 128-byte columns of B.  On a matrix whose
// rows scatter over all of B (GL7d25: 29 entries per row over 21 074 columns) a workgroup-per-row variant reading
// whole 512-byte B rows made every XCD fetch 4-8 MB against its 4 MB of L2, time proportional to N (N = 128 / 256:
// 10.9 / 21.3 us FAST); this shape fetches 21 074 x 128 bytes per XCD, which fits (6.9 / 9.8 us).
// Everything is wave-private (strip, partial sums, ordered re-sum): a wave leaves when its row is done.
//
// Order: the stateless entry point walks the rows in row order.  mispmm_csr_split_f32 walks a span list built once per
// matrix on the host -- rows by decreasing length, so the long ones start first (GL7d25 is sorted the other way round:
// in row order its long rows all start last and decide when the kernel ends: 9.9 against 6.9 us), and every row of more
// than 128 entries as 4 chunks on the 4 waves of one workgroup, which hand their chunk's sum and extremes over in LDS
// (the kernel's one barrier) for wave 0 to add in entry order and test as a whole row.
#pragma once
#include "spmm_common.hpp"

namespace mispmm {

// max|p| and min non-zero |p| of the products a lane has added (REFERENCE mode only).  The minimum is kept as
// 2 * bits(|p|) - 1 (unsigned): monotone in |p|, and a zero product wraps to 0xFFFFFFFF so it never wins.
struct ExactTrack {
    float hi = 0.f;
    uint32_t lo = 0xFFFFFFFFu;
    __device__ __forceinline__ void add2(float p0, float p1) {
        hi = __builtin_fmaxf(hi, __builtin_fmaxf(__builtin_fabsf(p0), __builtin_fabsf(p1)));
        const uint32_t t0 = (__float_as_uint(p0) << 1) + 0xFFFFFFFFu, t1 = (__float_as_uint(p1) << 1) + 0xFFFFFFFFu;
        lo = min(lo, min(t0, t1));
    }
};

// true when the fp64 sum of `len` fp32 products with the given extremes is exact in every order (see the header)
__device__ __forceinline__ bool reassociation_is_exact(uint32_t len, float hi, uint32_t lo) {
    if (lo == 0xFFFFFFFFu) return hi == 0.f;  // only zero products (or none)
    const float smallest = __uint_as_float((lo + 1u) >> 1);
    return len < (1u << 24) && hi < 0x1p100f && static_cast<float>(len) * hi <= 0x1p28f * smallest;
}

#ifdef MISPMM_STAMPS
// Diagnostic build only (tools/stamp_split.py): every wave of the split body leaves s_memrealtime stamps (100 MHz) here:
// 0 start, 1 span read, 2 split pass done, 3 partial sums reduced, 4 chunk hand-over barrier passed, 5 before the store
// (after the chunk loop / an ordered pass), 6 end; word 7 = row length | shared << 32 | took the ordered pass << 33.
static __device__ unsigned long long *mispmm_split_stamp_buf = nullptr;
#define MISPMM_SPLIT_STAMP(i) sstamp[i] = wall_clock64()
#else
#define MISPMM_SPLIT_STAMP(i)
#endif

#ifdef MISPMM_TUNING
// measurement build only: [0] waves that summed their row again in entry order since the last reset
__device__ unsigned long long mispmm_split_stats[2];
#endif

struct SplitTiling {
    uint32_t p, q;           // row parts x column parts over the XCDs, p * q == 8
    uint32_t rows_per_part;  // rows of a row part, ceil(M / p); with a span list: its workgroups
    uint32_t grid_x, grid_y;
};

// M: rows, or -- with a span list -- its positions, dealt to the row parts in groups of `waves` (one workgroup each)
inline SplitTiling split_tiling(uint32_t M, uint32_t N, uint32_t waves, bool spans) {
    const uint32_t colparts = ceil_div(N, 32u);
    uint32_t q = 1;
    while (q < 8u && q < colparts) q <<= 1;
    SplitTiling t;
    t.q = q;
    t.p = 8u / q;
    if (spans) {
        t.rows_per_part = ceil_div(ceil_div(M, waves), t.p);  // workgroups per row part
        t.grid_x = 8u * t.rows_per_part;
    } else {
        t.rows_per_part = ceil_div(M, t.p);
        t.grid_x = 8u * ceil_div(t.rows_per_part, waves);
    }  // a multiple of 8: workgroup id % 8 == blockIdx.x % 8 in every grid row
    t.grid_y = ceil_div(colparts, q);
    return t;
}

// the kernel's body; (bx, by) = the workgroup's position in a grid of its own (csr_hybrid runs it on the first workgroups
// of a launch whose other workgroups take the short rows through the row-gather body)
template <class Acc, int WAVES, int NB>
__device__ __forceinline__ void csr_split_body(const uint32_t bx, const uint32_t by, uint32_t M, const uint32_t *__restrict__ rowPtrs,
                                               const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                               const float *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                               float *__restrict__ C, uint32_t ldc, uint32_t tile_q, uint32_t rows_per_part,
                                               const uint32_t *__restrict__ spans) {
#ifdef MISPMM_STAMPS
    unsigned long long sstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    MISPMM_SPLIT_STAMP(0);
    using u2 = uint32_t __attribute__((ext_vector_type(2)));
    using T = typename Acc::T;
    constexpr bool kRef = std::is_same_v<Acc, AccRefWide>;
    // COO / ELL / BSR arithmetic (fp32 product, fp32 add): no sum of theirs may be split, the row goes through the
    // ordered pass straight away -- the same shape, order and reads in flight, one running fp32 sum per output element
    constexpr bool kSequential = std::is_same_v<Acc, AccRefF32>;
    using OrderedT = std::conditional_t<kSequential, float, double>;
    constexpr int G = 8;          // lanes per entry: 8 x 4 columns = one 128-byte segment of a B row
    constexpr int OWN = 8;        // entries per step = lane groups = partial sums per output element
    constexpr int COLS = 32;
    constexpr int RING = NB * 4;  // B-segment reads in flight per lane, refilled in blocks of 4 steps
    constexpr int PHASE = 256;    // entries staged in LDS at a time
    static_assert(PHASE % (RING * OWN) == 0, "staging is padded to whole rings");
    // per wave: the staged entries | the partial sums (8 groups x 32 columns), reused by the ordered re-sum as the list
    // of the products of two steps (16 entries x 32 columns, fp32)
    constexpr int kStripBytes = PHASE * 8;
    constexpr int kPartBytes = OWN * COLS * static_cast<int>(sizeof(T)) > 16 * COLS * 4 ? OWN * COLS * static_cast<int>(sizeof(T)) : 16 * COLS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[WAVES][kStripBytes + kPartBytes];
    __shared__ float track_hi_all[kRef ? WAVES * OWN * G : 1];
    __shared__ uint32_t track_lo_all[kRef ? WAVES * OWN * G : 1];

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    // workgroups are dealt round-robin over the XCDs: blockIdx.x & 7 is the XCD, which owns one (row part, column part).
    // Placement only: any dispatch order gives the same result.
    const uint32_t xcd = bx & 7u;   // (blockIdx.x of the split grid)
    const uint32_t part_row = xcd / tile_q, part_col = xcd % tile_q;
    // Without spans: row part k is the k-th contiguous range of rows, walked in row order.  With spans -- the rows as
    // (row, start, end, info) sorted by decreasing length, built once per matrix on the host -- the groups of WAVES
    // positions (one workgroup each) are dealt to the row parts in turn: the longest rows start first and every XCD gets
    // an equal share of them.  (GL7d25 is sorted the other way round, every row longer than 128 entries among its last
    // 93: in row order they all start last and decide when the kernel ends.)
    const uint32_t block = bx >> 3;
    const uint32_t position = spans ? (block * (8u / tile_q) + part_row) * WAVES + wave : part_row * rows_per_part + block * WAVES + wave;
    const uint32_t slab = (by * tile_q + part_col) * COLS;
    // wave-uniform; the one workgroup barrier below is reached by the 4 waves of a shared row, which are all valid
    if ((spans ? block : block * WAVES + wave) >= rows_per_part || position >= M || slab >= N) return;
    const uint32_t li = lane % G;
    const uint32_t group = lane / G;
    const uint32_t col0 = slab + li * 4u;
    const bool lane_live = col0 < N;  // lanes past the last column never fetch
    const uint32_t lane_off = lane_live ? col0 * 4u : kDropLoad;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t ldb4 = ldb * 4u;
    uint32_t row, start, end;
    bool shared = false;  // this wave holds one of the WAVES chunks of a long row (workgroup-uniform)
    if (spans) {  // kernel-uniform
        using u4 = uint32_t __attribute__((ext_vector_type(4)));
        const u4 span = *reinterpret_cast<const u4 *>(spans + static_cast<size_t>(position) * 4u);
        row = __builtin_amdgcn_readfirstlane(span[0]);
        start = __builtin_amdgcn_readfirstlane(span[1]);
        end = __builtin_amdgcn_readfirstlane(span[2]);
        shared = __builtin_amdgcn_readfirstlane(span[3]) != 0u;
    } else {
        row = position;
        start = __builtin_amdgcn_readfirstlane(rowPtrs[row]);
        end = __builtin_amdgcn_readfirstlane(rowPtrs[row + 1]);
    }
    MISPMM_SPLIT_STAMP(1);
#ifdef MISPMM_STAMPS
    const uint32_t stamp_len = end - start;
    auto stamp_dump = [&](bool took_ordered) {
        if (mispmm_split_stamp_buf && (threadIdx.x & 63u) == 0) {
            sstamp[7] = static_cast<unsigned long long>(stamp_len) | (static_cast<unsigned long long>(shared) << 32) |
                        (static_cast<unsigned long long>(took_ordered) << 33);
            unsigned long long *o = mispmm_split_stamp_buf + (static_cast<size_t>(bx) * WAVES + wave) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = sstamp[i];
        }
    };
#endif
    u2 *const strip = reinterpret_cast<u2 *>(smem[wave]);
    T *const part = reinterpret_cast<T *>(smem[wave] + kStripBytes);
    float *const listed = reinterpret_cast<float *>(smem[wave] + kStripBytes);
    float *const track_hi = track_hi_all + (kRef ? wave * OWN * G : 0);
    uint32_t *const track_lo = track_lo_all + (kRef ? wave * OWN * G : 0);
    // a wave's LDS is its own and a wave's LDS operations complete in order; this keeps the compiler from moving reads
    // ahead of the writes they depend on
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // the same between LDS operations only, for use while B reads are in flight: a fence would wait for those too
    // (measured: the ordered re-sum of a 392-entry row 7.6 us with fences, every list waiting out the whole ring)
    auto lds_order = [] {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    T acc[4] = {0, 0, 0, 0};
    ExactTrack track;
    bool staged_whole = true;
    OrderedT ordered = 0;  // ordered pass: the running sum of column `lane` (lanes 0..31)
    // One pass over the row, entry i read by lane group i % 8 in step i / 8.  kOrdered == false: every group adds its
    // products to its own partial sums.  kOrdered == true: the products go through LDS and are added in entry order.
    auto sweep = [&](auto ordered_tag) {
        constexpr bool kOrdered = decltype(ordered_tag)::value;
        for (uint32_t base = start; base < end; base += PHASE) {
            const uint32_t n = min(static_cast<uint32_t>(PHASE), end - base);
            const uint32_t steps = (n + OWN - 1u) / OWN;
            const uint32_t nring = (steps + RING - 1u) / RING;
            wave_sync();  // the previous phase / the partial sums have been read
            // (byte offset of the B row, coefficient); the padding up to whole rings is (dropped load, 0): exact no-ops.
            // A row of one phase that this wave staged whole is still in the strip when the ordered pass comes round.
            if (!kOrdered || end - start > static_cast<uint32_t>(PHASE) || !staged_whole) {
                for (uint32_t i = lane; i < nring * (RING * OWN); i += 64u) {
                    u2 pair{kDropLoad, 0u};
                    if (i < n) {
                        pair[0] = colIdxs[base + i] * ldb4;
                        pair[1] = __float_as_uint(vals[base + i]);
                    }
                    strip[i] = pair;
                }
                wave_sync();
            }

            f32x4 bv[RING];
            float av[RING];
            auto issue_block = [&](uint32_t blk, auto ring_tag) {
                constexpr int R = decltype(ring_tag)::value;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u2 pair = strip[(blk * 4u + t) * OWN + group];  // one address per lane group: LDS broadcasts
                    av[R * 4 + t] = __uint_as_float(pair[1]);
                    // (a padding entry in a dead lane: kDropLoad + kDropLoad would wrap to offset 0, a real read of B[0] whose
                    // Inf / NaN would reach the unused partial sums and their exactness trackers -- keep the drop bit)
                    bv[R * 4 + t] = buffer_load_vec<4>(rsrc, lane_live ? pair[0] + lane_off : kDropLoad, 0);
                }
            };
            auto consume_block = [&](auto ring_tag) {
                constexpr int R = decltype(ring_tag)::value;
                if constexpr (kOrdered) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        lds_order();  // the previous list has been added up
#pragma unroll
                        for (int t = 2 * half; t < 2 * half + 2; ++t) {
                            const f32x4 b = bv[R * 4 + t];
                            const float a = av[R * 4 + t];
                            // entry (step, group) -> list row (step % 2) * 8 + group: entry order
                            *reinterpret_cast<f32x4 *>(listed + ((t & 1) * OWN + group) * COLS + li * 4u) =
                                f32x4{a * b[0], a * b[1], a * b[2], a * b[3]};
                        }
                        lds_order();
                        if (lane < COLS) {
                            float term[16];
#pragma unroll
                            for (int k = 0; k < 16; ++k) term[k] = listed[k * COLS + lane];
                            // past the row's end the products are +0.0 (dropped load x 0), and sum + 0.0 == sum: the
                            // running sum starts at +0.0 and so is never -0.0
#pragma unroll
                            for (int k = 0; k < 16; ++k) ordered = ordered + static_cast<OrderedT>(term[k]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 b = bv[R * 4 + t];
                        if constexpr (kRef) {
                            using f2 = float __attribute__((ext_vector_type(2)));
                            const f2 a2{av[R * 4 + t], av[R * 4 + t]};
                            const f2 p01 = a2 * f2{b[0], b[1]}, p23 = a2 * f2{b[2], b[3]};
                            acc[0] += static_cast<double>(p01[0]);
                            acc[1] += static_cast<double>(p01[1]);
                            acc[2] += static_cast<double>(p23[0]);
                            acc[3] += static_cast<double>(p23[1]);
                            track.add2(p01[0], p01[1]);
                            track.add2(p23[0], p23[1]);
                        } else {
#pragma unroll
                            for (int v = 0; v < 4; ++v) Acc::mac(acc[v], av[R * 4 + t], b[v]);
                        }
                    }
                }
            };
            auto pin = [&] {  // a refill reuses the registers just consumed: keep it behind the sums (see row_gather.hpp)
                if constexpr (kOrdered) asm volatile("" : "+v"(ordered) : : "memory");
                else asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
            };
            static_for<0, NB>([&](auto r) { issue_block(decltype(r)::value, r); });
            uint32_t b0 = 0;
            for (; b0 + NB < nring * NB; b0 += NB) {  // steady state: RING reads in flight, each block refilled as consumed
                static_for<0, NB>([&](auto r) {
                    consume_block(r);
                    pin();
                    __builtin_amdgcn_sched_barrier(0);
                    issue_block(b0 + NB + decltype(r)::value, r);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // the last ring: blocks past the row's end hold only padding, skip their arithmetic
            static_for<0, NB>([&](auto r) {
                if ((b0 + decltype(r)::value) * 4u < steps) consume_block(r);
            });
        }
    };
    if constexpr (kSequential) {
        staged_whole = false;
        sweep(std::true_type{});
        if (lane < COLS && slab + lane < N) C[static_cast<size_t>(row) * ldc + slab + lane] = ordered;
        return;
    }
    sweep(std::false_type{});
#ifdef MISPMM_STAMPS
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
#endif
    MISPMM_SPLIT_STAMP(2);

    // partial sums -> LDS, then lanes 0..31 add their column's 8 partial sums in group order
    wave_sync();
#pragma unroll
    for (int v = 0; v < 4; ++v) part[group * COLS + li * 4u + v] = acc[v];
    if constexpr (kRef) {
        track_hi[group * G + li] = track.hi;
        track_lo[group * G + li] = track.lo;
    }
    wave_sync();
    const bool mine = lane < COLS && slab + lane < N;
    T total = 0;
    float hi = 0.f;
    uint32_t lo = 0xFFFFFFFFu;
    if (lane < COLS) {
        total = part[lane];
#pragma unroll
        for (int o = 1; o < OWN; ++o) total += part[o * COLS + lane];
        if constexpr (kRef) {
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                hi = __builtin_fmaxf(hi, track_hi[o * G + lane / 4u]);
                lo = min(lo, track_lo[o * G + lane / 4u]);
            }
        }
    }
    MISPMM_SPLIT_STAMP(3);
    bool need_ordered = false;
    if (shared) {
        // A long row as WAVES chunks, one per wave of this workgroup (the span list places them so): each wave hands its
        // chunk's (sum, extremes, bounds) over in its own LDS region, and after the one barrier wave 0 adds the chunks in
        // entry order -- the row that decides when the kernel ends takes a quarter of its memory round trips.
        constexpr int kSums = 0, kHi = COLS * static_cast<int>(sizeof(T)), kLo = kHi + COLS * 4, kBounds = kLo + COLS * 4;
        static_assert(kBounds + 8 <= kPartBytes, "the hand-over fits the partial-sum region");
        wave_sync();  // the partial sums have been read
        unsigned char *const hand = smem[wave] + kStripBytes;
        if (lane < COLS) {
            reinterpret_cast<T *>(hand + kSums)[lane] = total;
            reinterpret_cast<float *>(hand + kHi)[lane] = hi;
            reinterpret_cast<uint32_t *>(hand + kLo)[lane] = lo;
        }
        if (lane == 0) {
            reinterpret_cast<uint32_t *>(hand + kBounds)[0] = start;
            reinterpret_cast<uint32_t *>(hand + kBounds)[1] = end;
        }
        __syncthreads();
        MISPMM_SPLIT_STAMP(4);
        if (wave != 0) {
#ifdef MISPMM_STAMPS
            sstamp[5] = sstamp[6] = sstamp[4];
            stamp_dump(false);
#endif
            return;
        }
        // Chunk by chunk: as long as the union of the chunks so far passes the test, their sum is exact whatever the
        // order -- it IS the reference's running sum at that point -- so the ordered pass, if one is needed, starts at the
        // first chunk that breaks the test, from that sum, instead of at the row's first entry.
        total = 0;
        hi = 0.f;
        lo = 0xFFFFFFFFu;
        bool broken = false;
        uint32_t resume = start;
        // (Tried on the stamps' evidence -- 0.9-1.0 us between the barrier and the store of a 4-chunk row -- and not kept: the
        // hand-overs read in one batch of LDS reads (no change), and the whole row tested at once before this walk (that
        // segment 0.92 -> 0.60 us at p90, the launch unchanged: what ends it are the shortest rows of the split body, dispatched
        // last and reading under the load of the row-gather body; profiles/r3/stamps_split.log.)
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {  // wave order = entry order of the chunks
            const unsigned char *const from = smem[w] + kStripBytes;
            const uint32_t chunk_start = __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(from + kBounds)[0]);
            const uint32_t chunk_end = __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(from + kBounds)[1]);
            if (!broken) {  // wave-uniform
                T sum = total;
                float h = hi;
                uint32_t l = lo;
                if (lane < COLS) {
                    sum += reinterpret_cast<const T *>(from + kSums)[lane];  // 0 + x == x: the first term is exact
                    h = __builtin_fmaxf(h, reinterpret_cast<const float *>(from + kHi)[lane]);
                    l = min(l, reinterpret_cast<const uint32_t *>(from + kLo)[lane]);
                }
                if constexpr (kRef) {
                    // NaN products do not reach `hi` (fmax drops them) but do reach the sum
                    const bool bad = mine && (!reassociation_is_exact(chunk_end - start, h, l) || sum != sum);
                    if (__ballot(bad) != 0) {
                        broken = true;
                        resume = chunk_start;
                    }
                }
                if (!broken) {
                    total = sum;
                    hi = h;
                    lo = l;
                }
            }
            end = chunk_end;  // after the loop: the row's end
        }
        if (broken) {
            ordered = total;  // lanes 0..31: the exact sum of the entries before `resume`
            start = resume;
            staged_whole = false;
            need_ordered = true;
        }
    } else if constexpr (kRef) {
        // NaN products do not reach `hi` (fmax drops them) but do reach the sum
        need_ordered = __ballot(mine && (!reassociation_is_exact(end - start, hi, lo) || total != total)) != 0;  // wave-uniform
    }
    if constexpr (kRef) {
        if (need_ordered) {
#ifdef MISPMM_TUNING
            if (lane == 0) atomicAdd(&mispmm_split_stats[0], 1ull);
#endif
            sweep(std::true_type{});
            total = ordered;
        }
    }
#ifdef MISPMM_STAMPS
    if (!shared) sstamp[4] = sstamp[3];
    asm volatile("" : "+v"(total) : : "memory");
#endif
    MISPMM_SPLIT_STAMP(5);
    if (mine) C[static_cast<size_t>(row) * ldc + slab + lane] = Acc::finish(total);
#ifdef MISPMM_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MISPMM_SPLIT_STAMP(6);
    stamp_dump(need_ordered);
#endif
}

template <class Acc, int WAVES, int NB>
__global__ __launch_bounds__(WAVES * 64) void csr_split(uint32_t M, const uint32_t *__restrict__ rowPtrs,
                                                        const uint32_t *__restrict__ colIdxs, const float *__restrict__ vals,
                                                        const float *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                                        float *__restrict__ C, uint32_t ldc, uint32_t tile_q,
                                                        uint32_t rows_per_part, const uint32_t *__restrict__ spans) {
    csr_split_body<Acc, WAVES, NB>(blockIdx.x, blockIdx.y, M, rowPtrs, colIdxs, vals, B, b_bytes, N, ldb, C, ldc, tile_q, rows_per_part, spans);
}

// ---- host side: what the CSR, COO and BSR-list entry points launch ---------------------------------------------------
struct SplitArgs {
    hipStream_t stream;
    uint32_t M, K;
    const uint32_t *rowPtrs, *colIdxs;
    const float *vals, *B;
    uint32_t N, ldb;
    float *C;
    uint32_t ldc;
    const uint32_t *spans = nullptr;  // the span list (rows longest first, long rows as 4 chunks), or rows in order
    uint32_t numSpans = 0;
};

template <class Acc, int WAVES, int NB>
inline void launch_split_as(const SplitArgs &a) {
    const SplitTiling t = split_tiling(a.spans ? a.numSpans : a.M, a.N, WAVES, a.spans != nullptr);
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u);
    note_kernel("csr_split<W%d,R%d,%s%s> xcd %ux%u", WAVES, NB * 4, acc_tag<Acc>(), a.spans ? ",longest-first" : "", t.p, t.q);
    hipLaunchKernelGGL((csr_split<Acc, WAVES, NB>), dim3(t.grid_x, t.grid_y), dim3(WAVES * 64), 0, a.stream,
                       a.spans ? a.numSpans : a.M, a.rowPtrs, a.colIdxs, a.vals, a.B, b_bytes, a.N, a.ldb, a.C, a.ldc, t.q,
                       t.rows_per_part, a.spans);
}

// one wave per row x 32 columns, the row's entries dealt over its 8 lane groups.  GL7d25, us REFERENCE / FAST at N = 128
// with 4 / 8 / 16 reads in flight per lane: 10.4 / 6.8, 9.9 / 6.9, 12.2 / 9.5; 1, 2 or 4 waves per workgroup make no
// difference.  Needs B and C rows of 16-byte vectors and a B below 2 GiB (the callers check).
template <class Acc>
inline void launch_split(const SplitArgs &a) {
    static const int ring = knob_int("MISPMM_SPLIT_RING", 8);
    if (ring == 4) launch_split_as<Acc, 4, 1>(a);
    else if (ring == 16) launch_split_as<Acc, 4, 4>(a);
    else launch_split_as<Acc, 4, 2>(a);
}

}  // namespace mispmm
