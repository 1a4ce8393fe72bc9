// BSR x dense SpMM for gfx950.  Blocks are bR x bC row-major, stored in block-CSR order.
//
// Replaces /root/reference/src/spmm/bsr/spmm_bsr_k1.cu (one thread per block element, an
// atomicAdd per output element and term).  Three kernels, none with atomics:
//   bsr_rowblock  square blocks of 4 / 8 / 16 / 32 (kernel 1's fast path): a wave owns a block row,
//             keeps its BD x VEC partial sums in registers, fetches each B row once per block row.
//   bsr_valu  any block shape.  A C row is a CSR row whose terms are (block, column-in-block)
//             pairs: G lanes own one C row, the coefficients of 16 terms are fetched by one
//             coalesced load and broadcast by shuffle, B rows come as dropped-or-live buffer
//             loads.  Terms are summed block by block in storage order and by ascending column
//             inside a block: REFERENCE mode reproduces spmmBSRCpu (spmm_bsr.cpp:17-38) bit for bit.
//   bsr_mfma_f32   16x16 blocks on v_mfma_f32_16x16x4_f32: one wave owns a block row x 64 output
//             columns (4 accumulator tiles).  The instruction is an exact k-ordered fp32 fma chain,
//             and the k slots are mapped to ascending block columns, so the result equals the
//             FAST VALU kernel's bit for bit.
//   bsr_mfma_bf16  16x16 bf16 blocks on v_mfma_f32_16x16x32_bf16, two blocks per instruction
//             (K = 32), fp32 accumulate, B transposed into operand order in registers (v_perm);
//             the 4 waves of a workgroup split a block row's pairs and reduce through LDS.
// Output columns of a 64-wide super-tile are interleaved over the 4 accumulator tiles
// (tile t, lane column c <-> column 4c + t), so B rows are read and C rows written as whole
// 16-byte (fp32) / 8-byte (bf16) vectors.
// Roofline: HBM for bf16 (algorithmic bytes nb*bR*bC*e + nb*4 + (Mb+1)*4 + K*N*e + M*N*o);
// fp32 MFMA runs at the fp32 vector rate and is MFMA-bound on low-fill blocks.
#include <cstdlib>

#include "bsr_bf16_lds.hpp"
#include "bsr_slots.hpp"
#include "spmm_common.hpp"

namespace mispmm {

using f32x4_t = float __attribute__((ext_vector_type(4)));
using bf16x8_t = short __attribute__((ext_vector_type(8)));
using u32x2_t = uint32_t __attribute__((ext_vector_type(2)));
using u32x4_t = uint32_t __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------- bsr_valu
template <int G, int VEC, class Acc, bool WIDE>
__global__ __launch_bounds__(256) void bsr_valu(uint32_t M, uint32_t bR, uint32_t bC,
                                                const uint32_t *__restrict__ blockRowPtrs,
                                                const uint32_t *__restrict__ blockColIdxs,
                                                const float *__restrict__ blocks, const float *__restrict__ B,
                                                uint32_t b_bytes, uint32_t N, uint32_t ldb, float *__restrict__ C,
                                                uint32_t ldc) {
    constexpr int GROUPS = 256 / G;
    constexpr int U = (VEC == 4) ? 8 : 16;
    using vec_t = typename VecOf<VEC>::type;
    const uint32_t lane = threadIdx.x % G;
    const uint32_t row = blockIdx.x * GROUPS + threadIdx.x / G;
    const uint32_t col0 = blockIdx.y * (G * VEC) + lane * VEC;
    const bool row_ok = row < M;
    const bool col_ok = col0 < N;
    uint32_t bstart = 0, terms = 0, i_in = 0;
    if (row_ok) {
        const uint32_t R = row / bR;
        i_in = row - R * bR;
        bstart = blockRowPtrs[R];
        terms = (blockRowPtrs[R + 1] - bstart) * bC;
    }
    typename Acc::T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;
    const float *bcol = B + (col_ok ? col0 : 0);

    for (uint32_t base = 0; base < terms; base += G) {
        const uint32_t cnt = min(static_cast<uint32_t>(G), terms - base);
        const uint32_t t = base + min(lane, cnt - 1);  // term index inside the block row
        const uint32_t b = bstart + t / bC;
        const uint32_t j = t % bC;
        const uint32_t my_brow = blockColIdxs[b] * bC + j;  // B row of this term
        const float my_val = blocks[(static_cast<size_t>(b) * bR + i_in) * bC + j];
        for (uint32_t q = 0; q < cnt; q += U) {
            vec_t bv[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t src = (q + u) & (G - 1);
                const float a = __shfl(my_val, src, G);
                const bool live = q + u < cnt;
                if constexpr (WIDE) {
                    const uint32_t r = __shfl(my_brow, live ? src : (cnt - 1) & (G - 1), G);
                    bv[u] = load_vec<VEC>(bcol + static_cast<size_t>(r) * ldb);
                    av[u] = a;
                } else {
                    const uint32_t off = __shfl(my_brow, src, G) * (ldb * 4u);
                    bv[u] = buffer_load_vec<VEC>(rsrc, live ? off + lane_off : kDropLoad, 0);
                    av[u] = live ? a : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!WIDE || q + u < cnt) {
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

// ---------------------------------------------------------------------------------- bsr_rowblock
// Square blocks of 4, 8, 16 or 32: one wave owns RS = min(BD, 8) rows of a block row for 64 * VEC
// columns and keeps its RS x VEC partial sums per lane in registers, so every B row of a block is
// fetched once per RS rows instead of once per C row (bsr_valu moves RS x the bytes through L1); the
// BD / RS row slices of a block row run as neighbouring waves (enough waves to fill the chip: the
// kernel is bound by the unfused multiply + add stream, 2 x 8.5 M terms x N lane-ops).
// Each block is staged in the wave's own LDS slice by one coalesced load per 256 values and its
// coefficients come back as broadcast ds_reads; the terms of a C element are still added block by
// block in storage order and by ascending column inside a block, with the accumulate policy's
// rounding -- REFERENCE mode stays bit-identical to spmmBSRCpu.  Wave-private LDS: no barrier.
template <int BD, int RS, int VEC, class Acc>
__global__ __launch_bounds__(256) void bsr_rowblock(uint32_t Mb, uint32_t nCT, const uint32_t *__restrict__ blockRowPtrs,
                                                    const uint32_t *__restrict__ blockColIdxs,
                                                    const float *__restrict__ blocks, const float *__restrict__ B,
                                                    uint32_t b_bytes, uint32_t N, uint32_t ldb, float *__restrict__ C,
                                                    uint32_t ldc, uint32_t xcd_chunk) {
    using vec_t = typename VecOf<VEC>::type;
    constexpr int PARTS = BD / RS;            // row slices of a block row, one wave each
    constexpr int SUB = RS * BD;              // block values a wave needs: its RS rows
    constexpr int PER_LANE = (SUB + 63) / 64;  // values each lane stages
    __shared__ float stage[4][2][SUB < 64 ? 64 : SUB];  // [wave][double buffer][RS x BD values]
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = xcd_block(blockIdx.x, xcd_chunk) * 4 + wave;
    if (item >= Mb * nCT * PARTS) return;  // wave-uniform; no workgroup barrier anywhere in this kernel
    const uint32_t part = item % PARTS, rest = item / PARTS;  // row slices of one block row sit on one CU: shared B rows hit L1
    const uint32_t R = rest / nCT, ct = rest - R * nCT;
    const uint32_t col0 = ct * (64 * VEC) + lane * VEC;
    const bool col_ok = col0 < N;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;
    const uint32_t ldb4 = ldb * 4u;

    typename Acc::T acc[RS][VEC];
#pragma unroll
    for (int i = 0; i < RS; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[i][v] = 0;

    const uint32_t bs = blockRowPtrs[R], be = blockRowPtrs[R + 1];
    float areg[PER_LANE];
    auto load_block = [&](uint32_t b) {  // coalesced: lane l takes values l, l + 64, ... of the wave's RS rows
        const float *sub = blocks + static_cast<size_t>(b) * (BD * BD) + part * SUB;
#pragma unroll
        for (int p = 0; p < PER_LANE; ++p) {
            const uint32_t idx = p * 64 + lane;
            areg[p] = idx < SUB ? sub[idx] : 0.f;
        }
    };
    // B rows arrive one step (4 block columns) ahead of the multiply-adds that use them, across block boundaries:
    // with the reads issued inside the step that consumed them a wave paid one L2 round trip per 4 columns
    // (config 4a: 86.9 -> 84.4 us; the block values through the scalar cache instead of LDS: 96.5 us.  What is left is
    // imbalance: 2500 waves, 8 to 54 blocks each, 2.4 per SIMD, against a 28 us floor of the multiply + add stream).
    auto issue_b = [&](uint32_t brow, uint32_t off_or_drop, vec_t (&dst)[4]) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dst[jj] = buffer_load_vec<VEC>(rsrc, off_or_drop, (brow + jj) * ldb4);
    };
    vec_t cur[4];
    uint32_t brow_next = 0;
    if (bs < be) {
        load_block(bs);
        brow_next = blockColIdxs[bs] * BD;
        issue_b(brow_next, lane_off, cur);
    }
    for (uint32_t b = bs; b < be; ++b) {
        float *slot = stage[wave][(b - bs) & 1];
#pragma unroll
        for (int p = 0; p < PER_LANE; ++p) {
            const uint32_t idx = p * 64 + lane;
            if (idx < SUB) slot[idx] = areg[p];
        }
        const uint32_t brow0 = brow_next;
        const bool more = b + 1 < be;  // wave-uniform
        brow_next = more ? blockColIdxs[b + 1] * BD : 0u;
        if (more) load_block(b + 1);  // next block's values fly while this one is multiplied
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // ds_writes of this wave before its ds_reads
#pragma unroll
        for (int j0 = 0; j0 < BD; j0 += 4) {
            vec_t nxt[4];
            if (j0 + 4 < BD) issue_b(brow0 + j0 + 4, lane_off, nxt);
            else issue_b(brow_next, more ? lane_off : kDropLoad, nxt);  // first step of the next block (nothing past the row)
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                const f32x4 a4 = *reinterpret_cast<const f32x4 *>(slot + i * BD + j0);  // broadcast read
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) Acc::mac(acc[i][v], a4[jj], vec_get<VEC>(cur[jj], v));
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) cur[jj] = nxt[jj];
        }
    }
    if (col_ok) {
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            vec_t out;
#pragma unroll
            for (int v = 0; v < VEC; ++v) vec_set<VEC>(out, v, Acc::finish(acc[i][v]));
            store_vec<VEC>(C + static_cast<size_t>(R * BD + part * RS + i) * ldc + col0, out);
        }
    }
}

// ---------------------------------------------------------------------------------- bsr_mfma_f32
// Work item = (block row R, 64-column super-tile st) = one WORKGROUP; its 4 waves take the block row's blocks
// round-robin (block bs + wave, + 4, ...), each keeps a partial 16 x 64 tile and the partials are summed through
// LDS in fixed wave order (deterministic).  With one wave per block row the longest block row (54 blocks on
// ACTIVSg10K) was one sequential chain of index -> B-row round trips: 65 us against a 16 us MFMA floor.
// Lane l = (c = l & 15, g = l >> 4).  MFMA 16x16x4: A operand lane holds A[i = c][k = g],
// B operand lane holds B[k = g][j = c]; D register r of lane l is D[row 4g + r][col c].
// Step s of a block uses block columns 4s + g, so k runs over ascending columns.
// Each wave's loop is software-pipelined two blocks deep: while block b is on the matrix pipe the A/B
// fragments of its next block are in flight and the block column of the one after is being fetched.
struct F32Frag {
    float a[4];
    f32x4_t bv[4];
};

__global__ __launch_bounds__(256) void bsr_mfma_f32(uint32_t Mb, uint32_t nST, const uint32_t *__restrict__ blockRowPtrs,
                                                    const uint32_t *__restrict__ blockColIdxs,
                                                    const float *__restrict__ blocks, const float *__restrict__ B,
                                                    uint32_t b_bytes, uint32_t N, uint32_t ldb, float *__restrict__ C,
                                                    uint32_t ldc, uint32_t xcd_chunk) {
    __shared__ f32x4_t partial[3][4][64];  // waves 1..3, 4 tiles, one vector per lane
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = xcd_block(blockIdx.x, xcd_chunk);
    if (item >= Mb * nST) return;  // workgroup-uniform, before any barrier
    const uint32_t R = item / nST, st = item - R * nST;
    const uint32_t c = lane & 15, g = lane >> 4;
    const uint32_t ncol = st * 64 + c * 4;  // first of this lane's 4 interleaved output columns
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = ncol < N ? ncol * 4u : kDropLoad;  // columns past N read as zero
    const uint32_t ldb4 = ldb * 4u;

    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t bs = blockRowPtrs[R], be = blockRowPtrs[R + 1];
    if (bs + wave < be) {
        const uint32_t last = be - 1;
        auto load_frag = [&](uint32_t b, uint32_t bcol, F32Frag &f) {  // b may run past `last`: clamped, unused
            const float *ablk = blocks + static_cast<size_t>(min(b, last)) * 256u + c * 16u + g;
            const uint32_t voff = b <= last ? lane_off : kDropLoad;  // the prefetch past the row fetches nothing
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f.a[s] = ablk[4 * s];
                // the lane-dependent row (4s + g) rides in voffset: a non-uniform soffset would make
                // hipcc wrap every load in a waterfall loop
                f.bv[s] = buffer_load_vec<4>(rsrc, voff + (4 * s + g) * ldb4, bcol * 16u * ldb4);
            }
        };
        uint32_t col_next = blockColIdxs[min(bs + wave + 4, last)];
        F32Frag cur;
        load_frag(bs + wave, blockColIdxs[bs + wave], cur);
        for (uint32_t b = bs + wave; b < be; b += 4) {
            const uint32_t col_next2 = blockColIdxs[min(b + 8, last)];
            F32Frag nxt;
            load_frag(b + 4, col_next, nxt);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[s], cur.bv[s][t], acc[t], 0, 0, 0);
            }
            cur = nxt;
            col_next = col_next2;
        }
    }
    // fixed-order reduction of the four partial tiles: wave 0 adds waves 1, 2, 3 in that order
    if (wave != 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) partial[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] += partial[w][t][lane];
    }
    if (ncol < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4_t out{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            *reinterpret_cast<f32x4_t *>(C + static_cast<size_t>(R * 16 + g * 4 + r) * ldc + ncol) = out;
        }
    }
}

// --------------------------------------------------------------------------------- bsr_mfma_bf16
// One WORKGROUP per (block row, super-tile of 16 * TPL output columns); its 4 waves split the block
// row's block pairs round-robin (pair p -> wave p % 4), each keeps a partial 16 x (16 * TPL) fp32
// tile, and the partials are summed through LDS in fixed wave order (deterministic) before the
// store.  MFMA 16x16x32 bf16: lane (c, g) holds A[i = c][k = 8g .. 8g+7] and B[k = 8g .. 8g+7][j = c].
// k slots 0..15 are the 16 columns of the pair's first block, slots 16..31 those of its second
// block (zero when an odd block is left over).  A lane reads, for each of its 8 k rows, the TPL
// consecutive bf16 of columns TPL*c .. TPL*c + TPL-1 (one per accumulator tile: 16 bytes at
// TPL = 8, so one wave-instruction covers 128 columns of 4 B rows) and regroups them per tile with
// v_perm_b32: the transpose B needs, done in registers.  Optional two-deep software pipeline (PIPE).  TPL = 8 (N >= 128) halves the B-read instruction count of TPL = 4: the kernel is bound by
// the texture-address path (33 100 blocks each pull a 16 x N panel), not by MFMA or HBM.
template <int TPL>
struct Bf16Frag {
    u32x4_t araw;
    uint32_t braw[8][TPL / 2];
};

template <int TPL, bool C_BF16, bool PIPE = true, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, PIPE ? 1 : (WAVES == 4 ? 5 : 4)) void bsr_mfma_bf16(uint32_t Mb, uint32_t nST, const uint32_t *__restrict__ blockRowPtrs,
                                                     const uint32_t *__restrict__ blockColIdxs,
                                                     const uint16_t *__restrict__ blocks, const uint16_t *__restrict__ B,
                                                     uint32_t b_bytes, uint32_t N, uint32_t ldb, void *__restrict__ Cv,
                                                     uint32_t ldc, uint32_t xcd_chunk) {
    __shared__ f32x4_t partial[WAVES - 1][TPL][64];  // waves 1..WAVES-1, TPL tiles, one vector per lane
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = xcd_block(blockIdx.x, xcd_chunk);
    if (item >= Mb * nST) return;  // workgroup-uniform, before any barrier
    const uint32_t R = item / nST, st = item - R * nST;
    const uint32_t c = lane & 15, g = lane >> 4;
    const uint32_t ncol = st * (16 * TPL) + c * TPL;  // first of this lane's TPL interleaved output columns
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = ncol < N ? ncol * 2u : kDropLoad;
    const uint32_t ldb2 = ldb * 2u;
    const uint32_t khalf = (g & 1) * 8u;  // first block column of this lane's 8 k slots
    const uint32_t second = g >> 1;       // lanes g = 2,3 take the second block of the pair

    f32x4_t acc[TPL];
#pragma unroll
    for (int t = 0; t < TPL; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t bs = blockRowPtrs[R], be = blockRowPtrs[R + 1];
    const uint32_t npairs = (be - bs + 1) / 2;
    if (wave < npairs) {
        const uint32_t last = be - 1;
        auto block_of = [&](uint32_t pair) { return bs + 2 * pair + second; };
        auto load_frag = [&](uint32_t pair, uint32_t bcol, Bf16Frag<TPL> &f) {
            const uint32_t b = block_of(pair);
            const bool have = b <= last;  // false for the missing half of an odd pair and past the row
            f.araw = *reinterpret_cast<const u32x4_t *>(blocks + static_cast<size_t>(min(b, last)) * 256u + c * 16u + khalf);
            if (!have) f.araw = u32x4_t{0u, 0u, 0u, 0u};
            // the block column differs between the two halves of the wave, so the whole row offset
            // is per-lane (voffset); soffset stays 0 (a non-uniform soffset costs a waterfall loop)
            const uint32_t voff = have ? lane_off + (bcol * 16u + khalf) * ldb2 : kDropLoad;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (TPL == 8) {
                    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + e * ldb2, 0, 0);
#pragma unroll
                    for (int w = 0; w < 4; ++w) f.braw[e][w] = r[w];
                } else {
                    const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + e * ldb2, 0, 0);
                    f.braw[e][0] = r[0];
                    f.braw[e][1] = r[1];
                }
            }
        };
        auto multiply = [&](const Bf16Frag<TPL> &f) {
            const bf16x8_t afrag = __builtin_bit_cast(bf16x8_t, f.araw);
#pragma unroll
            for (int w = 0; w < TPL / 2; ++w) {
                // tiles 2w and 2w+1 take the low and high bf16 of dword w of every k row; once both are packed the
                // eight raw dwords are dead, so the regrouped operands never coexist with more than one dword column
                u32x4_t even, odd;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t lo = f.braw[2 * p][w], hi = f.braw[2 * p + 1][w];
                    even[p] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);  // {hi.h0, lo.h0}
                    odd[p] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);   // {hi.h1, lo.h1}
                }
                acc[2 * w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8_t, even), acc[2 * w], 0, 0, 0);
                acc[2 * w + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8_t, odd), acc[2 * w + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (PIPE) {
            uint32_t col_next = blockColIdxs[min(block_of(wave + WAVES), last)];
            Bf16Frag<TPL> cur;
            load_frag(wave, blockColIdxs[min(block_of(wave), last)], cur);
            for (uint32_t pair = wave; pair < npairs; pair += WAVES) {
                const uint32_t col_next2 = blockColIdxs[min(block_of(pair + 2 * WAVES), last)];
                Bf16Frag<TPL> nxt;
                load_frag(pair + WAVES, col_next, nxt);
                multiply(cur);
                cur = nxt;
                col_next = col_next2;
            }
        } else {
            // no register double buffer: half the fragment registers, more resident waves hide the latency.
            // Tried on top of this and measured no better (config 4, 11.5 us): A blocks and block columns fetched two
            // iterations ahead (11.7 us, costs a resident wave), fewer resident workgroups so that the dispatcher
            // balances the uneven block rows dynamically (12.4-13.5 us), write-through C stores (no gain).
            uint32_t col_cur = blockColIdxs[min(block_of(wave), last)];
            for (uint32_t pair = wave; pair < npairs; pair += WAVES) {
                const uint32_t col_next = blockColIdxs[min(block_of(pair + WAVES), last)];
                Bf16Frag<TPL> cur;
                load_frag(pair, col_cur, cur);
                multiply(cur);
                col_cur = col_next;
            }
        }
    }
    // fixed-order reduction of the four partial tiles: wave 0 adds waves 1, 2, 3 in that order
    if (wave != 0) {
#pragma unroll
        for (int t = 0; t < TPL; ++t) partial[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < WAVES - 1; ++w) {
#pragma unroll
        for (int t = 0; t < TPL; ++t) acc[t] += partial[w][t][lane];
    }
    if (ncol < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t crow = static_cast<size_t>(R * 16 + g * 4 + r) * ldc + ncol;
            if constexpr (C_BF16) {
                using bf2 = __bf16 __attribute__((ext_vector_type(2)));
                uint32_t o[TPL / 2];
#pragma unroll
                for (int w = 0; w < TPL / 2; ++w)
                    o[w] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(acc[2 * w][r]), static_cast<__bf16>(acc[2 * w + 1][r])});
                uint16_t *dst = static_cast<uint16_t *>(Cv) + crow;
                if constexpr (TPL == 8) *reinterpret_cast<u32x4_t *>(dst) = u32x4_t{o[0], o[1], o[2], o[3]};
                else *reinterpret_cast<u32x2_t *>(dst) = u32x2_t{o[0], o[1]};
            } else {
                float *dst = static_cast<float *>(Cv) + crow;
#pragma unroll
                for (int w = 0; w < TPL / 4; ++w)
                    *reinterpret_cast<f32x4_t *>(dst + 4 * w) =
                        f32x4_t{acc[4 * w][r], acc[4 * w + 1][r], acc[4 * w + 2][r], acc[4 * w + 3][r]};
            }
        }
    }
}

// ------------------------------------------------------------------------------- bsrc_mfma_bf16
// Column-compacted block rows (mispmm_bsr_compact_bf16_host): at 16 x 16 the SuiteSparse matrices of data/ fill their
// blocks to ~2 % (ACTIVSg10K: 26.5 blocks = 424 block columns per block row, of which 71 hold a non-zero), so the
// dense-block kernel above gathers a 16-row B panel for every block: 135 MB through the vector L1s for 5 MB of B.
// Here a block row is its list of occupied columns, padded to a multiple of 32, and its values gathered to those
// columns as [16 rows][32 k] bf16 tiles: one v_mfma_f32_16x16x32_bf16 K step per 32 occupied columns (2.7 steps per
// block row instead of 13.2 block pairs), each B row fetched by its own index.  One WAVE owns a (block row, 128
// output columns) item: no cross-wave reduction, no LDS, no barrier; a two-deep register pipeline keeps the next
// step's 8 B-row reads (16 bytes per lane each) and its A tile in flight while this step's eight tiles are multiplied.
// Lane (c, g): A operand row c, k 8g .. 8g+7 of the step; B operand rows cols[8g .. 8g+7], its TPL = 8 interleaved
// output columns 8c .. 8c+7 (tile t <-> column 8c + t), regrouped per tile with v_perm_b32 as in bsr_mfma_bf16.
// Terms of an output element are summed in ascending k = storage order of the occupied columns; deterministic.
template <bool C_BF16>
__global__ __launch_bounds__(256) void bsrc_mfma_bf16(uint32_t Mb, uint32_t nST, const uint32_t *__restrict__ stepPtrs,
                                                      const uint32_t *__restrict__ cols, const uint16_t *__restrict__ tiles,
                                                      const uint16_t *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                                      void *__restrict__ Cv, uint32_t ldc, uint32_t xcd_chunk) {
    constexpr int TPL = 8;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = xcd_block(blockIdx.x, xcd_chunk) * 4 + wave;
    if (item >= Mb * nST) return;  // wave-uniform; no barrier in this kernel
    const uint32_t R = item / nST, st = item - R * nST;
    const uint32_t c = lane & 15, g = lane >> 4;
    const uint32_t ncol = st * (16 * TPL) + c * TPL;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = ncol < N ? ncol * 2u : kDropLoad;
    const uint32_t ldb2 = ldb * 2u;

    f32x4_t acc[TPL];
#pragma unroll
    for (int t = 0; t < TPL; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t s0 = stepPtrs[R], s1 = stepPtrs[R + 1];

    struct Step {
        u32x4_t araw;
        uint32_t braw[8][TPL / 2];
    };
    auto load_idx = [&](uint32_t s, u32x4_t &lo, u32x4_t &hi) {  // this lane's 8 B-row indices of step s
        const u32x4_t *p = reinterpret_cast<const u32x4_t *>(cols + static_cast<size_t>(s) * 32u + g * 8u);
        lo = p[0];
        hi = p[1];
    };
    auto load_step = [&](uint32_t s, const u32x4_t &lo, const u32x4_t &hi, Step &f) {
        f.araw = *reinterpret_cast<const u32x4_t *>(tiles + static_cast<size_t>(s) * 512u + c * 32u + g * 8u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t col = e < 4 ? lo[e] : hi[e - 4];
            // padding (0xFFFFFFFF) and lanes past N: a dropped read (zeros); the coefficients there are zero as well
            const uint32_t voff = col == 0xFFFFFFFFu ? kDropLoad : col * ldb2 + lane_off;
            const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
#pragma unroll
            for (int w = 0; w < 4; ++w) f.braw[e][w] = r[w];
        }
    };
    auto multiply = [&](const Step &f) {
        const bf16x8_t afrag = __builtin_bit_cast(bf16x8_t, f.araw);
#pragma unroll
        for (int w = 0; w < TPL / 2; ++w) {
            u32x4_t even, odd;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t lo = f.braw[2 * p][w], hi = f.braw[2 * p + 1][w];
                even[p] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);  // {hi.h0, lo.h0}
                odd[p] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);   // {hi.h1, lo.h1}
            }
            acc[2 * w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8_t, even), acc[2 * w], 0, 0, 0);
            acc[2 * w + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8_t, odd), acc[2 * w + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (s0 < s1) {
        const uint32_t last = s1 - 1;
        u32x4_t ilo, ihi, nlo, nhi;
        load_idx(s0, ilo, ihi);
        load_idx(min(s0 + 1, last), nlo, nhi);
        Step cur;
        load_step(s0, ilo, ihi, cur);
        for (uint32_t s = s0; s < s1; ++s) {
            Step nxt;
            const bool more = s + 1 < s1;  // wave-uniform
            if (more) load_step(s + 1, nlo, nhi, nxt);
            load_idx(min(s + 2, last), nlo, nhi);
            multiply(cur);
            if (more) cur = nxt;
        }
    }
    if (ncol < N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t crow = static_cast<size_t>(R * 16 + g * 4 + r) * ldc + ncol;
            if constexpr (C_BF16) {
                using bf2 = __bf16 __attribute__((ext_vector_type(2)));
                uint32_t o[TPL / 2];
#pragma unroll
                for (int w = 0; w < TPL / 2; ++w)
                    o[w] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(acc[2 * w][r]), static_cast<__bf16>(acc[2 * w + 1][r])});
                *reinterpret_cast<u32x4_t *>(static_cast<uint16_t *>(Cv) + crow) = u32x4_t{o[0], o[1], o[2], o[3]};
            } else {
                float *dst = static_cast<float *>(Cv) + crow;
#pragma unroll
                for (int w = 0; w < TPL / 4; ++w)
                    *reinterpret_cast<f32x4_t *>(dst + 4 * w) = f32x4_t{acc[4 * w][r], acc[4 * w + 1][r], acc[4 * w + 2][r], acc[4 * w + 3][r]};
            }
        }
    }
}

// ----------------------------------------------------------------------------- bsr_mfma_bf16_b32
// 32 x 32 blocks: a block's 32 columns are exactly the K = 32 of one v_mfma_f32_16x16x32_bf16, so no
// pairing; its 32 rows are two 16-row halves that share the B fragment (one B fetch, two MFMAs per
// tile).  Work split, column interleave, pipeline and LDS reduction as in the 16 x 16 kernel.
template <int TPL>
struct Bf16Frag32 {
    u32x4_t araw[2];
    uint32_t braw[8][TPL / 2];
};

template <int TPL, bool C_BF16>
__global__ __launch_bounds__(256) void bsr_mfma_bf16_b32(uint32_t Mb, uint32_t nST, const uint32_t *__restrict__ blockRowPtrs,
                                                         const uint32_t *__restrict__ blockColIdxs,
                                                         const uint16_t *__restrict__ blocks,
                                                         const uint16_t *__restrict__ B, uint32_t b_bytes, uint32_t N,
                                                         uint32_t ldb, void *__restrict__ Cv, uint32_t ldc,
                                                         uint32_t xcd_chunk) {
    __shared__ f32x4_t partial[3][2 * TPL][64];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = xcd_block(blockIdx.x, xcd_chunk);
    if (item >= Mb * nST) return;  // workgroup-uniform, before any barrier
    const uint32_t R = item / nST, st = item - R * nST;
    const uint32_t c = lane & 15, g = lane >> 4;
    const uint32_t ncol = st * (16 * TPL) + c * TPL;
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = ncol < N ? ncol * 2u : kDropLoad;
    const uint32_t ldb2 = ldb * 2u;

    f32x4_t acc[2][TPL];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < TPL; ++t) acc[h][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint32_t bs = blockRowPtrs[R], be = blockRowPtrs[R + 1];
    if (bs + wave < be) {
        const uint32_t last = be - 1;
        auto load_frag = [&](uint32_t b, uint32_t bcol, Bf16Frag32<TPL> &f) {
            const bool have = b <= last;
            const uint16_t *ablk = blocks + static_cast<size_t>(min(b, last)) * 1024u + c * 32u + g * 8u;
            f.araw[0] = *reinterpret_cast<const u32x4_t *>(ablk);
            f.araw[1] = *reinterpret_cast<const u32x4_t *>(ablk + 16 * 32);
            const uint32_t voff = have ? lane_off + (bcol * 32u + g * 8u) * ldb2 : kDropLoad;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (TPL == 8) {
                    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + e * ldb2, 0, 0);
#pragma unroll
                    for (int w = 0; w < 4; ++w) f.braw[e][w] = r[w];
                } else {
                    const auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff + e * ldb2, 0, 0);
                    f.braw[e][0] = r[0];
                    f.braw[e][1] = r[1];
                }
            }
        };
        uint32_t col_next = blockColIdxs[min(bs + wave + 4, last)];
        Bf16Frag32<TPL> cur;
        load_frag(bs + wave, blockColIdxs[bs + wave], cur);
        for (uint32_t b = bs + wave; b < be; b += 4) {
            const uint32_t col_next2 = blockColIdxs[min(b + 8, last)];
            Bf16Frag32<TPL> nxt;
            load_frag(b + 4, col_next, nxt);  // dropped B reads past the row; the A it re-reads is never used
            const bf16x8_t a0 = __builtin_bit_cast(bf16x8_t, cur.araw[0]);
            const bf16x8_t a1 = __builtin_bit_cast(bf16x8_t, cur.araw[1]);
#pragma unroll
            for (int t = 0; t < TPL; ++t) {
                u32x4_t packed;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const uint32_t lo = cur.braw[2 * p][t >> 1], hi = cur.braw[2 * p + 1][t >> 1];
                    packed[p] = (t & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                }
                const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, packed);
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bfrag, acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bfrag, acc[1][t], 0, 0, 0);
            }
            cur = nxt;
            col_next = col_next2;
        }
    }
    if (wave != 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < TPL; ++t) partial[wave - 1][h * TPL + t][lane] = acc[h][t];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < TPL; ++t) acc[h][t] += partial[w][h * TPL + t][lane];
    if (ncol < N) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t crow = static_cast<size_t>(R * 32 + h * 16 + g * 4 + r) * ldc + ncol;
                if constexpr (C_BF16) {
                    using bf2 = __bf16 __attribute__((ext_vector_type(2)));
                    uint32_t o[TPL / 2];
#pragma unroll
                    for (int w = 0; w < TPL / 2; ++w)
                        o[w] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(acc[h][2 * w][r]),
                                                                static_cast<__bf16>(acc[h][2 * w + 1][r])});
                    uint16_t *dst = static_cast<uint16_t *>(Cv) + crow;
                    if constexpr (TPL == 8) *reinterpret_cast<u32x4_t *>(dst) = u32x4_t{o[0], o[1], o[2], o[3]};
                    else *reinterpret_cast<u32x2_t *>(dst) = u32x2_t{o[0], o[1]};
                } else {
                    float *dst = static_cast<float *>(Cv) + crow;
#pragma unroll
                    for (int w = 0; w < TPL / 4; ++w)
                        *reinterpret_cast<f32x4_t *>(dst + 4 * w) =
                            f32x4_t{acc[h][4 * w][r], acc[h][4 * w + 1][r], acc[h][4 * w + 2][r], acc[h][4 * w + 3][r]};
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------- dispatch
struct BsrArgs {
    hipStream_t stream;
    uint32_t Mb, K, bR, bC;
    const uint32_t *ptrs, *idxs;
    const float *blocks, *B;
    uint32_t N, ldb;
    float *C;
    uint32_t ldc;
};

template <int G, int VEC, class Acc>
static void launch_valu(const BsrArgs &a) {
    const uint32_t M = a.Mb * a.bR;
    dim3 grid(ceil_div(M, 256 / G), ceil_div(a.N, G * VEC));
    const uint64_t bytes = static_cast<uint64_t>(a.K) * a.ldb * 4u;
    note_kernel("bsr_valu<G%d,V%d,%s>", G, VEC, acc_tag<Acc>());
    if (bytes > 0x7FFFFFFFull)
        hipLaunchKernelGGL((bsr_valu<G, VEC, Acc, true>), grid, dim3(256), 0, a.stream, M, a.bR, a.bC, a.ptrs, a.idxs,
                           a.blocks, a.B, 0u, a.N, a.ldb, a.C, a.ldc);
    else
        hipLaunchKernelGGL((bsr_valu<G, VEC, Acc, false>), grid, dim3(256), 0, a.stream, M, a.bR, a.bC, a.ptrs, a.idxs,
                           a.blocks, a.B, static_cast<uint32_t>(bytes), a.N, a.ldb, a.C, a.ldc);
}

template <int VEC, class Acc>
static void launch_valu_g(const BsrArgs &a, int g) {
    switch (g) {
        case 8: launch_valu<8, VEC, Acc>(a); break;
        case 16: launch_valu<16, VEC, Acc>(a); break;
        case 32: launch_valu<32, VEC, Acc>(a); break;
        default: launch_valu<64, VEC, Acc>(a); break;
    }
}

template <int BD, int VEC, class Acc>
static void launch_rowblock(const BsrArgs &a) {
    constexpr int RS = BD < 8 ? BD : 8;
    const uint32_t nCT = ceil_div(a.N, 64u * VEC);
    const XcdGrid xg = xcd_grid(ceil_div(a.Mb * nCT * (BD / RS), 4u));
    note_kernel("bsr_rowblock<BD%d,RS%d,V%d,%s>", BD, RS, VEC, acc_tag<Acc>());
    hipLaunchKernelGGL((bsr_rowblock<BD, RS, VEC, Acc>), dim3(xg.grid), dim3(256), 0, a.stream, a.Mb, nCT, a.ptrs, a.idxs,
                       a.blocks, a.B, static_cast<uint32_t>(static_cast<uint64_t>(a.K) * a.ldb * 4u), a.N, a.ldb, a.C, a.ldc,
                       xg.chunk);
}

// true when the block-row kernel took the launch
template <class Acc>
static bool try_rowblock(const BsrArgs &a, int vec) {
    if (a.bR != a.bC || static_cast<uint64_t>(a.K) * a.ldb * 4u > 0x7FFFFFFFull) return false;
    if (vec == 4 && a.N <= 128) vec = 2;  // keep all 64 lanes busy at N <= 128
#define MISPMM_RB_CASE(BD, VV)                  \
    if (a.bR == BD && vec == VV) {              \
        launch_rowblock<BD, VV, Acc>(a);        \
        return true;                            \
    }
    MISPMM_RB_CASE(4, 1) MISPMM_RB_CASE(4, 2) MISPMM_RB_CASE(4, 4) MISPMM_RB_CASE(8, 1) MISPMM_RB_CASE(8, 2) MISPMM_RB_CASE(8, 4)
    MISPMM_RB_CASE(16, 1) MISPMM_RB_CASE(16, 2) MISPMM_RB_CASE(16, 4) MISPMM_RB_CASE(32, 1) MISPMM_RB_CASE(32, 2) MISPMM_RB_CASE(32, 4)
#undef MISPMM_RB_CASE
    return false;
}

template <class Acc>
static void launch_valu_v(const BsrArgs &a, int vec) {
    if (try_rowblock<Acc>(a, vec)) return;
    const int g = pick_group(a.N, vec);
    if (vec == 4) launch_valu_g<4, Acc>(a, g);
    else if (vec == 2) launch_valu_g<2, Acc>(a, g);
    else launch_valu_g<1, Acc>(a, g);
}

static bool mfma_shape_ok(uint32_t K, uint32_t N, uint32_t ldb, uint32_t ldc, const void *B, const void *C,
                          const void *blocks, size_t elem) {
    // whole 16-byte (fp32) / 8-byte (bf16) vectors of 4 columns; B must fit a < 2 GiB buffer descriptor
    return N % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && aligned16(blocks) &&
           (reinterpret_cast<uintptr_t>(B) % (4 * elem) == 0) && (reinterpret_cast<uintptr_t>(C) % 8 == 0) &&
           static_cast<uint64_t>(K) * ldb * elem <= 0x7FFFFFFFull;
}

}  // namespace mispmm

using namespace mispmm;

extern "C" int mispmm_bsr_f32(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t bR, uint32_t bC,
                              uint32_t numBlocks, const uint32_t *blockRowPtrs, const uint32_t *blockColIdxs,
                              const float *blocks, const float *B, uint32_t N, uint32_t ldb, float *C, uint32_t ldc,
                              int kernel, int acc_mode) {
    if (kernel < 0 || kernel > MISPMM_BSR_NUM_KERNELS) return fail(MISPMM_ERR_INVALID_ARG, "bsr: unknown kernel id %d", kernel);
    if (acc_mode != MISPMM_ACC_REFERENCE && acc_mode != MISPMM_ACC_FAST)
        return fail(MISPMM_ERR_INVALID_ARG, "bsr: unknown accumulate mode %d", acc_mode);
    if (bR == 0 || bC == 0) return fail(MISPMM_ERR_INVALID_ARG, "bsr: zero block dimension");
    if (numBlockRows == 0 || N == 0) return MISPMM_OK;
    if (!blockRowPtrs) return fail(MISPMM_ERR_INVALID_ARG, "bsr: blockRowPtrs is null");
    if (numBlocks != 0 && (!blockColIdxs || !blocks)) return fail(MISPMM_ERR_INVALID_ARG, "bsr: null block arrays");
    if (int s = check_dense_args(B, N, ldb, C, ldc)) return s;
    hipStream_t st = as_stream(stream);
    const bool mfma_ok = bR == 16 && bC == 16 && aligned16(C) && mfma_shape_ok(K, N, ldb, ldc, B, C, blocks, 4);
    if (kernel == 2 && !mfma_ok)
        return fail(MISPMM_ERR_UNSUPPORTED, "bsr: MFMA kernel needs 16x16 blocks, N/ldb/ldc multiples of 4, 16-byte aligned operands");
    if (kernel == 2 && acc_mode == MISPMM_ACC_REFERENCE)
        return fail(MISPMM_ERR_UNSUPPORTED, "bsr: the MFMA kernel has FAST (fused) numerics only");
    if (kernel == MISPMM_KERNEL_AUTO) kernel = (mfma_ok && acc_mode == MISPMM_ACC_FAST) ? 2 : 1;
    if (kernel == 2) {
        const uint32_t nST = ceil_div(N, 64u);
        const XcdGrid xg = xcd_grid(numBlockRows * nST);  // one workgroup per (block row, super-tile)
        note_kernel("bsr_mfma_f32");
        hipLaunchKernelGGL(bsr_mfma_f32, dim3(xg.grid), dim3(256), 0, st, numBlockRows, nST, blockRowPtrs, blockColIdxs,
                           blocks, B, static_cast<uint32_t>(static_cast<uint64_t>(K) * ldb * 4u), N, ldb, C, ldc, xg.chunk);
    } else {
        const BsrArgs a{st, numBlockRows, K, bR, bC, blockRowPtrs, blockColIdxs, blocks, B, N, ldb, C, ldc};
        const int vec = pick_vec(B, ldb, C, ldc, N);
        if (acc_mode == MISPMM_ACC_REFERENCE) launch_valu_v<AccRefF32>(a, vec);
        else launch_valu_v<AccFast>(a, vec);
    }
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

extern "C" int mispmm_bsr_bf16(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t bR, uint32_t bC,
                               uint32_t numBlocks, const uint32_t *blockRowPtrs, const uint32_t *blockColIdxs,
                               const uint16_t *blocks, const uint16_t *B, uint32_t N, uint32_t ldb, void *C,
                               uint32_t ldc, int c_bf16) {
    if (!((bR == 16 && bC == 16) || (bR == 32 && bC == 32)))
        return fail(MISPMM_ERR_UNSUPPORTED, "bsr_bf16: only 16x16 or 32x32 blocks (got %ux%u)", bR, bC);
    if (numBlockRows == 0 || N == 0) return MISPMM_OK;
    if (!blockRowPtrs || !B || !C) return fail(MISPMM_ERR_INVALID_ARG, "bsr_bf16: null pointer");
    if (numBlocks != 0 && (!blockColIdxs || !blocks)) return fail(MISPMM_ERR_INVALID_ARG, "bsr_bf16: null block arrays");
    if (ldb < N || ldc < N) return fail(MISPMM_ERR_INVALID_ARG, "bsr_bf16: leading dimension smaller than N");
    if (!mfma_shape_ok(K, N, ldb, ldc, B, C, blocks, 2) || (!c_bf16 && !aligned16(C)))
        return fail(MISPMM_ERR_UNSUPPORTED, "bsr_bf16: N/ldb/ldc must be multiples of 4 and operands vector-aligned");
    // 8 tiles per lane (128-column super-tiles, 16-byte B reads) once N fills them and 16-byte vectors
    // line up; else 4 tiles (64 columns, 8-byte reads)
    const bool wide = N >= 128 && N % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && aligned16(B) && aligned16(C);
    const uint32_t nST = ceil_div(N, wide ? 128u : 64u);
    const XcdGrid xg = xcd_grid(numBlockRows * nST);  // one workgroup per (block row, super-tile)
    dim3 grid(xg.grid);
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(K) * ldb * 2u);
#define MISPMM_BF16_LAUNCH(KERNEL, TPL, CB)                                                                     \
    hipLaunchKernelGGL((KERNEL<TPL, CB>), grid, dim3(256), 0, as_stream(stream), numBlockRows, nST, blockRowPtrs, \
                       blockColIdxs, blocks, B, b_bytes, N, ldb, C, ldc, xg.chunk)
    // the 128-column kernel runs WITHOUT the register double buffer by default: 72 instead of 104 VGPRs, and the
    // extra resident waves hide the fetch latency better than the prefetch did (13.3 -> 12.2 us on config 4);
    // MISPMM_BSR_PIPE=1 restores the two-deep software pipeline (measurement aid)
    static const bool bsr_pipe = knob_int("MISPMM_BSR_PIPE", 0) == 1;
#define MISPMM_BF16_PICK(KERNEL)                                                                 \
    do {                                                                                         \
        if (wide) {                                                                              \
            if (c_bf16) MISPMM_BF16_LAUNCH(KERNEL, 8, true); else MISPMM_BF16_LAUNCH(KERNEL, 8, false); \
        } else {                                                                                 \
            if (c_bf16) MISPMM_BF16_LAUNCH(KERNEL, 4, true); else MISPMM_BF16_LAUNCH(KERNEL, 4, false); \
        }                                                                                        \
    } while (0)
    // MISPMM_BSR_LDS=1: the LDS-staged kernel (bsr_bf16_lds.hpp: LDS-DMA ring + ds_read_b64_tr_b16, the plan of round 1).
    // Passes the same parity tests as the register-staged kernel below but is SLOWER on config 4 (14.9 us at ring depth 4,
    // 15.9 at 6, 18.1 at 8, against 11.4 us; profiles/r2/bsr_bf16_variants.log), so it stays opt-in.
    static const bool bsr_lds = knob_int("MISPMM_BSR_LDS", 0) == 1;
    const uint64_t blocks_bytes = static_cast<uint64_t>(numBlocks) * 512u;
    if (bR == 16 && bsr_lds && N % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && aligned16(B) && aligned16(C) && aligned16(blocks) &&
        blocks_bytes <= 0x7FFFFFFFull) {
        const uint32_t nst = ceil_div(N, 64u);
        const XcdGrid g = xcd_grid(numBlockRows * nst);
        // ring depth 4 (3 block pairs = 15 KiB in flight per workgroup, 8 workgroups per CU); MISPMM_BSR_DEPTH=6|8 deepens it
        static const int depth = knob_int("MISPMM_BSR_DEPTH", 4);
        note_kernel("bsr_bf16_lds<%s,D%d>", c_bf16 ? "c16" : "c32", depth == 4 ? 4 : depth == 6 ? 6 : 8);
#define MISPMM_LDS_LAUNCH(CB, DD)                                                                                              \
    hipLaunchKernelGGL((bsr_bf16_lds<CB, DD>), dim3(g.grid), dim3(128), 0, as_stream(stream), numBlockRows, nst, blockRowPtrs, \
                       blockColIdxs, blocks, static_cast<uint32_t>(blocks_bytes), B, b_bytes, N, ldb, C, ldc, g.chunk)
        if (depth == 4) {
            if (c_bf16) MISPMM_LDS_LAUNCH(true, 4); else MISPMM_LDS_LAUNCH(false, 4);
        } else if (depth == 6) {
            if (c_bf16) MISPMM_LDS_LAUNCH(true, 6); else MISPMM_LDS_LAUNCH(false, 6);
        } else {
            if (c_bf16) MISPMM_LDS_LAUNCH(true, 8); else MISPMM_LDS_LAUNCH(false, 8);
        }
#undef MISPMM_LDS_LAUNCH
        MISPMM_LAUNCH_CHECK();
        return MISPMM_OK;
    }
    // (8 waves per workgroup -- WAVES = 8, half the iterations per wave -- was measured slower: 13.2 vs 11.6 us)
    note_kernel("bsr_mfma_bf16%s<T%d,%s>", bR == 32 ? "_b32" : "", wide ? 8 : 4, c_bf16 ? "c16" : "c32");
    if (bR == 16 && !bsr_pipe && wide) {
        if (c_bf16) hipLaunchKernelGGL((bsr_mfma_bf16<8, true, false>), grid, dim3(256), 0, as_stream(stream), numBlockRows, nST, blockRowPtrs, blockColIdxs, blocks, B, b_bytes, N, ldb, C, ldc, xg.chunk);
        else hipLaunchKernelGGL((bsr_mfma_bf16<8, false, false>), grid, dim3(256), 0, as_stream(stream), numBlockRows, nST, blockRowPtrs, blockColIdxs, blocks, B, b_bytes, N, ldb, C, ldc, xg.chunk);
    } else if (bR == 16) MISPMM_BF16_PICK(bsr_mfma_bf16);
    else MISPMM_BF16_PICK(bsr_mfma_bf16_b32);
#undef MISPMM_BF16_PICK
#undef MISPMM_BF16_LAUNCH
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

extern "C" int mispmm_bsrc_bf16(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t nSteps, const uint32_t *stepPtrs,
                                const uint32_t *cols, const uint16_t *tiles, const uint16_t *B, uint32_t N, uint32_t ldb, void *C,
                                uint32_t ldc, int c_bf16) {
    if (numBlockRows == 0 || N == 0) return MISPMM_OK;
    if (!stepPtrs || !B || !C) return fail(MISPMM_ERR_INVALID_ARG, "bsrc_bf16: null pointer");
    if (nSteps != 0 && (!cols || !tiles)) return fail(MISPMM_ERR_INVALID_ARG, "bsrc_bf16: null step arrays");
    if (ldb < N || ldc < N) return fail(MISPMM_ERR_INVALID_ARG, "bsrc_bf16: leading dimension smaller than N");
    if (N % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || !aligned16(B) || !aligned16(C) || !aligned16(tiles) || !aligned16(cols) ||
        static_cast<uint64_t>(K) * ldb * 2u > 0x7FFFFFFFull)
        return fail(MISPMM_ERR_UNSUPPORTED, "bsrc_bf16: N / ldb / ldc must be multiples of 8, operands 16-byte aligned, B below 2 GiB");
    const uint32_t nST = ceil_div(N, 128u);
    const XcdGrid xg = xcd_grid(ceil_div(numBlockRows * nST, 4u));
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(K) * ldb * 2u);
    note_kernel("bsrc_mfma_bf16<%s>", c_bf16 ? "c16" : "c32");
    if (c_bf16)
        hipLaunchKernelGGL((bsrc_mfma_bf16<true>), dim3(xg.grid), dim3(256), 0, as_stream(stream), numBlockRows, nST, stepPtrs, cols, tiles,
                           B, b_bytes, N, ldb, C, ldc, xg.chunk);
    else
        hipLaunchKernelGGL((bsrc_mfma_bf16<false>), dim3(xg.grid), dim3(256), 0, as_stream(stream), numBlockRows, nST, stepPtrs, cols, tiles,
                           B, b_bytes, N, ldb, C, ldc, xg.chunk);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

extern "C" int mispmm_bsrc_slots_bf16(mispmm_stream_t stream, uint32_t numBlockRows, uint32_t K, uint32_t nSteps, const uint32_t *extraPtrs,
                                      const uint32_t *cols, const uint16_t *tiles, const uint16_t *B, uint32_t N, uint32_t ldb, void *C,
                                      uint32_t ldc, int c_bf16) {
    if (numBlockRows == 0 || N == 0) return MISPMM_OK;
    if (!extraPtrs || !cols || !tiles || !B || !C) return fail(MISPMM_ERR_INVALID_ARG, "bsrc_slots_bf16: null pointer");
    // every product of sizes in 64 bits: the slot count, the grid and the byte extents must fit what the kernel indexes with
    const uint64_t slot_steps = static_cast<uint64_t>(numBlockRows) * kBsrSlots;
    if (nSteps < slot_steps)
        return fail(MISPMM_ERR_INVALID_ARG, "bsrc_slots_bf16: nSteps %u is less than %u slots per block row", nSteps, kBsrSlots);
    if (static_cast<uint64_t>(numBlockRows) * ceil_div(N, 128u) > 0x7FFFFFFFull)   // one workgroup per (block row, 128 columns)
        return fail(MISPMM_ERR_UNSUPPORTED, "bsrc_slots_bf16: %u block rows x %u columns exceed the grid of this kernel", numBlockRows, N);
    if (ldb < N || ldc < N) return fail(MISPMM_ERR_INVALID_ARG, "bsrc_slots_bf16: leading dimension smaller than N");
    if (N % 8 != 0 || ldb % 8 != 0 || ldc % 8 != 0 || !aligned16(B) || !aligned16(C) || !aligned16(tiles) || !aligned16(cols) ||
        static_cast<uint64_t>(K) * ldb * 2u > 0x7FFFFFFFull)
        return fail(MISPMM_ERR_UNSUPPORTED, "bsrc_slots_bf16: N / ldb / ldc must be multiples of 8, operands 16-byte aligned, B below 2 GiB");
    const uint32_t nST = ceil_div(N, 128u);
    const XcdGrid xg = xcd_grid(numBlockRows * nST);  // one workgroup per (block row, 128 columns)
    const uint32_t b_bytes = static_cast<uint32_t>(static_cast<uint64_t>(K) * ldb * 2u);
    const uint64_t c_bytes = static_cast<uint64_t>(numBlockRows) * 16u * ldc * (c_bf16 ? 2u : 4u);
    // C store policy: MISPMM_BSR_STORE = -1 plain global stores, 2 non-temporal, 16 write-through (sc1), 18 both
    // (measurement aid; buffer stores need C below 2 GiB, else plain)
    static const int store_knob = knob_int("MISPMM_BSR_STORE", 2);
    const int st = c_bytes <= 0x7FFFFFFFull ? store_knob : -1;
    // no extra steps (every block row has at most 4) and an fp32 C: the kernel that deals the reduce and the store over the
    // four waves.  With a bf16 C -- half the store volume, 4 instead of 8 store instructions for wave 0 -- dealing them out
    // measured 6 % SLOWER (4.12 -> 4.37 us, profiles/r3/bsr_share_ab.log), with an fp32 C 5 % faster (4.73 -> 4.49).
    // MISPMM_BSR_SHARE=0 / 2: never / whenever there are no extra steps (measurement aid).
    static const int share_knob = knob_int("MISPMM_BSR_SHARE", 1);
    // nSteps == 4 * numBlockRows leaves no room for extra steps: by the layout's contract (mispmm.h) extraPtrs then describes
    // none, and the SHARE kernel does not read it
    const bool share = nSteps == slot_steps && (share_knob == 2 || (share_knob == 1 && !c_bf16));
    note_kernel("bsrc_slots_mfma_bf16<%s,%s%s>", c_bf16 ? "c16" : "c32", st == 2 ? "nt" : st == 16 ? "sc1" : st == 18 ? "sc1nt" : "plain",
                share ? ",share" : "");
#define MISPMM_SLOTS_LAUNCH(CB, ST, SH)                                                                                            \
    hipLaunchKernelGGL((bsrc_slots_mfma_bf16<CB, ST, SH>), dim3(xg.grid), dim3(256), 0, as_stream(stream), numBlockRows, nST, extraPtrs, \
                       cols, tiles, B, b_bytes, N, ldb, C, static_cast<uint32_t>(st >= 0 ? c_bytes : 0), ldc, xg.chunk)
#define MISPMM_SLOTS_PICK(CB, SH)                        \
    do {                                                 \
        if (st == 2) MISPMM_SLOTS_LAUNCH(CB, 2, SH);     \
        else if (st == 16) MISPMM_SLOTS_LAUNCH(CB, 16, SH); \
        else if (st == 18) MISPMM_SLOTS_LAUNCH(CB, 18, SH); \
        else MISPMM_SLOTS_LAUNCH(CB, -1, SH);            \
    } while (0)
    if (share) {
        if (c_bf16) MISPMM_SLOTS_PICK(true, true); else MISPMM_SLOTS_PICK(false, true);
    } else {
        if (c_bf16) MISPMM_SLOTS_PICK(true, false); else MISPMM_SLOTS_PICK(false, false);
    }
#undef MISPMM_SLOTS_PICK
#undef MISPMM_SLOTS_LAUNCH
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

#ifdef MISPMM_STAMPS
// diagnostic build only: where the waves of bsrc_slots_mfma_bf16 leave their stamps (8 x uint64 per wave, 4 waves per workgroup)
extern "C" int mispmm_debug_set_stamps_bsr(void *device_buffer) {
    MISPMM_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(mispmm_bsr_stamp_buf), &device_buffer, sizeof(device_buffer)));
    return MISPMM_OK;
}
#endif
