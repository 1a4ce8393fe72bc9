// bsr_bf16_lds -- 16 x 16 bf16 BSR x dense on v_mfma_f32_16x16x32_bf16 with every operand staged through LDS.
//
// What the register-staged kernel (bsr_mfma_bf16) could not fix by tuning (profiles/r1/bsr_bf16_pmc_summary.txt:
// 42 % of wave time idle at the workgroup barrier, 4 % MFMA busy, ~6x L2 request amplification): its four waves
// split a block row's block pairs unevenly and meet in an LDS reduction, and every wave chases its own
// pointer -> column -> panel chain, so a block row of 54 blocks is a chain of 7 dependent round trips.  Here a
// workgroup of TWO waves owns one (block row, 64 output columns):
//   * the waves split the COLUMNS (32 each = two accumulator tiles): no reduction, no partial tiles, both walk
//     the same pairs, so the one barrier per pair costs next to nothing;
//   * the B panels (16 rows x 128 bytes per block) and the A block pair (1 KiB) arrive by LDS-DMA
//     (buffer_load_dwordx4 ... lds: no VGPRs, no staging instructions) into a ring of DEPTH slots, DEPTH - 1 pairs
//     ahead of the MFMAs: what bounds a launch is the LONGEST block row (a sequential chain by construction), and
//     its time is pairs / (pairs in flight) x latency -- a first version with 128-column workgroups and 2 pairs in
//     flight ran 14.2 us on ACTIVSg10K (27 pairs x 0.5 us) against 11.4 us for the register-staged kernel;
//   * the B operand's transpose is the LDS read itself (ds_read_b64_tr_b16), the A operand one ds_read_b128:
//     no v_perm, no fragment registers -- 30 VGPRs instead of 72-104;
//   * the LDS image of a panel is XOR-swizzled through the SOURCE addresses of the DMA (its destination is
//     lane-linear by construction) so that every transposed read is bank-conflict free:
//       16-byte chunk `ch` (8 columns) of panel row `row` lives at chunk  row * 8 + (ch ^ s(row)),
//       s(row) = 2 * (((row >> 1) & 1) | ((row >> 3) << 1));
//     the 32 lanes of a transposed read touch rows {4h .. 4h+3} and {8+4h .. 8+4h+3} of one panel, chunks 2t and
//     2t+1: 16 chunks whose positions differ mod 16, i.e. all 64 banks once.
// k slots 0..15 of an MFMA are the columns of the pair's first block, 16..31 those of its second; a missing
// second block (odd count) is a dropped DMA: LDS receives zeros for both its A half and its panel.
// The output tile goes back through LDS once so that C leaves as 16-byte row segments.
// Numerics: fp32 accumulate in ascending block order per output element, fixed by construction (deterministic).
// Roofline: HBM; algorithmic bytes nb*512 + nb*4 + (Mb+1)*4 + K*N*2 + M*N*{2,4}.
// MEASURED (MI355X, ACTIVSg10K BSR-16 x K=128): 14.9 us at DEPTH 4, 15.9 at 6, 18.1 at 8 -- slower than the
// register-staged kernel's 11.4 us, and slower the DEEPER the ring: with two MFMAs per wave and pair, the ~70
// instructions of an iteration (three DMA issues, five LDS reads and their wait, a barrier) are what a SIMD spends
// its time on, and a deeper ring only removes resident workgroups.  Kept opt-in (MISPMM_BSR_LDS=1) and under test.
#pragma once
#include "spmm_common.hpp"

namespace mispmm {

namespace bsr_lds {
using f32x4_t = float __attribute__((ext_vector_type(4)));
using bf16x8_t = short __attribute__((ext_vector_type(8)));
using s16x4_t = short __attribute__((ext_vector_type(4)));
using u32x2_t = uint32_t __attribute__((ext_vector_type(2)));
using u32x4_t = uint32_t __attribute__((ext_vector_type(4)));
using lds_ptr_t = __attribute__((address_space(3))) void *;

constexpr uint32_t kSlotA = 0, kSlotP0 = 1024, kSlotP1 = 1024 + 2048, kSlotBytes = 1024 + 2 * 2048;

// one LDS-DMA: 16 bytes per active lane from rsrc[voffset] to lds_base + lane * 16 (out of range = zeros)
__device__ __forceinline__ void dma16(rsrc_t rsrc, uint32_t lds_base, uint32_t voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, reinterpret_cast<lds_ptr_t>(static_cast<uintptr_t>(lds_base)), 16, voffset, 0, 0, 0);
}
// LDS reads the compiler must not count: an LDS-DMA in flight makes hipcc wait vmcnt(0) before any LDS read it
// knows about, which would serialise the ring; these are waited for by the explicit lgkmcnt below
__device__ __forceinline__ u32x4_t lds_read_b128(uint32_t addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int OFFSET>
__device__ __forceinline__ u32x2_t lds_read_tr16_b64(uint32_t addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFFSET) : "memory");
    return v;
}
}  // namespace bsr_lds

template <bool C_BF16, int DEPTH>
__global__ __launch_bounds__(128) void bsr_bf16_lds(uint32_t Mb, uint32_t nST, const uint32_t *__restrict__ blockRowPtrs,
                                                    const uint32_t *__restrict__ blockColIdxs,
                                                    const uint16_t *__restrict__ blocks, uint32_t blocks_bytes,
                                                    const uint16_t *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                                    void *__restrict__ Cv, uint32_t ldc, uint32_t xcd_chunk) {
    using namespace bsr_lds;
    static_assert(DEPTH >= 3 && (DEPTH - 2) * 3 <= 63, "the in-flight DMAs must fit the vmcnt field");
    __shared__ __attribute__((aligned(1024))) unsigned char ring[DEPTH * kSlotBytes];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0 or 1
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t item = xcd_block(blockIdx.x, xcd_chunk);
    if (item >= Mb * nST) return;  // workgroup-uniform, before any barrier
    const uint32_t R = item / nST, st = item - R * nST;
    const uint32_t c = lane & 15, g = lane >> 4;
    const uint32_t ring0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(ring));  // LDS byte address of the ring

    const uint32_t bs = __builtin_amdgcn_readfirstlane(blockRowPtrs[R]), be = __builtin_amdgcn_readfirstlane(blockRowPtrs[R + 1]);
    const uint32_t npairs = (be - bs + 1) / 2;
    const rsrc_t rsrcA = make_rsrc(blocks, blocks_bytes), rsrcB = make_rsrc(B, b_bytes);
    const uint32_t ldb2 = ldb * 2u;

    // ---- DMA sources of this lane -----------------------------------------------------------------------------
    // panel half of this wave: rows 8 * wave .. + 7, LDS chunk `lane` of the half = (row, pos) with the swizzle
    // undone on the source side: it fetches 16-byte chunk ch = pos ^ s(row) of that B row
    const uint32_t prow = 8u * wave + (lane >> 3), ppos = lane & 7u;
    const uint32_t pch = ppos ^ (2u * (((prow >> 1) & 1u) | ((prow >> 3) << 1)));
    const uint32_t pcol = st * 64u + pch * 8u;  // first of the chunk's 8 columns
    const uint32_t panel_voff = pcol < N ? prow * ldb2 + pcol * 2u : kDropLoad;  // + block column * 16 rows
    // A pair half: lanes 32 * wave .. + 31 move chunk `lane` of the pair's 1 KiB
    const bool a_mine = (lane >> 5) == wave;
    const uint32_t a_voff = lane * 16u;

    // pair p -> its DMAs into ring slot `slot`; a pair past the row's end, or the missing half of an odd pair, is a
    // dropped load (zeros): every wave issues exactly three LDS-DMAs per pair, which keeps vmcnt countable
    auto issue_pair = [&](uint32_t p, uint32_t slot, uint32_t col0, uint32_t col1) {
        const uint32_t b0 = bs + 2u * p, base = ring0 + slot * kSlotBytes;
        const bool have0 = b0 < be, have1 = b0 + 1u < be;
        const uint32_t a_off = have0 ? ((have1 || lane < 32u) ? b0 * 512u + a_voff : kDropLoad) : kDropLoad;
        if (a_mine) dma16(rsrcA, base + kSlotA, a_off);
        dma16(rsrcB, base + kSlotP0 + wave * 1024u, have0 ? panel_voff + col0 * 16u * ldb2 : kDropLoad);
        dma16(rsrcB, base + kSlotP1 + wave * 1024u, have1 ? panel_voff + col1 * 16u * ldb2 : kDropLoad);
    };
    auto col_of = [&](uint32_t b) -> uint32_t { return b < be ? blockColIdxs[b] : 0u; };  // wave-uniform: a scalar load

    // ---- LDS read addresses of this lane ----------------------------------------------------------------------
    // A operand: row c, k chunk (g & 1) of block (g >> 1) of the pair
    const uint32_t a_read = kSlotA + (g >> 1) * 512u + c * 32u + (g & 1u) * 16u;
    // B operand, tile tt of this wave (columns 32 * wave + 16 * tt + c), half h: rows 8 * (g & 1) + 4 * h + q of the
    // panel of block (g >> 1); lane 4q + pp of a 16-lane group addresses row q, columns 4pp .. 4pp + 3 of the tile
    const uint32_t q = (lane & 15u) >> 2, pp = lane & 3u;
    const uint32_t row_b = 8u * (g & 1u) + q, swz = 2u * (((q >> 1) & 1u) | ((g & 1u) << 1));
    const uint32_t panel_base = ((g >> 1) ? kSlotP1 : kSlotP0) + row_b * 128u + (pp & 1u) * 8u;
    const uint32_t b_read0 = panel_base + (((4u * wave + 0u + (pp >> 1)) ^ swz) * 16u);
    const uint32_t b_read1 = panel_base + (((4u * wave + 2u + (pp >> 1)) ^ swz) * 16u);

    f32x4_t acc0{0.f, 0.f, 0.f, 0.f}, acc1{0.f, 0.f, 0.f, 0.f};
    if (npairs != 0) {
        // prologue: pairs 0 .. DEPTH-2 in flight, block columns of pair DEPTH-1 fetched
        static_for<0, DEPTH - 1>([&](auto d) {
            constexpr uint32_t P = decltype(d)::value;
            issue_pair(P, P, col_of(bs + 2u * P), col_of(bs + 2u * P + 1u));
        });
        uint32_t nc0 = col_of(bs + 2u * (DEPTH - 1)), nc1 = col_of(bs + 2u * (DEPTH - 1) + 1u);
        uint32_t slot = 0u;
        for (uint32_t p = 0; p < npairs; ++p) {
            // this wave's share of pair p has landed once at most the DEPTH - 2 pairs after it are outstanding ...
            asm volatile("s_waitcnt vmcnt(%0)" : : "i"((DEPTH - 2) * 3) : "memory");
            // ... and after the barrier so has the other wave's; both are also done reading pair p - 1's slot
            __builtin_amdgcn_s_barrier();
            const uint32_t refill = slot == 0u ? DEPTH - 1u : slot - 1u;  // (p + DEPTH - 1) % DEPTH == (p - 1) % DEPTH
            const uint32_t c0 = nc0, c1 = nc1;
            nc0 = col_of(bs + 2u * (p + DEPTH));
            nc1 = col_of(bs + 2u * (p + DEPTH) + 1u);
            issue_pair(p + DEPTH - 1u, refill, c0, c1);
            const uint32_t base = ring0 + slot * kSlotBytes;
            u32x4_t a = lds_read_b128(base + a_read);
            u32x2_t b00 = lds_read_tr16_b64<0>(base + b_read0), b01 = lds_read_tr16_b64<512>(base + b_read0);
            u32x2_t b10 = lds_read_tr16_b64<0>(base + b_read1), b11 = lds_read_tr16_b64<512>(base + b_read1);
            // the wait names the registers it guards, so no use of them can be scheduled above it
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b00), "+v"(b01), "+v"(b10), "+v"(b11) : : "memory");
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, a);
            const bf16x8_t bf0 = __builtin_bit_cast(bf16x8_t, u32x4_t{b00[0], b00[1], b01[0], b01[1]});
            const bf16x8_t bf1 = __builtin_bit_cast(bf16x8_t, u32x4_t{b10[0], b10[1], b11[0], b11[1]});
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf1, acc1, 0, 0, 0);
            slot = slot == DEPTH - 1u ? 0u : slot + 1u;
        }
    }
    // ---- epilogue: the wave's 16 x 32 tile through LDS, out as 16-byte row segments ----------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dropped DMAs past the row's end still write (zeros) into the ring
    __builtin_amdgcn_s_barrier();                     // nobody reads the ring any more
    constexpr uint32_t kStride = 36;                  // floats per staged row: 16-byte aligned, off the bank period
    float *stage = reinterpret_cast<float *>(ring) + wave * (16u * kStride);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        stage[(4u * g + r) * kStride + c] = acc0[r];
        stage[(4u * g + r) * kStride + 16u + c] = acc1[r];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t col_w = st * 64u + wave * 32u;
    if constexpr (C_BF16) {
        // 16 rows x 4 chunks of 8 columns: one 16-byte store per lane
        const uint32_t row = lane >> 2, ch = lane & 3u, col = col_w + ch * 8u;
        if (col < N) {
            const float *src = stage + row * kStride + ch * 8u;
            using bf2 = __bf16 __attribute__((ext_vector_type(2)));
            u32x4_t o;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                o[i] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(src[2 * i]), static_cast<__bf16>(src[2 * i + 1])});
            *reinterpret_cast<u32x4_t *>(static_cast<uint16_t *>(Cv) + static_cast<size_t>(R * 16u + row) * ldc + col) = o;
        }
    } else {
        // 16 rows x 8 chunks of 4 columns: two 16-byte stores per lane
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t id = lane + 64u * i, row = id >> 3, ch = id & 7u, col = col_w + ch * 4u;
            if (col < N) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t *>(stage + row * kStride + ch * 4u);
                *reinterpret_cast<f32x4_t *>(static_cast<float *>(Cv) + static_cast<size_t>(R * 16u + row) * ldc + col) = v;
            }
        }
    }
}

}  // namespace mispmm
