// bsrc_slots_mfma_bf16: column-compacted 16-row block rows, one WORKGROUP per (block row, 128 output columns), the
// block row's MFMA K steps dealt over its 4 waves.
//
// bsrc_mfma_bf16 (spmm_bsr.hip) gives a block row to ONE wave, which walks a chain of three dependent memory hops
// (stepPtrs -> column list -> B rows) once per step: ACTIVSg10K has 1250 block rows of 2-4 steps, so 1250 waves on
// 1024 SIMDs each wait out ~3 + 2.7 round trips -- latency-bound at 6.2 us for 19 MB.  Here
//   * a block row's first kSlots = 4 steps sit in FIXED slots (step R * 4 + w belongs to wave w of block row R): no
//     pointer hop in front of the column list; slots past the row's step count hold a padding column list (128 bytes
//     read, nothing else);  steps past the fourth are "extra" steps found through extraPtrs (scalar load, issued with
//     the first hop);
//   * every wave therefore runs hop 1 (its slot's 32 column indices) -> hop 2 (its 32 B rows x 256 bytes and its A
//     tile) -> 8 MFMAs, all 4 waves of a block row at the same time;
//   * the 4 partial 16 x 128 tiles are added in wave order (= ascending step order for rows of <= 4 steps) through
//     LDS, each wave finishing and storing 4 of the 16 rows as whole 512-byte (fp32 C) / 256-byte (bf16 C) segments.
// Deterministic; same operand rounding and the same bound against the oracle as bsrc_mfma_bf16, sums in a different
// (fixed) order.  Replaces the thread-per-block-element atomicAdd kernel of /root/reference/src/spmm/bsr/spmm_bsr_k1.cu:9-41
// for BASELINE config 4 (bf16 is a new capability: the reference has none).
#pragma once
#include "spmm_common.hpp"

namespace mispmm {

constexpr uint32_t kBsrSlots = 4;  // fixed step slots per block row = waves per workgroup

#ifdef MISPMM_STAMPS
// Diagnostic build only (tools/stamp_bsr.py): every wave of the LAST launch leaves s_memrealtime stamps (100 MHz) here.
static __device__ unsigned long long *mispmm_bsr_stamp_buf = nullptr;
#define MISPMM_BSR_STAMP(i) bstamp[i] = wall_clock64()
#else
#define MISPMM_BSR_STAMP(i)
#endif

// ST: cache policy of the C stores -- -1 = plain global stores, else buffer stores with that aux value (2 = nt, 16 = sc1)
template <bool C_BF16, int ST>
__global__ __launch_bounds__(256) void bsrc_slots_mfma_bf16(uint32_t Mb, uint32_t nST, const uint32_t *__restrict__ extraPtrs,
                                                            const uint32_t *__restrict__ cols, const uint16_t *__restrict__ tiles,
                                                            const uint16_t *__restrict__ B, uint32_t b_bytes, uint32_t N, uint32_t ldb,
                                                            void *__restrict__ Cv, uint32_t c_bytes, uint32_t ldc, uint32_t xcd_chunk) {
#ifdef MISPMM_STAMPS
    unsigned long long bstamp[7];
#endif
    MISPMM_BSR_STAMP(0);
    using f32x4_t = float __attribute__((ext_vector_type(4)));
    using bf16x8_t = short __attribute__((ext_vector_type(8)));
    using u32x4_t = uint32_t __attribute__((ext_vector_type(4)));
    constexpr int TPL = 8;
    // [wave][row of the block row][128 columns]: 32 KiB, so that 5 workgroups fit a CU's 160 KiB and all 1250 block rows of
    // config 4 are resident at once (4.9 per CU).  No padding needed: a ds_write_b128 is served in groups of 8 consecutive
    // lanes = 256 contiguous bytes, a ds_read_b128 row segment is 512 contiguous bytes.
    constexpr int LDP = 128;
    __shared__ float partial[kBsrSlots][16][LDP];

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

    f32x4_t acc[TPL];
#pragma unroll
    for (int t = 0; t < TPL; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

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
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t col = e < 4 ? lo[e] : hi[e - 4];
            // padding (0xFFFFFFFF) and lanes past N: a dropped read (zeros); the coefficients there are zero as well
            const uint32_t voff = col == 0xFFFFFFFFu ? kDropLoad : col * ldb2 + lane_off;
            const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
#pragma unroll
            for (int w = 0; w < 4; ++w) f.braw[e][w] = r[w];
        }
        f.araw = *reinterpret_cast<const u32x4_t *>(tiles + static_cast<size_t>(s) * 512u + c * 32u + g * 8u);
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
        }
    };

    // hop 1: this wave's slot (column list) and the extent of the block row's extra steps
    u32x4_t ilo, ihi;
    load_idx(R * kBsrSlots + wave, ilo, ihi);
    const uint32_t ex0 = extraPtrs[R], ex1 = extraPtrs[R + 1];
    // a slot past the row's step count is all padding; a used slot starts with a real column (padding only ever trails)
    const uint32_t first = __builtin_amdgcn_readfirstlane(ilo[0]);
    MISPMM_BSR_STAMP(1);
    if (first != 0xFFFFFFFFu) {
        Step cur;
        load_step(R * kBsrSlots + wave, ilo, ihi, cur);  // hop 2
#ifdef MISPMM_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MISPMM_BSR_STAMP(2);
        multiply(cur);
    }
#ifdef MISPMM_STAMPS
    else bstamp[2] = bstamp[1];
#endif
    // block rows of more than 4 steps (more than 128 occupied columns): wave w also takes extra steps w, w + 4, ...
    for (uint32_t e = ex0 + wave; e < ex1; e += kBsrSlots) {
        const uint32_t s = Mb * kBsrSlots + e;
        load_idx(s, ilo, ihi);
        Step cur;
        load_step(s, ilo, ihi, cur);
        multiply(cur);
    }

    // partial tiles to LDS: lane (c, g) holds rows 4g .. 4g+3, columns 8c .. 8c+7 (tile t <-> column 8c + t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float *dst = &partial[wave][g * 4 + r][c * 8];
        *reinterpret_cast<f32x4_t *>(dst) = f32x4_t{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        *reinterpret_cast<f32x4_t *>(dst + 4) = f32x4_t{acc[4][r], acc[5][r], acc[6][r], acc[7][r]};
    }
    MISPMM_BSR_STAMP(3);
    __syncthreads();
    MISPMM_BSR_STAMP(4);
    // wave w finishes rows 4w .. 4w+3: lane -> (row 4w + lane / 16, columns 8 (lane % 16) .. +7), partials added in wave order
    const uint32_t orow = wave * 4 + (lane >> 4), ocol = (lane & 15) * 8;
    f32x4_t s0 = *reinterpret_cast<const f32x4_t *>(&partial[0][orow][ocol]);
    f32x4_t s1 = *reinterpret_cast<const f32x4_t *>(&partial[0][orow][ocol + 4]);
#pragma unroll
    for (uint32_t p = 1; p < kBsrSlots; ++p) {
        s0 += *reinterpret_cast<const f32x4_t *>(&partial[p][orow][ocol]);
        s1 += *reinterpret_cast<const f32x4_t *>(&partial[p][orow][ocol + 4]);
    }
    const uint32_t gcol = st * (16 * TPL) + ocol;
    if (gcol < N) {
        const size_t crow = static_cast<size_t>(R * 16 + orow) * ldc + gcol;
        if constexpr (C_BF16) {
            using bf2 = __bf16 __attribute__((ext_vector_type(2)));
            u32x4_t o;
            o[0] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(s0[0]), static_cast<__bf16>(s0[1])});
            o[1] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(s0[2]), static_cast<__bf16>(s0[3])});
            o[2] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(s1[0]), static_cast<__bf16>(s1[1])});
            o[3] = __builtin_bit_cast(uint32_t, bf2{static_cast<__bf16>(s1[2]), static_cast<__bf16>(s1[3])});
            if constexpr (ST >= 0) {
                __builtin_amdgcn_raw_buffer_store_b128(o, make_rsrc(Cv, c_bytes), static_cast<uint32_t>(crow * 2u), 0, ST);
            } else {
                *reinterpret_cast<u32x4_t *>(static_cast<uint16_t *>(Cv) + crow) = o;
            }
        } else {
            if constexpr (ST >= 0) {
                const rsrc_t crs = make_rsrc(Cv, c_bytes);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, s0), crs, static_cast<uint32_t>(crow * 4u), 0, ST);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, s1), crs, static_cast<uint32_t>(crow * 4u + 16u), 0, ST);
            } else {
                float *dst = static_cast<float *>(Cv) + crow;
                *reinterpret_cast<f32x4_t *>(dst) = s0;
                *reinterpret_cast<f32x4_t *>(dst + 4) = s1;
            }
        }
    }
#ifdef MISPMM_STAMPS
    MISPMM_BSR_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MISPMM_BSR_STAMP(6);
    if (mispmm_bsr_stamp_buf && lane == 0) {
        unsigned long long *o = mispmm_bsr_stamp_buf + (static_cast<size_t>(blockIdx.x) * kBsrSlots + wave) * 8;
#pragma unroll
        for (int i = 0; i < 7; ++i) o[i] = bstamp[i];
        o[7] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) |
               static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 4));
    }
#endif
}

}  // namespace mispmm
