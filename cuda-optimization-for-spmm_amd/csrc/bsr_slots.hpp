// bsrc_slots_mfma_bf16: column-compacted 16-row block rows, one WORKGROUP per (block row, 128 output columns), the
// block row's MFMA K steps dealt over its 4 waves.
//
// bsrc_mfma_bf16 (spmm_bsr.hip) gives a block row to ONE wave, which walks a chain of three dependent memory hops
// (stepPtrs -> column list -> B rows) once per step: ACTIVSg10K has 1250 block rows of 2-4 steps, so 1250 waves on
// 1024 SIMDs each wait out ~3 + 2.7 round trips -- latency-bound at 6.2 us for 19 MB.  Here
//   * a block row's first kSlots = 4 steps sit in FIXED slots (step R * 4 + w belongs to wave w of block row R): no
//     pointer hop in front of the column list, which -- the step being wave-uniform -- is read through the scalar
//     cache; slots past the row's step count hold a padding column list (128 bytes read, nothing else);  steps past the fourth are "extra" steps found through extraPtrs (scalar load, issued with
//     the first hop);
//   * every wave therefore runs hop 1 (its slot's 32 column indices) -> hop 2 (its 32 B rows x 256 bytes and its A
//     tile) -> 8 MFMAs, all 4 waves of a block row at the same time;
//   * the 4 partial 16 x 128 tiles are added in wave order (= ascending step order for rows of <= 4 steps): waves 1..3
//     park theirs in LDS, wave 0 adds them to its own and stores the 16 rows.
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
//
// Resources decide whether this kernel works at all: config 4 has 1250 block rows for 256 CUs (4.9 workgroups per CU
// on average, but the dispatcher hands some shader engines 6 per CU), and a workgroup that has to wait for a slot
// starts ~3 us late (stamps: profiles/r3/stamps_bsrc_slots.log).  So the footprint is kept to 6+ workgroups per CU:
//   * every step is multiplied into FRESH accumulators (the first MFMA of a tile takes the constant 0 as its addend), so
//     no accumulator is live while the 8 B-row reads (32 VGPRs) are in flight;
//   * waves 1..3 park their tile in LDS (24 KiB per workgroup), wave 0 keeps its own in registers and adds the three
//     parked tiles in wave order; extra steps (block rows of more than 4 steps) go to waves 1..3 only, which add them
//     into their parked tile -- a lane re-reads exactly the words it wrote, so that needs no barrier.
//
// SHARE (every block row has <= 4 steps, i.e. no extra steps -- config 4): the reduce and the store are dealt over the
// four waves as well.  Wave q owns rows 4q .. 4q+3 of the block row: every wave parks the three row quads it does not
// own (the lanes of the other three lane groups; 6 KiB per wave, the same 24 KiB per workgroup), and after the barrier
// the 16 lanes of lane group q of wave q add the quads the other USED slots parked to their own, in wave order, and
// store 4 rows.  With wave 0 doing all of it the block row's result waited for 24 LDS reads + 8 stores of one wave
// behind the barrier (stamps: 0.84 us p90 from barrier to the last store issued); dealt out it is a quarter of that
// per wave: 4.73 -> 4.50 us on config 4 in a same-process A/B (profiles/r3/bsr_share_ab.log).  Tried on top of it and not
// kept: the A tile read issued before the column list returns (+0.2 %), the LDS reads of two rows in flight at once
// (more registers, a resident workgroup less per CU: +73 %).  The stamps of this version show the kernel in two phases,
// reads until ~2.0 us (the barrier), then 10 MB of C stores from every workgroup at once.
template <bool C_BF16, int ST, bool SHARE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void bsrc_slots_mfma_bf16(uint32_t Mb, uint32_t nST, const uint32_t *__restrict__ extraPtrs,
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
    // [wave - 1][row of the block row][128 columns]: 24 KiB.  No padding needed: a ds_write_b128 is served in groups of
    // 8 consecutive lanes = 256 contiguous bytes here, and so is the read back.
    __shared__ float parked[kBsrSlots - 1][16][128];
    static_assert(sizeof(parked) == kBsrSlots * (kBsrSlots - 1) * 4 * 128 * sizeof(float), "SHARE views the same 24 KiB as [wave][quad slot][4][128]");

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

    // this lane's 8 B-row indices of step s (lane group g multiplies k = 8g .. 8g+7).  The step index is wave-uniform, so
    // the 32 indices come through the SCALAR cache (two s_load_dwordx16) and are dealt to the lane groups by selects:
    // a vector load here queues behind the B reads of the waves that started earlier (stamps: 0.6 us median, 1.4 us p90
    // for 128 bytes) and every B read of this wave waits for it
    struct Idx {
        uint32_t v[8];
    };
    auto load_idx = [&](uint32_t s, Idx &ix) {
        using u32x16_t = uint32_t __attribute__((ext_vector_type(16)));
        const u32x16_t *p = reinterpret_cast<const u32x16_t *>(cols + static_cast<size_t>(__builtin_amdgcn_readfirstlane(s)) * 32u);
        const u32x16_t lo = p[0], hi = p[1];  // whole-vector loads: hipcc otherwise turns the selects into branches around single loads
        // branch-free deal (v_and_or_b32 with the index in an SGPR): ternaries here become 32 exec-masked branches
        const uint32_t m0 = g == 0 ? ~0u : 0u, m1 = g == 1 ? ~0u : 0u, m2 = g == 2 ? ~0u : 0u, m3 = g == 3 ? ~0u : 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) ix.v[e] = (lo[e] & m0) | (lo[8 + e] & m1) | (hi[e] & m2) | (hi[8 + e] & m3);
        return lo[0];
    };
    // one MFMA K step (32 occupied columns) into fresh accumulators: tile t <-> output column 8c + t
    auto step_tile = [&](uint32_t s, const Idx &ix, f32x4_t (&t)[TPL], bool stamp) {
        uint32_t braw[8][TPL / 2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t col = ix.v[e];
            // padding (0xFFFFFFFF) and lanes past N: a dropped read (zeros); the coefficients there are zero as well
            const uint32_t voff = col == 0xFFFFFFFFu ? kDropLoad : col * ldb2 + lane_off;
            const auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
#pragma unroll
            for (int w = 0; w < 4; ++w) braw[e][w] = r[w];
        }
        const u32x4_t araw = *reinterpret_cast<const u32x4_t *>(tiles + static_cast<size_t>(s) * 512u + c * 32u + g * 8u);
        // all nine reads are on their way before the first v_perm: without this fence hipcc interleaves the operand
        // regrouping with the loads and waits for them two at a time (four round trips instead of one)
        __builtin_amdgcn_sched_barrier(0);
#ifdef MISPMM_STAMPS
        if (stamp) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MISPMM_BSR_STAMP(2);
        }
#endif
        const bf16x8_t afrag = __builtin_bit_cast(bf16x8_t, araw);
        const f32x4_t zero{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < TPL / 2; ++w) {
            u32x4_t even, odd;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t l2 = braw[2 * p][w], h2 = braw[2 * p + 1][w];
                even[p] = __builtin_amdgcn_perm(h2, l2, 0x05040100u);  // {hi.h0, lo.h0}
                odd[p] = __builtin_amdgcn_perm(h2, l2, 0x07060302u);   // {hi.h1, lo.h1}
            }
            t[2 * w] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8_t, even), zero, 0, 0, 0);
            t[2 * w + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8_t, odd), zero, 0, 0, 0);
        }
    };

    // hop 1: this wave's slot (column list) and the extent of the block row's extra steps
    Idx ix;
    // a slot past the row's step count is all padding; a used slot starts with a real column (padding only ever trails)
    const uint32_t first = load_idx(R * kBsrSlots + wave, ix);
    const uint32_t ex0 = extraPtrs[R], ex1 = extraPtrs[R + 1];
    MISPMM_BSR_STAMP(1);
    // The two roles are separate branches (wave-uniform; each executes exactly one barrier) so that the register
    // allocator sees their lifetimes apart: wave 0 holds its tile across the barrier, waves 1..3 hold nothing across it.
    auto slot_tile = [&](f32x4_t (&t)[TPL]) {
        if (first != 0xFFFFFFFFu) {
            step_tile(R * kBsrSlots + wave, ix, t, true);  // hop 2
        } else {
#pragma unroll
            for (int i = 0; i < TPL; ++i) t[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#ifdef MISPMM_STAMPS
            bstamp[2] = bstamp[1];
#endif
        }
    };
    // one row of the result: this lane's 8 columns
    auto store_row = [&](uint32_t row, const f32x4_t &s0, const f32x4_t &s1) {
        const size_t crow = static_cast<size_t>(row) * ldc + ncol;
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
    };
    if constexpr (SHARE) {
        f32x4_t t[TPL];
        slot_tile(t);  // zeros for an unused slot: parked like any other, so the adds below need no case distinction
        float *quads = &parked[0][0][0];  // [wave][quad slot 0..2][row of the quad][128 columns]
        if (g != wave) {
            // lane (c, g) holds rows 4g .. 4g+3 = quad g, which wave g owns; quad slot = g with this wave's own quad left out
            float *dst = quads + ((wave * (kBsrSlots - 1) + (g - (g > wave ? 1u : 0u))) * 4u) * 128u + c * 8u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                *reinterpret_cast<f32x4_t *>(dst + r * 128) = f32x4_t{t[0][r], t[1][r], t[2][r], t[3][r]};
                *reinterpret_cast<f32x4_t *>(dst + r * 128 + 4) = f32x4_t{t[4][r], t[5][r], t[6][r], t[7][r]};
            }
        }
        MISPMM_BSR_STAMP(3);
        __syncthreads();
        MISPMM_BSR_STAMP(4);
        // one straight-line body per owner wave (the position of the wave's own tile in the sum is then a constant: every
        // LDS read of a row is issued before the first add; with the wave index as data hipcc serialised them behind branches)
        auto reduce_quad = [&](auto owner) {
            constexpr uint32_t Q = decltype(owner)::value;
            if (g != Q || ncol >= N) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4_t p0[kBsrSlots], p1[kBsrSlots];
#pragma unroll
                for (uint32_t w = 0; w < kBsrSlots; ++w) {
                    if (w == Q) {
                        p0[w] = f32x4_t{t[0][r], t[1][r], t[2][r], t[3][r]};
                        p1[w] = f32x4_t{t[4][r], t[5][r], t[6][r], t[7][r]};
                    } else {
                        const float *src = quads + ((w * (kBsrSlots - 1) + (Q - (Q > w ? 1u : 0u))) * 4u + r) * 128u + c * 8u;
                        p0[w] = *reinterpret_cast<const f32x4_t *>(src);
                        p1[w] = *reinterpret_cast<const f32x4_t *>(src + 4);
                    }
                }
                // wave order = ascending step order, as the kernel without SHARE adds them
                store_row(R * 16 + Q * 4 + r, ((p0[0] + p0[1]) + p0[2]) + p0[3], ((p1[0] + p1[1]) + p1[2]) + p1[3]);
            }
        };
        if (wave == 0) reduce_quad(std::integral_constant<uint32_t, 0>{});
        else if (wave == 1) reduce_quad(std::integral_constant<uint32_t, 1>{});
        else if (wave == 2) reduce_quad(std::integral_constant<uint32_t, 2>{});
        else reduce_quad(std::integral_constant<uint32_t, 3>{});
    } else if (wave != 0) {
        {
            f32x4_t t[TPL];
            slot_tile(t);
            // park the tile: lane (c, g) holds rows 4g .. 4g+3, columns 8c .. 8c+7
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *dst = &parked[wave - 1][g * 4 + r][c * 8];
                *reinterpret_cast<f32x4_t *>(dst) = f32x4_t{t[0][r], t[1][r], t[2][r], t[3][r]};
                *reinterpret_cast<f32x4_t *>(dst + 4) = f32x4_t{t[4][r], t[5][r], t[6][r], t[7][r]};
            }
        }
        // block rows of more than 4 steps (more than 128 occupied columns): extra step i goes to wave 1 + i % 3
        for (uint32_t e = ex0 + (wave - 1); e < ex1; e += kBsrSlots - 1) {
            const uint32_t s = Mb * kBsrSlots + e;
            load_idx(s, ix);
            f32x4_t t[TPL];
            step_tile(s, ix, t, false);
            // the parked tile really lives in LDS between two extra steps (the clobber keeps hipcc from carrying it in
            // 32 registers across the B reads, which would cost every launch of this kernel a resident workgroup per CU)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float *dst = &parked[wave - 1][g * 4 + r][c * 8];
                f32x4_t a = *reinterpret_cast<f32x4_t *>(dst), b = *reinterpret_cast<f32x4_t *>(dst + 4);
                a += f32x4_t{t[0][r], t[1][r], t[2][r], t[3][r]};
                b += f32x4_t{t[4][r], t[5][r], t[6][r], t[7][r]};
                *reinterpret_cast<f32x4_t *>(dst) = a;
                *reinterpret_cast<f32x4_t *>(dst + 4) = b;
            }
            asm volatile("" ::: "memory");
        }
        MISPMM_BSR_STAMP(3);
        __syncthreads();
        MISPMM_BSR_STAMP(4);
    } else {
        f32x4_t acc[TPL];
        slot_tile(acc);
        MISPMM_BSR_STAMP(3);
        __syncthreads();
        MISPMM_BSR_STAMP(4);
        if (ncol < N) {
            // wave 0 adds the parked tiles in wave order and stores the block row's 16 x 128 result
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4_t s0{acc[0][r], acc[1][r], acc[2][r], acc[3][r]}, s1{acc[4][r], acc[5][r], acc[6][r], acc[7][r]};
#pragma unroll
                for (uint32_t p = 0; p < kBsrSlots - 1; ++p) {
                    s0 += *reinterpret_cast<const f32x4_t *>(&parked[p][g * 4 + r][c * 8]);
                    s1 += *reinterpret_cast<const f32x4_t *>(&parked[p][g * 4 + r][c * 8 + 4]);
                }
                store_row(R * 16 + g * 4 + r, s0, s1);
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

// (An experiment kernel with TWO waves per block row -- each taking two of the four step slots with all 16 B-row reads in
// flight: half the waves to start, one parked tile instead of three, 104 VGPRs -- measured 2.5 % slower, 4.86 against
// 4.75 us in a same-process A/B, profiles/r3/bsr_waves_ab.log, and was removed again.)

}  // namespace mispmm
