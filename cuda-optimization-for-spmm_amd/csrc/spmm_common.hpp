// Device-side building blocks shared by the CSR / ELL / BSR / COO kernels.
// gfx950 only: wave = 64 lanes, 16-byte global accesses, no CUDA-isms.
#pragma once
#include <type_traits>
#include <cstdlib>

#include "mispmm_internal.hpp"

namespace mispmm {

using f32x2 = float __attribute__((ext_vector_type(2)));
using f32x4 = float __attribute__((ext_vector_type(4)));

template <int VEC> struct VecOf;
template <> struct VecOf<1> { using type = float; };
template <> struct VecOf<2> { using type = f32x2; };
template <> struct VecOf<4> { using type = f32x4; };

template <int VEC>
__device__ __forceinline__ typename VecOf<VEC>::type load_vec(const float *p) {
    return *reinterpret_cast<const typename VecOf<VEC>::type *>(p);
}
template <int VEC>
__device__ __forceinline__ void store_vec(float *p, typename VecOf<VEC>::type v) {
    *reinterpret_cast<typename VecOf<VEC>::type *>(p) = v;
}
template <int VEC>
__device__ __forceinline__ float vec_get(const typename VecOf<VEC>::type &v, int i) {
    if constexpr (VEC == 1) return v; else return v[i];
}
template <int VEC>
__device__ __forceinline__ void vec_set(typename VecOf<VEC>::type &v, int i, float x) {
    if constexpr (VEC == 1) v = x; else v[i] = x;
}

// ---- XCD-aware workgroup order ---------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an XCD and its 4 MiB
// L2; MI355X_MICROARCH.md, workgroup dispatch).  Walking the rows in dispatch order would spread
// every XCD over the whole matrix, so each L2 would see all of B.  xcd_block() renumbers block b to
// (b % 8) * chunk + b / 8: each XCD then owns one contiguous eighth of the row blocks and only the
// B rows that eighth touches compete for its L2.  Placement is a speed matter only: any dispatch
// order gives the same result.  chunk == 0 keeps dispatch order.
__device__ __forceinline__ uint32_t xcd_block(uint32_t b, uint32_t chunk) {
    return chunk ? (b & 7u) * chunk + (b >> 3) : b;
}

// ---- bounds-checked buffer loads --------------------------------------------------------------
// B is read through a raw buffer descriptor spanning all of B (< 2 GiB): the hardware range-checks
// voffset (per lane) + soffset (wave-uniform SGPR) against the descriptor's byte count (measured on
// gfx950: soffset IS part of the check).  A load whose voffset has bit 31 set is therefore out of
// range whatever the soffset: it returns zeros and fetches nothing, which is how the kernels
// drop the unused slots of a fixed-size load batch without branching around a load (a branch
// there makes hipcc drain vmcnt before the next load and serialises the gather).
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr uint32_t kDropLoad = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), /*stride*/ 0, static_cast<int>(bytes),
                                             0x00020000);
}

// AUX = cache-policy bits of the load (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
template <int VEC, int AUX = 0>
__device__ __forceinline__ typename VecOf<VEC>::type buffer_load_vec(rsrc_t rsrc, uint32_t voffset, uint32_t soffset) {
    if constexpr (VEC == 1) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset, soffset, AUX));
    } else if constexpr (VEC == 2) {
        auto r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voffset, soffset, AUX);
        return __builtin_bit_cast(f32x2, r);
    } else {
        auto r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, AUX);
        return __builtin_bit_cast(f32x4, r);
    }
}
#ifndef MISPMM_B_LOAD_AUX
#define MISPMM_B_LOAD_AUX 0  // cache policy of the B-row reads of the row-gather kernel (measurement builds set 2 = nt)
#endif

// C leaves through buffer stores with a cache policy (aux bits: 1 = sc0, 2 = nt, 16 = sc1).  Measured on the headline
// (3.2 MB of C per launch, back-to-back launches; profiles/r3/store_policy.log): plain stores 3.66 us per launch (the
// dirty lines are written back at the kernel boundary), sc1 write-through 3.60-3.66 (round 2's choice: every store waits
// for the fabric), NON-TEMPORAL 3.39-3.41: nt lines are the L2's first candidates for eviction, so they stream out while
// the kernel still runs and nobody waits for them.  Same ordering of the gains on ELL K=256 (6.18 -> 5.62 us) and on the
// bf16 BSR kernel (sc1 6.9, plain 6.0, nt 5.1 us); sc1|nt together is the slowest form (11.2 us there).
// Visibility is that of a plain store: the end-of-kernel release writes back whatever is still dirty.
#ifdef MISPMM_X_STORE_AUX
constexpr int kCStoreAux = MISPMM_X_STORE_AUX;  // experiment builds (make variant ... DEFS=-DMISPMM_X_STORE_AUX=16)
#else
constexpr int kCStoreAux = 2;  // nt
#endif
template <int VEC>
__device__ __forceinline__ void buffer_store_vec_c(rsrc_t rsrc, uint32_t voffset, typename VecOf<VEC>::type v) {
    if constexpr (VEC == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, voffset, 0, kCStoreAux);
    } else if constexpr (VEC == 2) {
        using u2 = uint32_t __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rsrc, voffset, 0, kCStoreAux);
    } else {
        using u4 = uint32_t __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), rsrc, voffset, 0, kCStoreAux);
    }
}

// ---- accumulation policies (mispmm.h: enum mispmm_acc_mode) -----------------------------------
// This library is compiled with -ffp-contract=off, so `a * b` followed by `+` stays two
// roundings exactly as written; the fused form is spelled __builtin_fmaf.

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [BEGIN, END)
template <int BEGIN, int END, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

// Lane SRC of every G-lane group to the whole group, SRC known at compile time: one DPP move
// (row_newbcast, ctrl 0x150 + lane of the 16-lane row) instead of an LDS-crossbar permute, so the value
// needs no register until the instruction that uses it.  G = 8: two groups share a DPP row.
template <int G, int SRC>
__device__ __forceinline__ uint32_t group_bcast(uint32_t x) {
    static_assert(G == 8 || G == 16, "row_newbcast works inside one 16-lane row");
    if constexpr (G == 16) {
        return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x150 + SRC, 0xf, 0xf, true));
    } else {
        const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x150 + SRC, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x150 + 8 + SRC, 0xf, 0xf, true);
        return static_cast<uint32_t>((threadIdx.x & 8u) ? hi : lo);
    }
}

// Reference CSR: fp32 product, double running sum, one final rounding
// (/root/reference/src/spmm/csr/spmm_csr.cpp:20-25 with AccT = double).
struct AccRefWide {
    using T = double;
    static __device__ __forceinline__ void mac(T &acc, float a, float b) {
        float p = a * b;
        acc += static_cast<double>(p);
    }
    static __device__ __forceinline__ float finish(T acc) { return static_cast<float>(acc); }
    // four terms at once: the fp32 products as two packed multiplies (same IEEE rounding per element), then the
    // widening adds
    static __device__ __forceinline__ void mac4(T *acc, float a, float b0, float b1, float b2, float b3) {
        using f2 = float __attribute__((ext_vector_type(2)));
        const f2 av{a, a};
        const f2 p01 = av * f2{b0, b1}, p23 = av * f2{b2, b3};
        acc[0] += static_cast<double>(p01[0]);
        acc[1] += static_cast<double>(p01[1]);
        acc[2] += static_cast<double>(p23[0]);
        acc[3] += static_cast<double>(p23[1]);
    }
    // two terms: one packed multiply (same IEEE rounding per element), two widening adds
    static __device__ __forceinline__ void mac2(T *acc, float a, float b0, float b1) {
        using f2 = float __attribute__((ext_vector_type(2)));
        const f2 p = f2{a, a} * f2{b0, b1};
        acc[0] += static_cast<double>(p[0]);
        acc[1] += static_cast<double>(p[1]);
    }
};

// Reference COO / ELL / BSR: fp32 product then fp32 add (e.g. spmm_ell.cpp:25).
struct AccRefF32 {
    using T = float;
    static __device__ __forceinline__ void mac(T &acc, float a, float b) {
        float p = a * b;
        acc = acc + p;
    }
    static __device__ __forceinline__ float finish(T acc) { return acc; }
};

// Fast: one fused multiply-add per term.
struct AccFast {
    using T = float;
    static __device__ __forceinline__ void mac(T &acc, float a, float b) { acc = __builtin_fmaf(a, b, acc); }
    static __device__ __forceinline__ float finish(T acc) { return acc; }
};

template <class Acc> constexpr const char *acc_tag() {
    if constexpr (std::is_same_v<Acc, AccRefWide>) return "ref64";
    else if constexpr (std::is_same_v<Acc, AccRefF32>) return "ref32";
    else return "fast";
}

// Widest vector width usable for row-major dense operands B (ldb) and C (ldc) with N columns.
inline int pick_vec(const float *B, uint32_t ldb, const float *C, uint32_t ldc, uint32_t N) {
    static const int cap = knob_int("MISPMM_VEC", 4);
    if (cap >= 4 && N % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && aligned16(B) && aligned16(C)) return 4;
    if (cap >= 2 && N % 2 == 0 && ldb % 2 == 0 && ldc % 2 == 0 && aligned8(B) && aligned8(C)) return 2;
    return 1;
}

// Lanes per row group for the grouped kernels: the smallest power of two in [8, 64] whose
// G * VEC covers N (64 when even a full wave does not).
inline int pick_group(uint32_t N, int vec) {
    uint32_t need = ceil_div(N, static_cast<uint32_t>(vec));
    int g = 8;
    while (g < 64 && static_cast<uint32_t>(g) < need) g <<= 1;
    return g;
}

inline int check_dense_args(const float *B, uint32_t N, uint32_t ldb, const float *C, uint32_t ldc) {
    if (!B || !C) return fail(MISPMM_ERR_INVALID_ARG, "B or C is null");
    if (ldb < N || ldc < N) return fail(MISPMM_ERR_INVALID_ARG, "leading dimension smaller than N (N=%u ldb=%u ldc=%u)", N, ldb, ldc);
    return MISPMM_OK;
}

}  // namespace mispmm
