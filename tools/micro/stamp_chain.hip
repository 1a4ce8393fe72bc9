// Diagnostic build (never shipped): where does a wave of the headline CSR launch spend its time?
// Re-states the row-gather wave for G = 32, VEC = 4 with s_memrealtime stamps (100 MHz) after each
// dependent hop, written to a side buffer nothing else reads.
//   python tools/micro/dump_matrix.py /tmp/n4c6.bin ; hipcc --offload-arch=gfx950 -O3 -Icuda-optimization-for-spmm_amd/csrc -Iinclude tools/micro/stamp_chain.hip -o /tmp/stamp_chain ; /tmp/stamp_chain /tmp/n4c6.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "spmm_common.hpp"
using namespace mispmm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void stamped(uint32_t M, uint32_t rb_chunk, uint32_t log2p, uint32_t cols_per_part, uint32_t N,
                                               uint32_t ldb, const uint32_t *__restrict__ rowPtrs, const uint32_t *__restrict__ colIdxs,
                                               const float *__restrict__ vals, uint32_t b_bytes, const float *__restrict__ B,
                                               float *__restrict__ C, uint32_t c_bytes, uint32_t ldc, unsigned long long *stamps) {
    constexpr int G = 32, VEC = 4, GROUPS = 8, U = 16;
    const unsigned long long t0 = wall_clock64();
    const uint32_t lane = threadIdx.x % G;
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t p = xcd & ((1u << log2p) - 1u), q = xcd >> log2p;
    const uint32_t row = (p * rb_chunk + slot) * GROUPS + threadIdx.x / G;
    const uint32_t col0 = q * cols_per_part + blockIdx.y * (G * VEC) + lane * VEC;
    const bool row_ok = row < M, col_ok = col0 < min(N, (q + 1) * cols_per_part);
    uint32_t start = 0, len = 0;
    if (row_ok) { start = rowPtrs[row]; len = rowPtrs[row + 1] - start; }
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(start), "v"(len) : "memory");
    const unsigned long long t1 = wall_clock64();
    double acc[VEC] = {0, 0, 0, 0};
    const rsrc_t rsrc = make_rsrc(B, b_bytes);
    const uint32_t lane_off = col_ok ? col0 * 4u : kDropLoad;
    unsigned long long t2 = t1, t3 = t1;
    for (uint32_t base = 0; base < len; base += G) {
        const uint32_t cnt = min((uint32_t)G, len - base);
        const size_t mine = (size_t)start + base + min(lane, cnt - 1);
        uint32_t my_off = colIdxs[mine] * (ldb * 4u);
        float my_val = vals[mine];
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(my_off), "v"(my_val) : "memory");
        t2 = wall_clock64();
        for (uint32_t j = 0; j < cnt; j += U) {
            f32x4 bv[U]; float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t src = (j + u) & (G - 1);
                const uint32_t off = __shfl(my_off, src, G);
                const float a = __shfl(my_val, src, G);
                const bool live = j + u < cnt;
                av[u] = live ? a : 0.f;
                bv[u] = buffer_load_vec<4>(rsrc, live ? off + lane_off : kDropLoad, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t3 = wall_clock64();
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) AccRefWide::mac(acc[v], av[u], bv[u][v]);
        }
    }
    if (row_ok && col_ok) {
        f32x4 out{(float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]};
        buffer_store_vec_sc1<4>(make_rsrc(C, c_bytes), (row * ldc + col0) * 4u, out);
    }
    const unsigned long long t4 = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t5 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *s = stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4; s[5] = t5; s[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
    }
}

int main(int argc, char **argv) {
    FILE *f = fopen(argc > 1 ? argv[1] : "/tmp/n4c6.bin", "rb");
    if (!f) { printf("no matrix file\n"); return 1; }
    uint32_t hdr[3]; if (fread(hdr, 4, 3, f) != 3) return 1;
    const uint32_t M = hdr[0], K = hdr[1], nnz = hdr[2], N = 128;
    std::vector<uint32_t> rp(M + 1), ci(nnz); std::vector<float> va(nnz);
    if (fread(rp.data(), 4, M + 1, f) != M + 1 || fread(ci.data(), 4, nnz, f) != nnz || fread(va.data(), 4, nnz, f) != nnz) return 1;
    fclose(f);
    uint32_t *drp, *dci; float *dva, *dB, *dC; unsigned long long *dst;
    CK(hipMalloc(&drp, (M + 1) * 4)); CK(hipMalloc(&dci, nnz * 4)); CK(hipMalloc(&dva, nnz * 4));
    CK(hipMalloc(&dB, (size_t)K * N * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(drp, rp.data(), (M + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dci, ci.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dva, va.data(), nnz * 4, hipMemcpyHostToDevice));
    std::vector<float> hb((size_t)K * N); for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    for (int cfg = 0; cfg < 2; ++cfg) {
        const uint32_t log2p = cfg == 0 ? 2 : 3, q = cfg == 0 ? 2 : 1, cols_per_part = N / q;
        const uint32_t rb = (M + 7) / 8, rb_chunk = (rb + (1u << log2p) - 1) >> log2p;
        dim3 grid(8 * rb_chunk, cols_per_part / 128 ? cols_per_part / 128 : 1);
        const size_t nwaves = (size_t)grid.x * grid.y * 4;
        CK(hipMalloc(&dst, nwaves * 64)); 
        for (int it = 0; it < 20; ++it) {  // back to back like the bench; the last launch's stamps survive
            hipLaunchKernelGGL(stamped, grid, dim3(256), 0, 0, M, rb_chunk, log2p, cols_per_part, N, N, drp, dci, dva, (uint32_t)((size_t)K * N * 4), dB, dC,
                               (uint32_t)((size_t)M * N * 4), N, dst);
        }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(nwaves * 8);
        CK(hipMemcpy(st.data(), dst, nwaves * 64, hipMemcpyDeviceToHost));
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t w = 0; w < nwaves; ++w) { if (st[w * 8] == 0) continue; tmin = std::min(tmin, st[w * 8]); tmax = std::max(tmax, st[w * 8 + 5]); }
        auto pct = [&](std::vector<double> &v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        std::vector<double> s0, h1, h2, h3, h4, h5, e5;
        for (size_t w = 0; w < nwaves; ++w) {
            const unsigned long long *s = &st[w * 8]; if (s[0] == 0) continue;
            s0.push_back((s[0] - tmin) * 0.01); h1.push_back((s[1] - s[0]) * 0.01); h2.push_back((s[2] - s[1]) * 0.01);
            h3.push_back((s[3] - s[2]) * 0.01); h4.push_back((s[4] - s[3]) * 0.01); h5.push_back((s[5] - s[4]) * 0.01); e5.push_back((s[5] - tmin) * 0.01);
        }
        printf("tiling P=%u Q=%u grid %u x %u, waves %zu, first start -> last end %.2f us\n", 1u << log2p, q, grid.x, grid.y, s0.size(), (tmax - tmin) * 0.01);
        printf("  [us]            p10    p50    p90    max\n");
        auto row = [&](const char *n, std::vector<double> &v) { printf("  %-14s %6.2f %6.2f %6.2f %6.2f\n", n, pct(v, .1), pct(v, .5), pct(v, .9), pct(v, 1.0)); };
        row("wave start", s0); row("rowPtrs hop", h1); row("col/val hop", h2); row("B gather", h3); row("fma+store iss", h4); row("store drain", h5); row("wave end", e5);
        CK(hipFree(dst));
    }
    return 0;
}
