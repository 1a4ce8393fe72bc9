// Prototype (never shipped): does sharing B-row segments through LDS inside row clusters beat the row-gather kernel
// on the headline?  One workgroup = one cluster of R rows x one 64-column part: phase 1 copies the cluster's DISTINCT
// 256-byte B segments into LDS once, phase 2 sums every row from LDS in storage order (fp32 product, double
// accumulate: the REFERENCE numerics).  Input: tools/micro/cluster_plan.py.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/cluster_lds.hip -o /tmp/cluster_lds && /tmp/cluster_lds /tmp/plan.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using u4 = uint32_t __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));

template <int WIDTH>
__global__ __launch_bounds__(256) void cluster_lds(uint32_t R, uint32_t maxdist, const uint32_t *__restrict__ ndist_cols,
                                                   const uint32_t *__restrict__ row_of, const uint16_t *__restrict__ local,
                                                   const float *__restrict__ vals, const float *__restrict__ B, uint32_t ldb,
                                                   float *__restrict__ C, uint32_t ldc) {
    extern __shared__ u4 seg[];  // [ndist][16 lanes] = 256-byte segments
    const uint32_t cluster = blockIdx.x >> 1, q = blockIdx.x & 1u;
    const uint32_t lane = threadIdx.x & 15u, group = threadIdx.x >> 4;
    const uint32_t *head = ndist_cols + static_cast<size_t>(cluster) * (maxdist + 1);
    const uint32_t ndist = head[0];
    const uint32_t *cols = head + 1;
    const float *bpart = B + q * 64u + lane * 4u;
    // phase 1: 16 lane groups, 4 segments each in flight
    for (uint32_t s0 = group; s0 < ndist; s0 += 64) {
        u4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t s = s0 + 16u * u;
            const uint32_t col = cols[min(s, ndist - 1)];
            v[u] = *reinterpret_cast<const u4 *>(bpart + static_cast<size_t>(col) * ldb);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t s = s0 + 16u * u;
            if (s < ndist) seg[s * 16u + lane] = v[u];
        }
    }
    __syncthreads();
    // phase 2: each lane group walks its rows
    for (uint32_t rr = group; rr < R; rr += 16) {
        const uint32_t slot = cluster * R + rr;
        const uint32_t row = row_of[slot];
        if (row == 0xFFFFFFFFu) continue;  // group-uniform
        const uint32_t e = min(lane, static_cast<uint32_t>(WIDTH - 1));
        const uint32_t my_idx = local[static_cast<size_t>(slot) * WIDTH + e];
        const float my_val = vals[static_cast<size_t>(slot) * WIDTH + e];
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < WIDTH; ++s) {
            const uint32_t idx = __shfl(my_idx, s, 16);
            const float a = __shfl(my_val, s, 16);
            const f4 b = __builtin_bit_cast(f4, seg[idx * 16u + lane]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float p = a * b[v];
                acc[v] += static_cast<double>(p);
            }
        }
        const f4 out{static_cast<float>(acc[0]), static_cast<float>(acc[1]), static_cast<float>(acc[2]), static_cast<float>(acc[3])};
        *reinterpret_cast<f4 *>(C + static_cast<size_t>(row) * ldc + q * 64u + lane * 4u) = out;
    }
}

int main(int argc, char **argv) {
    FILE *f = fopen(argc > 1 ? argv[1] : "/tmp/plan.bin", "rb");
    if (!f) { printf("no plan file\n"); return 1; }
    uint32_t h[6];
    if (fread(h, 4, 6, f) != 6) return 1;
    const uint32_t M = h[0], K = h[1], W = h[2], R = h[3], NC = h[4], MD = h[5], N = 128;
    if (W != 14) { printf("prototype is built for width 14\n"); return 1; }
    std::vector<uint32_t> nc(static_cast<size_t>(NC) * (MD + 1)), rowof(static_cast<size_t>(NC) * R);
    std::vector<uint16_t> loc(static_cast<size_t>(NC) * R * W);
    std::vector<float> va(static_cast<size_t>(NC) * R * W);
    if (fread(nc.data(), 4, nc.size(), f) != nc.size() || fread(rowof.data(), 4, rowof.size(), f) != rowof.size() ||
        fread(loc.data(), 2, loc.size(), f) != loc.size() || fread(va.data(), 4, va.size(), f) != va.size()) return 1;
    fclose(f);
    std::vector<float> hb(static_cast<size_t>(K) * N);
    for (size_t i = 0; i < hb.size(); ++i) hb[i] = static_cast<float>((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    uint32_t *dnc, *drow; uint16_t *dloc; float *dva, *dB, *dC;
    CK(hipMalloc(&dnc, nc.size() * 4)); CK(hipMalloc(&drow, rowof.size() * 4)); CK(hipMalloc(&dloc, loc.size() * 2));
    CK(hipMalloc(&dva, va.size() * 4)); CK(hipMalloc(&dB, hb.size() * 4)); CK(hipMalloc(&dC, static_cast<size_t>(M) * N * 4));
    CK(hipMemcpy(dnc, nc.data(), nc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(drow, rowof.data(), rowof.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dloc, loc.data(), loc.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dva, va.data(), va.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dC, 0xff, static_cast<size_t>(M) * N * 4));
    const size_t lds = static_cast<size_t>(MD) * 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(cluster_lds<14>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipStream_t s; CK(hipStreamCreate(&s));
    auto launch = [&] { hipLaunchKernelGGL(cluster_lds<14>, dim3(NC * 2), dim3(256), lds, s, R, MD, dnc, drow, dloc, dva, dB, N, dC, N); };
    launch(); CK(hipStreamSynchronize(s)); CK(hipGetLastError());
    // exact check of every row against the host (fp32 product, double accumulate, storage order)
    std::vector<float> hc(static_cast<size_t>(M) * N);
    CK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, rows = 0;
    for (uint32_t c = 0; c < NC; ++c)
        for (uint32_t rr = 0; rr < R; ++rr) {
            const size_t slot = static_cast<size_t>(c) * R + rr;
            const uint32_t row = rowof[slot];
            if (row == 0xFFFFFFFFu) continue;
            ++rows;
            for (uint32_t j = 0; j < N; ++j) {
                double acc = 0;
                for (uint32_t e = 0; e < W; ++e) {
                    const uint32_t col = nc[static_cast<size_t>(c) * (MD + 1) + 1 + loc[slot * W + e]];
                    const float p = va[slot * W + e] * hb[static_cast<size_t>(col) * N + j];
                    acc += static_cast<double>(p);
                }
                if (static_cast<float>(acc) != hc[static_cast<size_t>(row) * N + j]) ++bad;
            }
        }
    printf("clusters %u x %u rows, max distinct %u (LDS %zu KB), rows checked %zu, mismatching elements %zu\n", NC, R, MD, lds / 1024, rows, bad);
    const int iters = 300;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    printf("cluster_lds: %.3f us per launch (row-gather kernel: 3.6 us)\n", best * 1e3f / iters);
    return bad ? 2 : 0;
}
