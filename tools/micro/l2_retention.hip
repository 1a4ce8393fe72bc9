// Does an XCD's L2 keep a buffer from one kernel to the next of a hipGraph?  And what does the first touch of a page cost?
// The SpMM bench loop multiplies the SAME B every launch; rotated over several B's it slows down long before the rotation
// exceeds the 256 MiB Infinity Cache (profiles/r4/operand_sets_sweep.log: 3.53 us with 1 operand set, 4.88 with 2, 5.36 with 4).
// This micro-benchmark separates the candidates with a dependent pointer chase (one lane per XCD, 256 hops per launch, every hop
// a line of its own): per-hop latency = where the line came from.
//   regions = how many distinct buffers the launches of the graph rotate over; spacing = distance between them.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/l2_retention.hip -o tools/micro/l2_retention && tools/micro/l2_retention
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kHops = 256;

__global__ void chase(const uint32_t *__restrict__ region, uint32_t lines, uint32_t *__restrict__ sink) {
    if (threadIdx.x != 0) return;
    // block b starts at a different point of the cycle, so the 8 XCDs chase 8 disjoint stretches
    uint32_t idx = (blockIdx.x * 2053u) % lines;
    for (int i = 0; i < kHops; ++i) idx = __builtin_nontemporal_load(region + static_cast<size_t>(idx) * 32u);   // 128-byte lines
    if (idx == 0xFFFFFFFFu) sink[0] = idx;
}
__global__ void empty_k() {}

int main() {
    const size_t region_bytes = 2u << 20;                       // one region: 2 MiB = 16384 lines of 128 B, a random cycle through them
    const uint32_t lines = region_bytes / 128;
    const size_t pool_bytes = 1024ull << 20;
    uint32_t *pool, *sink;
    CK(hipMalloc(&pool, pool_bytes));
    CK(hipMalloc(&sink, 64));
    std::vector<uint32_t> perm(lines), host(region_bytes / 4, 0);
    std::iota(perm.begin(), perm.end(), 0u);
    std::shuffle(perm.begin(), perm.end(), std::mt19937(7));
    for (uint32_t i = 0; i < lines; ++i) host[static_cast<size_t>(perm[i]) * 32] = perm[(i + 1) % lines];
    for (size_t off = 0; off + region_bytes <= pool_bytes; off += region_bytes)
        CK(hipMemcpy(reinterpret_cast<char *>(pool) + off, host.data(), region_bytes, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto time_graph = [&](int regions, size_t spacing, bool chase_it) -> float {
        const int launches = std::max(regions, 1) * std::max(1, 256 / std::max(regions, 1));
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < launches; ++i) {
            const uint32_t *r = reinterpret_cast<const uint32_t *>(reinterpret_cast<char *>(pool) + static_cast<size_t>(i % regions) * spacing);
            if (chase_it) hipLaunchKernelGGL(chase, dim3(8), dim3(64), 0, s, r, lines, sink);
            else hipLaunchKernelGGL(empty_k, dim3(8), dim3(64), 0, s);
        }
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(a, s);
            for (int i = 0; i < 4; ++i) hipGraphLaunch(ge, s);
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = std::min(best, ms);
        }
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
        return best * 1e3f / (4 * launches);
    };
    const float floor_us = time_graph(1, 0, false);
    printf("# empty kernel, 8 workgroups: %.3f us per launch; below: (launch - that) / %d hops = ns per dependent read\n", floor_us, kHops);
    struct Case { int regions; size_t spacing; const char *what; };
    const Case cases[] = {
        {1, 0, "1 region (the same 2 MiB every launch)"},
        {2, region_bytes, "2 regions, adjacent"},
        {2, 64ull << 20, "2 regions, 64 MiB apart"},
        {4, region_bytes, "4 regions, adjacent (8 MiB)"},
        {16, region_bytes, "16 regions, adjacent (32 MiB)"},
        {16, 16ull << 20, "16 regions, 16 MiB apart"},
        {64, region_bytes, "64 regions, adjacent (128 MiB)"},
        {64, 16ull << 20, "64 regions, 16 MiB apart (1 GiB span)"},
        {256, region_bytes, "256 regions, adjacent (512 MiB: beyond the Infinity Cache)"},
        {512, region_bytes, "512 regions, adjacent (1 GiB)"},
    };
    for (const Case &c : cases) {
        const float t = time_graph(c.regions, c.spacing, true);
        printf("%-62s %7.3f us per launch  %6.1f ns per hop\n", c.what, t, (t - floor_us) * 1e3f / kHops);
    }
    return 0;
}
