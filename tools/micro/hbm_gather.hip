// What does MI355X deliver for a gather of segments that come from HBM, as a function of the segment shape?
// The SpMM row-gather kernels read, per non-zero, one contiguous piece of a B row: 64 B ... 2 KiB depending on the width of B and
// on how the columns are split over the XCDs.  This micro-benchmark streams a table FAR larger than the 256 MiB Infinity Cache:
// every segment of `seg` bytes (segment-aligned) is read exactly once, in a pseudo-random order (a multiplicative permutation
// of the segment index; the table size is a power of two), `depth` loads of 16 B per lane in flight, `waves` waves per workgroup, one workgroup per CU x k.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/hbm_gather.hip -o tools/micro/hbm_gather && tools/micro/hbm_gather
// Output: TB/s per (segment bytes, order).  `seq` = segments in address order (the streaming ceiling of this loop shape).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using f4 = float __attribute__((ext_vector_type(4)));

// each group of LPS = seg / 16 lanes reads one segment per step; a wave covers 64 / LPS segments per load instruction
template <int DEPTH>
__global__ __launch_bounds__(256) void gather(const f4 *__restrict__ table, uint64_t nseg, uint32_t lps_log2, uint64_t mult, int random,
                                              float *__restrict__ sink) {
    const uint32_t lps = 1u << lps_log2;
    const uint64_t groups_per_wave = 64u >> lps_log2;
    uint64_t gid = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> lps_log2;   // lane-group id
    uint64_t ngroups = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> lps_log2;
    if (random == 2) {   // lane groups of one XCD walk the segment GROUPS (of 8) between them
        gid = ((static_cast<uint64_t>(blockIdx.x >> 3) * blockDim.x + threadIdx.x) >> lps_log2) * 8u;
        ngroups = ((static_cast<uint64_t>(gridDim.x >> 3) * blockDim.x) >> lps_log2) * 8u;
    }
    const uint32_t lane = threadIdx.x & (lps - 1);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    (void)groups_per_wave;
    for (uint64_t s0 = gid; s0 < nseg; s0 += ngroups * DEPTH) {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            uint64_t s = s0 + static_cast<uint64_t>(d) * ngroups;
            if (s >= nseg) s = gid;
            uint64_t seg = random ? (s * mult) & (nseg - 1) : s;           // nseg a power of two, mult odd: a permutation
            // random == 2: the pattern of a COLUMN-PART tiling -- a workgroup on XCD x (blockIdx.x % 8) reads only the x-th of every 8
            // consecutive segments (B rows of 8 segments, each XCD its own slice of every row); every segment is still read once
            if (random == 2) seg = (seg & ~7ull) | (blockIdx.x & 7u);
            v[d] = __builtin_nontemporal_load(table + seg * lps + lane);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

int main(int argc, char **argv) {
    const size_t bytes = (argc > 1 ? atoll(argv[1]) : 2048ll) << 20;     // table size in MiB (default 2 GiB)
    f4 *table;
    float *sink;
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(table, 0, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    printf("# table %zu MiB, every segment read once; TB/s (best of 3)\n", bytes >> 20);
    for (int wg_per_cu : {2}) {
        for (int depth : {8}) {
            printf("## %d workgroups of 256 threads per CU, %d loads of 16 B per lane in flight\n", wg_per_cu, depth);
            for (uint32_t lps_log2 = 2; lps_log2 <= 6; ++lps_log2) {   // 64 B .. 1 KiB segments (one wave instruction: 1 KiB)
                const uint64_t seg_bytes = 16ull << lps_log2, nseg = bytes / seg_bytes;
                const uint64_t mult = 2654435761ull | 1ull;   // odd
                for (int random : {0, 1, 2}) {
                    float best = 1e30f;
                    for (int rep = 0; rep < (bytes <= (128ull << 20) ? 6 : 3); ++rep) {
                        CK(hipEventRecord(a));
                        if (depth == 4) hipLaunchKernelGGL(gather<4>, dim3(256 * wg_per_cu), dim3(256), 0, 0, table, nseg, lps_log2, mult, random, sink);
                        else hipLaunchKernelGGL(gather<8>, dim3(256 * wg_per_cu), dim3(256), 0, 0, table, nseg, lps_log2, mult, random, sink);
                        CK(hipEventRecord(b));
                        CK(hipEventSynchronize(b));
                        float ms;
                        CK(hipEventElapsedTime(&ms, a, b));
                        if (ms < best) best = ms;
                    }
                    printf("segment %5llu B  %s  %.2f TB/s\n", (unsigned long long)seg_bytes, random == 2 ? "random, XCD x reads slice x of every 8 segments" : random ? "random" : "seq   ", bytes / (best * 1e-3) / 1e12);
                }
            }
        }
    }
    return 0;
}
