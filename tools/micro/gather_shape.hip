// Per-CU rate of a row-segment gather as a function of its SHAPE: a wave instruction of 64 lanes x 16 B (or 8 B)
// reads 64 / G segments of G * 16 B each from rows picked at random in a table.  Everything else is fixed:
// 8 loads in flight per lane, rolling refill, 14 loads per "row" like the headline matrix, W waves per CU.
//   hipcc -O3 --offload-arch=gfx950 gather_shape.hip -o gather_shape && ./gather_shape
// Prints GB/s per CU and in total for G = 16 (256 B), 32 (512 B), 64 (1 KiB), for a table that fits the L2s
// and one that only fits the Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));

template <int G, int VEC, int DEPTH>
__global__ __launch_bounds__(256) void gather(uint32_t log2rows, const float *__restrict__ table, uint32_t row_floats,
                                              uint32_t loads_per_wave, float *__restrict__ out) {
    using vec_t = typename std::conditional<VEC == 4, f32x4, f32x2>::type;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x % 64;
    const uint32_t group = lane / G, gl = lane % G;
    constexpr int GROUPS = 64 / G;
    // row picked by a multiplicative hash of (wave, group, step): no index array, so no dependent hop in front of a load
    const uint32_t seed = (wave * GROUPS + group) * 2246822519u + 374761393u;
    auto row_of = [&](uint32_t i) { return ((seed + i) * 2654435761u) >> (32u - log2rows); };
    vec_t acc = {};
    vec_t buf[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        buf[d] = *reinterpret_cast<const vec_t *>(table + static_cast<size_t>(row_of(d)) * row_floats + gl * VEC);
    for (uint32_t i = DEPTH; i < loads_per_wave; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += buf[d];
            asm volatile("" : "+v"(acc) : : "memory");
            buf[d] = *reinterpret_cast<const vec_t *>(table + static_cast<size_t>(row_of(i + d)) * row_floats + gl * VEC);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += buf[d];
    out[static_cast<size_t>(wave) * 64 + lane] = acc[0] + acc[1];
}

template <int G, int VEC, int DEPTH>
static void run(const char *what, uint32_t idx, const float *table, uint32_t row_floats, uint32_t loads, int waves_per_cu, float *out) {
    const int cus = 256, blocks = cus * waves_per_cu / 4;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gather<G, VEC, DEPTH>), dim3(blocks), dim3(256), 0, 0, idx, table, row_floats, loads, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather<G, VEC, DEPTH>), dim3(blocks), dim3(256), 0, 0, idx, table, row_floats, loads, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double bytes = static_cast<double>(blocks) * 4 * loads * 64 * VEC * 4;
    const double gbs = bytes * reps / (ms * 1e-3) / 1e9;
    printf("%-34s G=%2d (%4d B segments) x%d depth %2d, %2d waves/CU: %7.1f GB/s total, %6.1f GB/s per CU, %5.1f B/clk/CU @2.1GHz\n", what, G,
           G * VEC * 4, VEC, DEPTH, waves_per_cu, gbs, gbs / cus, gbs / cus / 2.1);
}

int main() {
    const uint32_t row_floats = 256;          // 1 KiB rows: a segment is the first G*VEC floats of a row
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t rows = pass == 0 ? 2048 : 32768;   // 2 MiB (L2-resident on every XCD) / 32 MiB (Infinity Cache)
        const uint32_t loads = 14 * 64;                    // per wave and group: 64 "matrix rows" of 14 entries
        const int max_waves = 256 * 24;
        const uint32_t idx = pass == 0 ? 11 : 15;          // log2(rows)
        float *table, *out;
        CHECK(hipMalloc(&table, static_cast<size_t>(rows) * row_floats * 4));
        CHECK(hipMemset(table, 0, static_cast<size_t>(rows) * row_floats * 4));
        CHECK(hipMalloc(&out, static_cast<size_t>(max_waves) * 64 * 4));
        const char *what = pass == 0 ? "table 2 MiB (L2)" : "table 32 MiB (Infinity Cache)";
        for (int w : {8, 12, 16, 24}) {
            run<16, 4, 8>(what, idx, table, row_floats, loads, w, out);
            run<32, 4, 8>(what, idx, table, row_floats, loads, w, out);
            run<64, 4, 8>(what, idx, table, row_floats, loads, w, out);
        }
        run<16, 4, 4>(what, idx, table, row_floats, loads, 12, out);
        run<16, 4, 16>(what, idx, table, row_floats, loads, 12, out);
        run<64, 4, 16>(what, idx, table, row_floats, loads, 12, out);
        run<32, 2, 8>(what, idx, table, row_floats, loads, 12, out);
        run<64, 2, 8>(what, idx, table, row_floats, loads, 12, out);
        run<64, 2, 16>(what, idx, table, row_floats, loads, 24, out);
        CHECK(hipFree(table)); CHECK(hipFree(out));
    }
    return 0;
}
