// Measures the back-to-back launch interval of kernels shaped like the CSR SpMM launch, to separate
// "kernel boundary" cost from kernel work.  hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_store(float4* c, int n) {  // each thread writes one float4: n float4 total (dirty lines at kernel end)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
using f4 = float __attribute__((ext_vector_type(4)));
__global__ void k_store_nt(float4* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, 4.f}, reinterpret_cast<f4*>(c) + i);
}
using u4 = unsigned __attribute__((ext_vector_type(4)));
__global__ void k_store_sc1(float4* c, int n) {   // write-through buffer store (aux bit 4 = sc1)
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(c, 0, n * 16, 0x00020000);
    if (i < n) __builtin_amdgcn_raw_buffer_store_b128(u4{1u, 2u, 3u, 4u}, rs, i * 16, 0, 16);
}

template <class F> float time_graph(hipStream_t s, int iters, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < iters; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1e3f / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int iters = 500;
    float4* c; const int n = 6300 * 128 / 4;  // C of the headline problem: 3.2 MB
    CK(hipMalloc(&c, n * sizeof(float4)));
    for (int blocks : {256, 788, 1575, 3150, 6300}) {
        for (int threads : {64, 256}) {
            float t = time_graph(s, iters, [&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, s); });
            printf("empty  grid %5d x %3d : %.3f us per launch\n", blocks, threads, t);
        }
    }
    float t1 = time_graph(s, iters, [&] { hipLaunchKernelGGL(k_store, dim3((n + 255) / 256), dim3(256), 0, s, c, n); });
    printf("store 3.2MB plain      : %.3f us per launch\n", t1);
    float t2 = time_graph(s, iters, [&] { hipLaunchKernelGGL(k_store_nt, dim3((n + 255) / 256), dim3(256), 0, s, c, n); });
    printf("store 3.2MB nontemporal: %.3f us per launch\n", t2);
    float t3 = time_graph(s, iters, [&] { hipLaunchKernelGGL(k_store_sc1, dim3((n + 255) / 256), dim3(256), 0, s, c, n); });
    printf("store 3.2MB sc1 buffer : %.3f us per launch\n", t3);
    return 0;
}
