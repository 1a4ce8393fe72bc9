// Row-sharded CSR SpMM over the GPUs of one node, single process: the multi-GPU slice of the C ABI.
//
// New capability -- the reference pins one device (cudaSetDevice(7), /root/reference/src/main.cu:176) and has
// no exchange step of any kind.  Rows of A (hence rows of C) are independent, so device d owns the contiguous
// row range [rowBounds[d], rowBounds[d + 1]) (cut by mispmm_shard_rows_by_nnz_host), holds a replica of B and
// runs the single-GPU kernel on its slice; the only exchange is the gather of the C row slabs:
//   MISPMM_GATHER_TO_FIRST   every device copies its slab into device 0's C          (peer copies over xGMI)
//   MISPMM_GATHER_ALL_PEER   every device copies its slab into every other device's C (all-gather by peer copies)
//   MISPMM_GATHER_ALL_RCCL   equal slabs: ONE in-place ncclAllGather per device; uneven slabs: grouped in-place
//                            ncclBroadcast of each slab from its owner (all-gather-v)
//   MISPMM_GATHER_ALL_RCCL_EQUAL  rows cut into equal chunks of ceil(M / ndev) (the last may be short, C padded to
//                            ndev * chunk rows by the caller): always ONE in-place ncclAllGather per device
// A C with ldc > N keeps its gap columns: peer copies move N columns per row (2-D copies), the RCCL modes refuse it.
// Everything is enqueued on the caller's per-device streams; nothing here synchronises or allocates.
// librccl.so is loaded on first use of mispmm_comm_create, so single-GPU users never pay for it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "mispmm_internal.hpp"

namespace mispmm {

// one launch copies a slab to up to 16 destinations: blockIdx.y picks the destination, 16 bytes per lane,
// grid-stride over the slab.  Destinations may be peer memory (IPC-mapped or peer-enabled): plain stores.
struct ScatterDsts {
    void *p[16];
};

__global__ __launch_bounds__(256) void slab_scatter_kernel(const uint4 *__restrict__ src, size_t n16, ScatterDsts d) {
    uint4 *__restrict__ dst = static_cast<uint4 *>(d.p[blockIdx.y]);
    const size_t stride = static_cast<size_t>(gridDim.x) * 256;
    for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// ---- RCCL, bound at run time --------------------------------------------------------------------------------
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) return;
        auto sym = [&](const char *n) { return dlsym(r.lib, n); };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Broadcast && r.AllGather && r.GetErrorString;
    });
    return r;
}

}  // namespace mispmm

struct mispmm_comm_s {
    std::vector<ncclComm_t> comms;
    std::vector<int> devices;
};

using namespace mispmm;

#define MISPMM_RCCL_TRY(expr)                                                                                    \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess)                                                                                   \
            return fail(MISPMM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

namespace {
// restores the calling thread's current device on every return path
struct DeviceGuard {
    int saved = -1;
    DeviceGuard() { (void)hipGetDevice(&saved); }
    ~DeviceGuard() {
        if (saved >= 0) (void)hipSetDevice(saved);
    }
};
}  // namespace

extern "C" {

int mispmm_slab_scatter(mispmm_stream_t stream, const void *src, size_t bytes, void *const *dsts_host, uint32_t ndst) {
    if (bytes == 0 || ndst == 0) return MISPMM_OK;
    if (!src || !dsts_host) return fail(MISPMM_ERR_INVALID_ARG, "slab_scatter: null pointer");
    if (ndst > 16) return fail(MISPMM_ERR_INVALID_ARG, "slab_scatter: at most 16 destinations (got %u)", ndst);
    if (bytes % 16 != 0 || !aligned16(src)) return fail(MISPMM_ERR_INVALID_ARG, "slab_scatter: src and size must be 16-byte multiples");
    ScatterDsts d{};
    for (uint32_t i = 0; i < ndst; ++i) {
        if (!dsts_host[i] || !aligned16(dsts_host[i])) return fail(MISPMM_ERR_INVALID_ARG, "slab_scatter: destination %u null or unaligned", i);
        d.p[i] = dsts_host[i];
    }
    const size_t n16 = bytes / 16;
    const size_t blocks = (n16 + 255) / 256;
    dim3 grid(static_cast<uint32_t>(blocks < 1024 ? blocks : 1024), ndst);
    hipLaunchKernelGGL(slab_scatter_kernel, grid, dim3(256), 0, as_stream(stream), static_cast<const uint4 *>(src), n16, d);
    MISPMM_LAUNCH_CHECK();
    return MISPMM_OK;
}

int mispmm_enable_peer_access(uint32_t ndev, const int *devices) {
    if (ndev == 0) return MISPMM_OK;
    if (!devices) return fail(MISPMM_ERR_INVALID_ARG, "peer access: devices is null");
    DeviceGuard guard;
    for (uint32_t a = 0; a < ndev; ++a) {
        MISPMM_HIP_TRY(hipSetDevice(devices[a]));
        for (uint32_t b = 0; b < ndev; ++b) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            MISPMM_HIP_TRY(hipDeviceCanAccessPeer(&can, devices[a], devices[b]));
            if (!can) return fail(MISPMM_ERR_UNSUPPORTED, "device %d cannot access device %d", devices[a], devices[b]);
            const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else MISPMM_HIP_TRY(e);
        }
    }
    return MISPMM_OK;
}

int mispmm_comm_create(mispmm_comm_t *comm, uint32_t ndev, const int *devices) {
    if (!comm || !devices || ndev == 0) return fail(MISPMM_ERR_INVALID_ARG, "comm_create: null pointer or no device");
    *comm = nullptr;
    if (!rccl().ok) return fail(MISPMM_ERR_UNSUPPORTED, "comm_create: librccl.so could not be loaded or lacks a symbol");
    auto *c = new mispmm_comm_s;
    c->comms.resize(ndev);
    c->devices.assign(devices, devices + ndev);
    DeviceGuard guard;
    const ncclResult_t r = rccl().CommInitAll(c->comms.data(), static_cast<int>(ndev), devices);
    if (r != ncclSuccess) {
        delete c;
        return fail(MISPMM_ERR_HIP, "ncclCommInitAll failed: %s", rccl().GetErrorString(r));
    }
    *comm = c;
    return MISPMM_OK;
}

int mispmm_comm_destroy(mispmm_comm_t comm) {
    if (!comm) return MISPMM_OK;
    for (ncclComm_t c : comm->comms) (void)rccl().CommDestroy(c);
    delete comm;
    return MISPMM_OK;
}

}  // extern "C"

namespace {

// What the three sharded entry points share: argument checks, then the slab exchange.  `row_bytes` / `ld_bytes` are the bytes
// of one C row that carry values / from one row to the next (fp32 or bf16 C alike); slab d = C rows [bounds[d], bounds[d + 1]).
struct Shards {
    uint32_t ndev;
    const int *devices;
    const mispmm_stream_t *streams;
    const uint32_t *bounds;       // in C rows
    void *const *C;
    size_t row_bytes, ld_bytes;
    int gather_mode;
    mispmm_comm_t comm;
    bool equal_chunks = false, whole_chunks = false;
    uint32_t M = 0, chunk = 0;
};

int check_shards(Shards &sh, const char *who) {
    if (sh.ndev == 0) return fail(MISPMM_ERR_INVALID_ARG, "%s: no device", who);
    if (!sh.devices || !sh.streams || !sh.bounds || !sh.C) return fail(MISPMM_ERR_INVALID_ARG, "%s: null argument array", who);
    if (sh.gather_mode < MISPMM_GATHER_NONE || sh.gather_mode > MISPMM_GATHER_ALL_RCCL_EQUAL)
        return fail(MISPMM_ERR_INVALID_ARG, "%s: unknown gather mode %d", who, sh.gather_mode);
    if (sh.bounds[0] != 0) return fail(MISPMM_ERR_INVALID_ARG, "%s: rowBounds[0] must be 0", who);
    for (uint32_t d = 0; d < sh.ndev; ++d) {
        if (sh.bounds[d + 1] < sh.bounds[d]) return fail(MISPMM_ERR_INVALID_ARG, "%s: rowBounds must be non-decreasing", who);
        if (!sh.C[d]) return fail(MISPMM_ERR_INVALID_ARG, "%s: C of device slot %u is null", who, d);
    }
    if (sh.ld_bytes < sh.row_bytes) return fail(MISPMM_ERR_INVALID_ARG, "%s: ldc smaller than N", who);
    // equal chunks: slab d starts at row d * chunk, only the last may be short (its gather then reads and writes up to
    // ndev * chunk rows of C: the EQUAL mode's contract says the caller allocated them)
    sh.M = sh.bounds[sh.ndev];
    sh.chunk = ceil_div(sh.M, sh.ndev);
    sh.equal_chunks = true;
    for (uint32_t d = 0; d < sh.ndev; ++d) sh.equal_chunks = sh.equal_chunks && sh.bounds[d] == std::min(sh.M, d * sh.chunk);
    sh.whole_chunks = sh.equal_chunks && static_cast<uint64_t>(sh.chunk) * sh.ndev == sh.M;
    if (sh.gather_mode == MISPMM_GATHER_ALL_RCCL || sh.gather_mode == MISPMM_GATHER_ALL_RCCL_EQUAL) {
        if (!sh.comm || sh.comm->comms.size() != sh.ndev)
            return fail(MISPMM_ERR_INVALID_ARG, "%s: RCCL gather needs a communicator over the same %u devices", who, sh.ndev);
        for (uint32_t d = 0; d < sh.ndev; ++d)
            if (sh.comm->devices[d] != sh.devices[d]) return fail(MISPMM_ERR_INVALID_ARG, "%s: communicator device order differs", who);
        // a collective moves whole contiguous runs: with ldc > N it would carry the gap columns of every row from the
        // owner into everybody's C
        if (sh.ld_bytes != sh.row_bytes) return fail(MISPMM_ERR_UNSUPPORTED, "%s: the RCCL gathers need a dense C (ldc == N); use a peer gather", who);
        if (sh.gather_mode == MISPMM_GATHER_ALL_RCCL_EQUAL && !sh.equal_chunks)
            return fail(MISPMM_ERR_INVALID_ARG, "%s: GATHER_ALL_RCCL_EQUAL needs rowBounds[d] = d * ceil(M / ndev)", who);
    }
    return MISPMM_OK;
}

// exchange the slabs (stream order on the owner's stream: the copy follows the kernel that wrote the slab)
int gather_slabs(const Shards &sh) {
    if (sh.gather_mode == MISPMM_GATHER_NONE || sh.row_bytes == 0) return MISPMM_OK;
    auto at = [&](uint32_t d, uint32_t row) { return static_cast<char *>(sh.C[d]) + static_cast<size_t>(row) * sh.ld_bytes; };
    if (sh.gather_mode == MISPMM_GATHER_TO_FIRST || sh.gather_mode == MISPMM_GATHER_ALL_PEER) {
        for (uint32_t d = 0; d < sh.ndev; ++d) {
            const uint32_t r0 = sh.bounds[d], rows = sh.bounds[d + 1] - r0;
            if (rows == 0) continue;
            MISPMM_HIP_TRY(hipSetDevice(sh.devices[d]));
            const uint32_t last = sh.gather_mode == MISPMM_GATHER_TO_FIRST ? 1u : sh.ndev;
            for (uint32_t e = 0; e < last; ++e) {
                if (e == d || sh.C[e] == sh.C[d]) continue;
                if (sh.ld_bytes == sh.row_bytes) {  // dense C: the slab is one contiguous run
                    MISPMM_HIP_TRY(hipMemcpyPeerAsync(at(e, r0), sh.devices[e], at(d, r0), sh.devices[d], static_cast<size_t>(rows) * sh.row_bytes,
                                                      as_stream(sh.streams[d])));
                } else {  // strided C: N columns of every row, the gap columns of the destination stay as they are
                    MISPMM_HIP_TRY(hipMemcpy2DAsync(at(e, r0), sh.ld_bytes, at(d, r0), sh.ld_bytes, sh.row_bytes, rows, hipMemcpyDeviceToDevice,
                                                    as_stream(sh.streams[d])));
                }
            }
        }
        return MISPMM_OK;
    }
    if (sh.equal_chunks && (sh.whole_chunks || sh.gather_mode == MISPMM_GATHER_ALL_RCCL_EQUAL)) {
        // ONE in-place all-gather per device over equal slabs (device d's slab already sits at chunk d of its own C)
        const size_t count = static_cast<size_t>(sh.chunk) * sh.row_bytes;
        MISPMM_RCCL_TRY(rccl().GroupStart());
        for (uint32_t d = 0; d < sh.ndev; ++d) {
            const ncclResult_t r = rccl().AllGather(static_cast<char *>(sh.C[d]) + static_cast<size_t>(d) * count, sh.C[d], count, ncclInt8,
                                                    sh.comm->comms[d], as_stream(sh.streams[d]));
            if (r != ncclSuccess) {
                (void)rccl().GroupEnd();
                return fail(MISPMM_ERR_HIP, "ncclAllGather failed: %s", rccl().GetErrorString(r));
            }
        }
        MISPMM_RCCL_TRY(rccl().GroupEnd());
        return MISPMM_OK;
    }
    // uneven slabs (nnz-balanced row ranges): all-gather-v as grouped in-place broadcasts, one per slab
    MISPMM_RCCL_TRY(rccl().GroupStart());
    for (uint32_t root = 0; root < sh.ndev; ++root) {
        const uint32_t r0 = sh.bounds[root], rows = sh.bounds[root + 1] - r0;
        if (rows == 0) continue;
        const size_t count = static_cast<size_t>(rows) * sh.row_bytes;
        for (uint32_t d = 0; d < sh.ndev; ++d) {
            const ncclResult_t r = rccl().Broadcast(at(d, r0), at(d, r0), count, ncclInt8, static_cast<int>(root), sh.comm->comms[d],
                                                    as_stream(sh.streams[d]));
            if (r != ncclSuccess) {
                (void)rccl().GroupEnd();
                return fail(MISPMM_ERR_HIP, "ncclBroadcast failed: %s", rccl().GetErrorString(r));
            }
        }
    }
    MISPMM_RCCL_TRY(rccl().GroupEnd());
    return MISPMM_OK;
}

}  // namespace

extern "C" {

int mispmm_multi_csr_f32(uint32_t ndev, const int *devices, const mispmm_stream_t *streams, const uint32_t *rowBounds_host,
                         uint32_t K, const uint32_t *const *rowPtrs, const uint32_t *const *colIdxs,
                         const float *const *vals, const uint32_t *nnz_host, const uint32_t *uniformRowNnz_host,
                         const float *const *B, uint32_t N, uint32_t ldb, float *const *C, uint32_t ldc, int kernel,
                         int acc_mode, int gather_mode, mispmm_comm_t comm) {
    if (!rowPtrs || !colIdxs || !vals || !nnz_host || !B) return fail(MISPMM_ERR_INVALID_ARG, "multi: null argument array");
    Shards sh{ndev, devices, streams, rowBounds_host, reinterpret_cast<void *const *>(C), static_cast<size_t>(N) * 4u, static_cast<size_t>(ldc) * 4u,
              gather_mode, comm};
    if (int st = check_shards(sh, "multi")) return st;
    for (uint32_t d = 0; d < ndev; ++d)
        if (!B[d]) return fail(MISPMM_ERR_INVALID_ARG, "multi: B of device slot %u is null", d);
    DeviceGuard guard;
    // every device multiplies its row range into its own rows of its C
    for (uint32_t d = 0; d < ndev; ++d) {
        const uint32_t r0 = rowBounds_host[d], rows = rowBounds_host[d + 1] - r0;
        if (rows == 0 || N == 0) continue;
        MISPMM_HIP_TRY(hipSetDevice(devices[d]));
        float *slab = C[d] + static_cast<size_t>(r0) * ldc;
        const uint32_t w = uniformRowNnz_host ? uniformRowNnz_host[d] : 0u;
        int st = MISPMM_ERR_UNSUPPORTED;
        if (w != 0 && (kernel == MISPMM_KERNEL_AUTO || kernel == 5))
            st = mispmm_csr_uniform_f32(streams[d], rows, K, w, colIdxs[d], vals[d], B[d], N, ldb, slab, ldc, acc_mode);
        if (st == MISPMM_ERR_UNSUPPORTED)
            st = mispmm_csr_f32(streams[d], rows, K, nnz_host[d], rowPtrs[d], colIdxs[d], vals[d], B[d], N, ldb, slab, ldc, kernel,
                                acc_mode);
        if (st != MISPMM_OK) return st;
    }
    return gather_slabs(sh);
}

// ELL sharded by rows ("ELL by rows"): device slot d holds rows [rowBounds[d], rowBounds[d + 1]) of the
// ROW-MAJOR ELL (its own [rows x width] index and value arrays), a replica of B and a buffer for the full C.
int mispmm_multi_ell_f32(uint32_t ndev, const int *devices, const mispmm_stream_t *streams, const uint32_t *rowBounds_host, uint32_t K,
                         uint32_t width, const uint32_t *const *colIdxs, const float *const *vals, const float *const *B, uint32_t N,
                         uint32_t ldb, float *const *C, uint32_t ldc, int kernel, int acc_mode, int gather_mode, mispmm_comm_t comm) {
    if (!colIdxs || !vals || !B) return fail(MISPMM_ERR_INVALID_ARG, "multi_ell: null argument array");
    Shards sh{ndev, devices, streams, rowBounds_host, reinterpret_cast<void *const *>(C), static_cast<size_t>(N) * 4u, static_cast<size_t>(ldc) * 4u,
              gather_mode, comm};
    if (int st = check_shards(sh, "multi_ell")) return st;
    for (uint32_t d = 0; d < ndev; ++d)
        if (!B[d]) return fail(MISPMM_ERR_INVALID_ARG, "multi_ell: B of device slot %u is null", d);
    DeviceGuard guard;
    for (uint32_t d = 0; d < ndev; ++d) {
        const uint32_t r0 = rowBounds_host[d], rows = rowBounds_host[d + 1] - r0;
        if (rows == 0 || N == 0) continue;
        MISPMM_HIP_TRY(hipSetDevice(devices[d]));
        const int st = mispmm_ell_f32(streams[d], rows, K, width, colIdxs[d], vals[d], B[d], N, ldb, C[d] + static_cast<size_t>(r0) * ldc, ldc, kernel,
                                      acc_mode);
        if (st != MISPMM_OK) return st;
    }
    return gather_slabs(sh);
}

// BSR sharded by BLOCK rows ("BSR shards by block-rows"), in the layout of BASELINE config 4: device
// slot d holds block rows [blockRowBounds[d], blockRowBounds[d + 1]) as column-compacted bf16 block rows in fixed step slots
// (its own mispmm_bsr_compact_slots_bf16_host output), a replica of the bf16 B and a buffer for the full C (fp32 or bf16).
int mispmm_multi_bsrc_slots_bf16(uint32_t ndev, const int *devices, const mispmm_stream_t *streams, const uint32_t *blockRowBounds_host,
                                 uint32_t K, const uint32_t *nSteps_host, const uint32_t *const *extraPtrs, const uint32_t *const *cols,
                                 const uint16_t *const *tiles, const uint16_t *const *B, uint32_t N, uint32_t ldb, void *const *C, uint32_t ldc,
                                 int c_bf16, int gather_mode, mispmm_comm_t comm) {
    if (!nSteps_host || !extraPtrs || !cols || !tiles || !B || !blockRowBounds_host) return fail(MISPMM_ERR_INVALID_ARG, "multi_bsrc: null argument array");
    if (ndev == 0 || ndev > 64) return fail(MISPMM_ERR_INVALID_ARG, "multi_bsrc: 1 .. 64 device slots");
    const size_t esz = c_bf16 ? 2u : 4u;
    uint32_t rows[65];                                          // the slabs in C ROWS: 16 per block row
    for (uint32_t d = 0; d <= ndev; ++d) {
        if (static_cast<uint64_t>(blockRowBounds_host[d]) * 16u > 0xFFFFFFFFull) return fail(MISPMM_ERR_INVALID_ARG, "multi_bsrc: more than 2^32 rows");
        rows[d] = blockRowBounds_host[d] * 16u;
    }
    Shards sh{ndev, devices, streams, rows, C, static_cast<size_t>(N) * esz, static_cast<size_t>(ldc) * esz, gather_mode, comm};
    if (int st = check_shards(sh, "multi_bsrc")) return st;
    for (uint32_t d = 0; d < ndev; ++d)
        if (!B[d]) return fail(MISPMM_ERR_INVALID_ARG, "multi_bsrc: B of device slot %u is null", d);
    DeviceGuard guard;
    for (uint32_t d = 0; d < ndev; ++d) {
        const uint32_t br0 = blockRowBounds_host[d], brows = blockRowBounds_host[d + 1] - br0;
        if (brows == 0 || N == 0) continue;
        MISPMM_HIP_TRY(hipSetDevice(devices[d]));
        void *slab = static_cast<char *>(C[d]) + static_cast<size_t>(br0) * 16u * ldc * esz;
        const int st = mispmm_bsrc_slots_bf16(streams[d], brows, K, nSteps_host[d], extraPtrs[d], cols[d], tiles[d], B[d], N, ldb, slab, ldc, c_bf16);
        if (st != MISPMM_OK) return st;
    }
    return gather_slabs(sh);
}

}  // extern "C"
