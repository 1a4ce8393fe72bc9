// Runtime slice of the C ABI: device selection, memory, streams, events, graphs.
// Lets the C++ host layer (host/) stay free of HIP headers.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_set>

#include "mispmm_internal.hpp"

namespace mispmm {

static thread_local char g_last_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

int fail(int status, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return status;
}

static thread_local char g_last_kernel[256] = "";

void note_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_kernel, sizeof(g_last_kernel), fmt, ap);
    va_end(ap);
}

XcdGrid xcd_grid(uint32_t nblk) {
    // MISPMM_XCD_REMAP=0 disables the renumbering (A/B measurements only)
    static const bool enabled = [] {
        const char *e = knob_str("MISPMM_XCD_REMAP");
        return !(e && e[0] == '0');
    }();
    if (!enabled || nblk < 16) return {nblk, 0u};
    const uint32_t chunk = (nblk + 7u) / 8u;
    return {chunk * 8u, chunk};
}

}  // namespace mispmm

using namespace mispmm;

extern "C" {

int mispmm_version(void) { return MISPMM_VERSION; }

const char *mispmm_status_string(int status) {
    switch (status) {
        case MISPMM_OK: return "ok";
        case MISPMM_ERR_INVALID_ARG: return "invalid argument";
        case MISPMM_ERR_UNSUPPORTED: return "shape not supported by this kernel";
        case MISPMM_ERR_HIP: return "HIP runtime error";
        case MISPMM_ERR_NO_DEVICE: return "no HIP device";
        case MISPMM_ERR_ALLOC: return "allocation failed";
        default: return "unknown status";
    }
}

const char *mispmm_last_error(void) { return g_last_error; }

const char *mispmm_last_kernel(void) { return g_last_kernel; }

int mispmm_device_count(int *count) {
    if (!count) return fail(MISPMM_ERR_INVALID_ARG, "count is null");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e == hipErrorNoDevice) {
        *count = 0;
        return MISPMM_OK;
    }
    MISPMM_HIP_TRY(e);
    return MISPMM_OK;
}

int mispmm_set_device(int ordinal) {
    MISPMM_HIP_TRY(hipSetDevice(ordinal));
    return MISPMM_OK;
}

int mispmm_get_device(int *ordinal) {
    if (!ordinal) return fail(MISPMM_ERR_INVALID_ARG, "ordinal is null");
    MISPMM_HIP_TRY(hipGetDevice(ordinal));
    return MISPMM_OK;
}

int mispmm_device_info(int ordinal, char *name, int *cu_count, size_t *hbm_bytes) {
    hipDeviceProp_t prop;
    MISPMM_HIP_TRY(hipGetDeviceProperties(&prop, ordinal));
    if (name) {
        snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return MISPMM_OK;
}

int mispmm_device_bus_id(int ordinal, char *bus_id, int len) {
    if (!bus_id || len < 16) return fail(MISPMM_ERR_INVALID_ARG, "device_bus_id: buffer of at least 16 bytes needed");
    MISPMM_HIP_TRY(hipDeviceGetPCIBusId(bus_id, len, ordinal));
    return MISPMM_OK;
}

int mispmm_malloc(void **dev_ptr, size_t bytes) {
    if (!dev_ptr) return fail(MISPMM_ERR_INVALID_ARG, "dev_ptr is null");
    *dev_ptr = nullptr;
    if (bytes == 0) return MISPMM_OK;
    hipError_t e = hipMalloc(dev_ptr, bytes);
    if (e == hipErrorOutOfMemory) return fail(MISPMM_ERR_ALLOC, "hipMalloc(%zu) out of memory", bytes);
    MISPMM_HIP_TRY(e);
    MISPMM_HIP_TRY(hipMemset(*dev_ptr, 0, bytes));
    return MISPMM_OK;
}

int mispmm_free(void *dev_ptr) {
    if (dev_ptr) MISPMM_HIP_TRY(hipFree(dev_ptr));
    return MISPMM_OK;
}

// Pinned host memory needs a device context.  On a machine without a GPU (the CLI's --cpu-only
// path, BASELINE config 1) the allocation degrades to ordinary page-aligned memory; those
// pointers are remembered so mispmm_host_free releases them the right way.
static std::mutex g_plain_mutex;
static std::unordered_set<void *> g_plain_host;

int mispmm_host_alloc(void **host_ptr, size_t bytes) {
    if (!host_ptr) return fail(MISPMM_ERR_INVALID_ARG, "host_ptr is null");
    *host_ptr = nullptr;
    if (bytes == 0) return MISPMM_OK;
    hipError_t e = hipHostMalloc(host_ptr, bytes, hipHostMallocDefault);
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) {
        (void)hipGetLastError();
        void *p = nullptr;
        if (posix_memalign(&p, 4096, bytes) != 0) return fail(MISPMM_ERR_ALLOC, "posix_memalign(%zu) failed", bytes);
        std::lock_guard<std::mutex> lock(g_plain_mutex);
        g_plain_host.insert(p);
        *host_ptr = p;
    } else {
        if (e == hipErrorOutOfMemory) return fail(MISPMM_ERR_ALLOC, "hipHostMalloc(%zu) out of memory", bytes);
        MISPMM_HIP_TRY(e);
    }
    memset(*host_ptr, 0, bytes);
    return MISPMM_OK;
}

int mispmm_host_free(void *host_ptr) {
    if (!host_ptr) return MISPMM_OK;
    {
        std::lock_guard<std::mutex> lock(g_plain_mutex);
        auto it = g_plain_host.find(host_ptr);
        if (it != g_plain_host.end()) {
            g_plain_host.erase(it);
            free(host_ptr);
            return MISPMM_OK;
        }
    }
    MISPMM_HIP_TRY(hipHostFree(host_ptr));
    return MISPMM_OK;
}

static bool copy_kind(int kind, hipMemcpyKind *out) {
    switch (kind) {
        case MISPMM_H2H: *out = hipMemcpyHostToHost; return true;
        case MISPMM_H2D: *out = hipMemcpyHostToDevice; return true;
        case MISPMM_D2H: *out = hipMemcpyDeviceToHost; return true;
        case MISPMM_D2D: *out = hipMemcpyDeviceToDevice; return true;
        default: return false;
    }
}

int mispmm_memcpy(void *dst, const void *src, size_t bytes, int kind) {
    hipMemcpyKind k;
    if (!copy_kind(kind, &k)) return fail(MISPMM_ERR_INVALID_ARG, "bad copy kind %d", kind);
    if (bytes == 0) return MISPMM_OK;
    if (!dst || !src) return fail(MISPMM_ERR_INVALID_ARG, "null pointer in memcpy");
    MISPMM_HIP_TRY(hipMemcpy(dst, src, bytes, k));
    return MISPMM_OK;
}

int mispmm_memcpy_async(void *dst, const void *src, size_t bytes, int kind, mispmm_stream_t stream) {
    hipMemcpyKind k;
    if (!copy_kind(kind, &k)) return fail(MISPMM_ERR_INVALID_ARG, "bad copy kind %d", kind);
    if (bytes == 0) return MISPMM_OK;
    if (!dst || !src) return fail(MISPMM_ERR_INVALID_ARG, "null pointer in memcpy");
    MISPMM_HIP_TRY(hipMemcpyAsync(dst, src, bytes, k, as_stream(stream)));
    return MISPMM_OK;
}

int mispmm_memset_async(void *dev_ptr, int value, size_t bytes, mispmm_stream_t stream) {
    if (bytes == 0) return MISPMM_OK;
    if (!dev_ptr) return fail(MISPMM_ERR_INVALID_ARG, "null pointer in memset");
    MISPMM_HIP_TRY(hipMemsetAsync(dev_ptr, value, bytes, as_stream(stream)));
    return MISPMM_OK;
}

int mispmm_stream_create(mispmm_stream_t *stream) {
    if (!stream) return fail(MISPMM_ERR_INVALID_ARG, "stream is null");
    hipStream_t s;
    MISPMM_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return MISPMM_OK;
}

int mispmm_stream_destroy(mispmm_stream_t stream) {
    if (stream) MISPMM_HIP_TRY(hipStreamDestroy(as_stream(stream)));
    return MISPMM_OK;
}

int mispmm_stream_sync(mispmm_stream_t stream) {
    MISPMM_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return MISPMM_OK;
}

int mispmm_device_sync(void) {
    MISPMM_HIP_TRY(hipDeviceSynchronize());
    return MISPMM_OK;
}

int mispmm_event_create(mispmm_event_t *event) {
    if (!event) return fail(MISPMM_ERR_INVALID_ARG, "event is null");
    hipEvent_t e;
    MISPMM_HIP_TRY(hipEventCreate(&e));
    *event = e;
    return MISPMM_OK;
}

int mispmm_event_destroy(mispmm_event_t event) {
    if (event) MISPMM_HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return MISPMM_OK;
}

int mispmm_event_record(mispmm_event_t event, mispmm_stream_t stream) {
    MISPMM_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return MISPMM_OK;
}

int mispmm_event_sync(mispmm_event_t event) {
    MISPMM_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return MISPMM_OK;
}

int mispmm_event_elapsed_ms(mispmm_event_t start, mispmm_event_t stop, float *ms) {
    if (!ms) return fail(MISPMM_ERR_INVALID_ARG, "ms is null");
    MISPMM_HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return MISPMM_OK;
}

int mispmm_graph_begin(mispmm_stream_t stream) {
    if (!stream) return fail(MISPMM_ERR_INVALID_ARG, "graph capture needs a non-null stream");
    MISPMM_HIP_TRY(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
    return MISPMM_OK;
}

int mispmm_graph_end(mispmm_stream_t stream, mispmm_graph_t *graph) {
    if (!graph) return fail(MISPMM_ERR_INVALID_ARG, "graph is null");
    hipGraph_t g = nullptr;
    MISPMM_HIP_TRY(hipStreamEndCapture(as_stream(stream), &g));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    MISPMM_HIP_TRY(e);
    *graph = exec;
    return MISPMM_OK;
}

int mispmm_graph_launch(mispmm_graph_t graph, mispmm_stream_t stream) {
    MISPMM_HIP_TRY(hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph), as_stream(stream)));
    return MISPMM_OK;
}

int mispmm_graph_destroy(mispmm_graph_t graph) {
    if (graph) MISPMM_HIP_TRY(hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph)));
    return MISPMM_OK;
}

}  // extern "C"
