// Internal helpers shared by the translation units of libmispmm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "mispmm.h"

namespace mispmm {

constexpr int kWave = 64;  // CDNA wavefront width; hard-coded on purpose

// thread-local detail string behind mispmm_last_error()
void set_error(const char *fmt, ...);
int fail(int status, const char *fmt, ...);
// thread-local tag of the device kernel the last compute call enqueued (mispmm_last_kernel(): lets a
// measurement tie a PMC figure to the kernel that actually ran)
void note_kernel(const char *fmt, ...);

inline hipStream_t as_stream(mispmm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define MISPMM_HIP_TRY(expr)                                                                          \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return ::mispmm::fail(e_ == hipErrorNoDevice ? MISPMM_ERR_NO_DEVICE : MISPMM_ERR_HIP,     \
                                  "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                                  __LINE__);                                                          \
    } while (0)

// A launch that follows; picks up configuration errors without synchronising.
#define MISPMM_LAUNCH_CHECK() MISPMM_HIP_TRY(hipGetLastError())

// Measurement knobs (MISPMM_* environment variables that pick a kernel variant, a tiling, a depth).  The production
// library is built WITHOUT -DMISPMM_TUNING: every knob IS its default, a constant the compiler folds, and the
// dispatch contains no getenv.  `make tune` builds libmispmm_tune.so with the knobs live; only tools/ sweeps and the
// tests of opt-in kernels load that one (MISPMM_LIB=.../libmispmm_tune.so).  Results never depend on a knob.
#ifdef MISPMM_TUNING
inline int knob_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
inline const char *knob_str(const char *name) { return getenv(name); }
#else
constexpr int knob_int(const char *, int dflt) { return dflt; }
constexpr const char *knob_str(const char *) { return nullptr; }
#endif

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned8(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline uint64_t ceil_div64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// XCD-aware workgroup order (device half: xcd_block() in spmm_common.hpp): blocks to launch and the
// chunk to pass for `nblk` logical row blocks; chunk 0 = keep dispatch order.
struct XcdGrid {
    uint32_t grid, chunk;
};
XcdGrid xcd_grid(uint32_t nblk);

}  // namespace mispmm
