// Error handling and result comparison for the host layer.  The host never includes HIP headers:
// the device runtime is reached through the C ABI (include/mispmm.h).
#pragma once

#include <cstdio>
#include <cstdlib>

#include "mispmm.h"

// Print-and-exit on a failed runtime call: the CLI behaviour of the reference's cudaCheckError
// (/root/reference/include/cuda_utils.hpp:13-22).  The C ABI itself never exits.
#define mispmmCheckError(ans) ::cuspmm::mispmmAssert((ans), __FILE__, __LINE__)

namespace cuspmm {

inline void mispmmAssert(int status, const char *file, int line) {
    if (status != MISPMM_OK) {
        const char *detail = mispmm_last_error();
        std::fprintf(stderr, "HIP Error: %s (%s) at %s:%d\n", mispmm_status_string(status),
                     (detail && detail[0]) ? detail : "-", file, line);
        std::exit(status < 0 ? -status : status);
    }
}

// |c - ref| <= atol + rtol * |ref| for every element and no NaN: what torch::allclose computes in
// the reference's wrappers (e.g. src/spmm/csr/spmm_csr_k3.cu:97-99).  Host pointers.
template <typename DT>
bool allclose(const DT *c, const DT *ref, size_t n, double rtol, double atol, double *maxAbsErr = nullptr);

}  // namespace cuspmm
