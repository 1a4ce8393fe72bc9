// Result record and tolerances of the engine.  Same record shape as the reference's reportTime
// (/root/reference/include/utils.hpp:24-49: one pseudo-JSON object followed by a comma, every
// value a quoted string, keys testcase .. sequentialTimeMs in that order) so scripts that grep
// the reference's output keep working; MI355X-side measurements are appended as extra keys.
#pragma once

#include <cstdint>
#include <iostream>
#include <string>

// set by main (the directory being processed), read by every record
extern std::string testcase;

// tolerances of the in-binary self-check (reference: include/utils.hpp:10-11)
#define REL_TOL 1e-2f
#define ABS_TOL 1e-3f

inline bool endsWith(const std::string &s, const std::string &suffix) {
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

// Optional steady-state figures a wrapper attaches to its record (all zero = not measured).
struct SteadyStats {
    int iters = 0;
    double usPerSpmm = 0;    // HIP-event time per launch over `iters` back-to-back launches
    double gflops = 0;       // 2 * nnz * N / time
    double hbmGBps = 0;      // algorithmic bytes / time
    double rooflineFrac = 0; // hbmGBps / 8000 (MI355X HBM3E peak; times ngpus for a sharded run)
    int ngpus = 0;           // > 0: the record is a row-sharded multi-GPU run over this many devices
    const char *dtype = nullptr;  // set when the kernel did not compute in the engine's DT (e.g. "bf16")
    int batch = 0;           // > 0 (`--batch n`): one launch multiplied this many dense operands; times are per product
    std::string kernelTag;   // mispmm_last_kernel() of the launch the record times: several kernel ids may share one device kernel
};

// `ordering`: 0 = ROW_MAJOR.  `kernelNum`: 0 = sequential CPU engine, -1 = vendor library.
void reportTime(const std::string &testcase, uint32_t aNumRows, uint32_t aNumCols, uint32_t aNumNonZero,
                const std::string &format, int ordering, int kernelNum, double pro, double kernel, double epilog,
                bool correct, const SteadyStats *steady = nullptr);
