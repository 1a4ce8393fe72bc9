// Engine orchestration (/root/reference/include/engine.hpp:12-13, src/engine/engine.cpp:16-61).
#pragma once

#include "engine/engine_base.hpp"
#include "engine/engine_bsr.hpp"
#include "engine/engine_coo.hpp"
#include "engine/engine_csr.hpp"
#include "engine/engine_ell.hpp"
#include "engine/vendor.hpp"

namespace cuspmm {

// H2D copies, sequential CPU run (unless skipSeq), every HIP kernel of the engine checked against
// the CPU result, then the vendor cross-check.  abs_tol / rel_tol are accepted for signature
// compatibility; like the reference, the self-check uses ABS_TOL / REL_TOL.
// cpuOnly (new): stop after the CPU run -- BASELINE config 1, "CPU engine path, no GPU".
template <typename EngT>
void runEngine(EngT *engine, typename EngT::MataT *a, typename EngT::MatbT *b, float abs_tol, float rel_tol,
               bool skipSeq, bool cpuOnly = false, const std::string &savePath = "");

}  // namespace cuspmm
