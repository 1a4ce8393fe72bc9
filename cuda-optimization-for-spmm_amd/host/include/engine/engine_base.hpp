// Dispatch interface of the engines (/root/reference/include/engine/engine_base.hpp:5-10).
#pragma once

namespace cuspmm {

class EngineBase {
  public:
    int numKernels = 0;
    virtual ~EngineBase() = default;
    // num 0: sequential CPU engine (host operands); num >= 1: HIP kernel `num` (device a, b; host
    // reference result as third argument); num -1: the format's default HIP kernel.
    virtual void *runKernel(int num, void *_ma, void *_mb, void *_mc) = 0;
};

// Knobs shared by every wrapper, set once by the CLI.  Defaults reproduce the reference's flow
// (one un-warmed launch per kernel, accumulate mode from the engine's AccT).
struct EngineOptions {
    int steadyIters = 0;     // > 0: additionally time this many back-to-back launches with HIP events
    int accOverride = -1;    // -1: from AccT; else MISPMM_ACC_REFERENCE / MISPMM_ACC_FAST
    bool vendorCheck = true; // run (and compare!) the rocSPARSE SpMM where the format supports it
    int gpus = 0;            // > 0 (`--gpus n`): also run the CSR SpMM row-sharded over n devices of this node
    int gatherMode = 1;      // mispmm_gather_mode of that run (default MISPMM_GATHER_TO_FIRST)
    int batch = 0;           // > 1 (`--batch n`, CSR): also multiply n dense operands by A in ONE launch (mispmm_csr_batch_f32)
    bool bf16 = false;       // `--dtype bf16` (BSR with 16-row blocks): also run the bf16 MFMA kernels (BASELINE config 4)
};
EngineOptions &engineOptions();

}  // namespace cuspmm
