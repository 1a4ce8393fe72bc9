// cuspmm -- the reference's command line (/root/reference/src/main.cu:19-29, 36-44) on MI355X.
//   cuspmm [--bsr] [--coo] [--csr] [--ell] [--cuda] -d <directory> [-h]
// New optional flags (defaults keep the reference's behaviour):
//   --device <n>     GPU ordinal (default 0; the reference hard-codes 7)
//   -k <cols>        use a seeded synthetic dense operand with <cols> columns instead of dense.in
//   --synth <mode>   uniform (default) | exact
//   --iters <n>      also time n back-to-back launches per kernel, replayed from a hipGraph: GFLOP/s, GB/s, roofline fraction
//   --acc <mode>     reference | fast   (default: from the engine's AccT = double -> reference)
//   --cpu-only       run only the sequential CPU engine (no GPU needed)
//   --no-vendor      skip the rocSPARSE cross-check
//   --vendor-bsr     also run the rocSPARSE cross-check for --bsr (square blocks; the reference builds the BSR
//                    descriptor, sparse_bsr.cu:138-160, but never enables the check: engine_bsr.hpp:24)
//   --save <file>    write the last result matrix as text
//   --dtype <t>      fp32 (default) | bf16: with --bsr and 16-row blocks also run the bf16 MFMA kernels (BASELINE config 4)
//   --gpus <n>       (--csr, --ell) also run the product row-sharded over n GPUs of this node: B replicated, C row slabs
//                    gathered over xGMI; --gather first|peer|rccl|none picks how (default first = into device 0)
//   --batch <n>      (--csr) also multiply n dense operands (n device copies of B) by A in ONE launch
//                    (mispmm_csr_batch_f32): one more record, key "batch", steady-state figures per product
#include <getopt.h>

#include <cstdlib>
#include <filesystem>

#include "engine.hpp"
#include "format.hpp"

static void printHelp(const char *prog) {
    std::cout << "Usage: " << prog << " [OPTIONS]\n"
              << "Options:\n"
              << "  --bsr           Process data in Block Sparse Row format\n"
              << "  --coo           Process data in Coordinate format\n"
              << "  --csr           Process data in Compressed Sparse Row format\n"
              << "  --ell           Process data in ELLPACK format\n"
              << "  --cuda          Accepted for compatibility (the GPU path is always on)\n"
              << "  -d <directory>  Data directory\n"
              << "  --device <n>    GPU ordinal (default 0)\n"
              << "  -k <cols>       Synthetic dense operand with <cols> columns instead of dense.in\n"
              << "  --synth <mode>  uniform | exact\n"
              << "  --iters <n>     Steady-state timing: n launches per kernel replayed from a hipGraph\n"
              << "  --acc <mode>    reference | fast\n"
              << "  --cpu-only      Sequential CPU engine only\n"
              << "  --no-vendor     Skip the rocSPARSE cross-check\n"
              << "  --vendor-bsr    rocSPARSE cross-check for --bsr as well (square blocks)\n"
              << "  --save <file>   Save the last result matrix\n"
              << "  --dtype <t>     fp32 | bf16 (with --bsr, 16-row blocks: bf16 MFMA kernels as well)\n"
              << "  --gpus <n>      With --csr / --ell: also run row-sharded over n GPUs (B replicated, C slabs gathered)\n"
              << "  --batch <n>     With --csr: also multiply n dense operands by A in ONE launch (record key \"batch\")\n"
              << "  --gather <how>  first | peer | rccl | none (default first: slabs copied into device 0)\n"
              << "  -h, --help      Display this help message\n";
}

int main(int argc, char *argv[]) {
    std::string dir, savePath, synthMode = "uniform";
    bool wantCoo = false, wantCsr = false, wantBsr = false, wantEll = false, cpuOnly = false, vendorBsr = false;
    int device = 0;
    long synthCols = 0;
    enum { OPT_DEVICE = 1000, OPT_SYNTH, OPT_ITERS, OPT_ACC, OPT_CPU, OPT_NOVENDOR, OPT_SAVE, OPT_VENDORBSR, OPT_GPUS, OPT_GATHER, OPT_DTYPE, OPT_BATCH };
    const option longOpts[] = {{"bsr", no_argument, nullptr, 'B'},           {"coo", no_argument, nullptr, 'O'},
                               {"csr", no_argument, nullptr, 'S'},           {"ell", no_argument, nullptr, 'E'},
                               {"cuda", no_argument, nullptr, 'U'},          {"help", no_argument, nullptr, 'h'},
                               {"device", required_argument, nullptr, OPT_DEVICE},
                               {"synth", required_argument, nullptr, OPT_SYNTH},
                               {"iters", required_argument, nullptr, OPT_ITERS},
                               {"acc", required_argument, nullptr, OPT_ACC},
                               {"cpu-only", no_argument, nullptr, OPT_CPU},
                               {"no-vendor", no_argument, nullptr, OPT_NOVENDOR},
                               {"save", required_argument, nullptr, OPT_SAVE},
                               {"vendor-bsr", no_argument, nullptr, OPT_VENDORBSR},
                               {"gpus", required_argument, nullptr, OPT_GPUS},
                               {"batch", required_argument, nullptr, OPT_BATCH},
                               {"gather", required_argument, nullptr, OPT_GATHER},
                               {"dtype", required_argument, nullptr, OPT_DTYPE},
                               {nullptr, 0, nullptr, 0}};
    int opt;
    while ((opt = getopt_long(argc, argv, "hd:k:", longOpts, nullptr)) != -1) {
        switch (opt) {
            case 'B': wantBsr = true; break;
            case 'O': wantCoo = true; break;
            case 'S': wantCsr = true; break;
            case 'E': wantEll = true; break;
            case 'U': break;
            case 'h': printHelp(argv[0]); return 0;
            case 'd': dir = optarg; break;
            case 'k': synthCols = std::atol(optarg); break;
            case OPT_DEVICE: device = std::atoi(optarg); break;
            case OPT_SYNTH: synthMode = optarg; break;
            case OPT_ITERS: cuspmm::engineOptions().steadyIters = std::atoi(optarg); break;
            case OPT_ACC:
                cuspmm::engineOptions().accOverride =
                    std::string(optarg) == "fast" ? MISPMM_ACC_FAST : MISPMM_ACC_REFERENCE;
                break;
            case OPT_CPU: cpuOnly = true; break;
            case OPT_NOVENDOR: cuspmm::engineOptions().vendorCheck = false; break;
            case OPT_SAVE: savePath = optarg; break;
            case OPT_VENDORBSR: vendorBsr = true; break;
            case OPT_DTYPE: {
                const std::string t = optarg;
                if (t == "bf16") cuspmm::engineOptions().bf16 = true;
                else if (t != "fp32") {
                    std::cerr << "Error: --dtype takes fp32 | bf16\n";
                    return EXIT_FAILURE;
                }
                break;
            }
            case OPT_GPUS: cuspmm::engineOptions().gpus = std::atoi(optarg); break;
            case OPT_BATCH: cuspmm::engineOptions().batch = std::atoi(optarg); break;
            case OPT_GATHER: {
                const std::string g = optarg;
                if (g == "none") cuspmm::engineOptions().gatherMode = MISPMM_GATHER_NONE;
                else if (g == "first") cuspmm::engineOptions().gatherMode = MISPMM_GATHER_TO_FIRST;
                else if (g == "peer") cuspmm::engineOptions().gatherMode = MISPMM_GATHER_ALL_PEER;
                else if (g == "rccl") cuspmm::engineOptions().gatherMode = MISPMM_GATHER_ALL_RCCL;
                else {
                    std::cerr << "Error: --gather takes first | peer | rccl | none\n";
                    return EXIT_FAILURE;
                }
                break;
            }
            default: return 1;  // getopt_long already printed a message
        }
    }
    if (dir.empty() || (!wantCoo && !wantCsr && !wantBsr && !wantEll)) {
        printHelp(argv[0]);
        return EXIT_FAILURE;
    }

    // inputs are found by suffix, as the reference does (main.cu:98-144)
    std::string cooFile, csrFile, bsrFile, denseFile, ellColind, ellValues, ellRowind, ellValuesCm;
    std::error_code ec;
    for (const auto &entry : std::filesystem::directory_iterator(dir, ec)) {
        if (!entry.is_regular_file()) continue;
        const std::string name = entry.path().filename().string(), path = entry.path().string();
        if (endsWith(name, ".coo")) cooFile = path;
        else if (endsWith(name, ".csr")) csrFile = path;
        else if (endsWith(name, ".bsr")) bsrFile = path;
        else if (endsWith(name, "_colind.ell")) ellColind = path;
        else if (endsWith(name, "_values.ell")) ellValues = path;
        else if (endsWith(name, "_rowind.ell")) ellRowind = path;
        else if (endsWith(name, "_values_colmajor.ell")) ellValuesCm = path;
        else if (endsWith(name, "dense.in")) denseFile = path;
    }
    if (ec) {
        std::cerr << "Error: cannot read directory " << dir << ": " << ec.message() << "\n";
        return EXIT_FAILURE;
    }
    auto missing = [&](const char *what) {
        std::cerr << "Error: Missing required files " << what << " in " << dir << "\n";
        std::exit(EXIT_FAILURE);
    };
    if (wantCoo && cooFile.empty()) missing("*.coo");
    if (wantCsr && csrFile.empty()) missing("*.csr");
    if (wantBsr && bsrFile.empty()) missing("*.bsr");
    // the reference insists on the row-major pair too although it loads only the column-major one
    // (main.cu:160-169, 209-213); only what is loaded is required here
    if (wantEll && (ellRowind.empty() || ellValuesCm.empty())) missing("*_rowind.ell and/or *_values_colmajor.ell");
    if (synthCols <= 0 && denseFile.empty()) missing("dense.in (or pass -k <cols> for a synthetic operand)");

    if (!cpuOnly) {
        int count = 0;
        mispmmCheckError(mispmm_device_count(&count));
        if (count == 0) {
            std::cerr << "Error: no HIP device found (use --cpu-only for the sequential engine alone)\n";
            return EXIT_FAILURE;
        }
        mispmmCheckError(mispmm_set_device(device));
    }
    testcase = dir;
    const float abs_tol = 1.0e-3f, rel_tol = 1.0e-2f;
    const bool skipSeq = false;

    try {
        auto run = [&](auto *a, auto *engine) {
            using Mat = cuspmm::DenseMatrix<float, uint32_t>;
            Mat *dense = synthCols > 0 ? Mat::synthetic(a->numCols, (uint32_t)synthCols, 20241218, synthMode == "exact" ? 1 : 0)
                                       : new Mat(denseFile);
            cuspmm::runEngine(engine, a, dense, abs_tol, rel_tol, skipSeq, cpuOnly, savePath);
            delete dense;
            delete engine;
            delete a;
        };
        if (wantCoo) run(new cuspmm::SparseMatrixCOO<float, uint32_t>(cooFile), new cuspmm::EngineCOO<float, uint32_t, double>(dir));
        if (wantCsr) run(new cuspmm::SparseMatrixCSR<float, uint32_t>(csrFile), new cuspmm::EngineCSR<float, uint32_t, double>(dir));
        if (wantBsr) {
            auto *a = new cuspmm::SparseMatrixBSR<float, uint32_t>(bsrFile);
            auto *engine = new cuspmm::EngineBSR<float, uint32_t, double>(dir);
            engine->SUPPORT_CUSPARSE = vendorBsr && a->blockRowSize == a->blockColSize;
            run(a, engine);
        }
        if (wantEll)
            run(new cuspmm::SparseMatrixELL<float, uint32_t>(ellRowind, ellValuesCm), new cuspmm::EngineELL<float, uint32_t, double>(dir));
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << "\n";
        return EXIT_FAILURE;
    }
    return 0;
}
