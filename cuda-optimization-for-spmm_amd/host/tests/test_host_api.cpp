// Exercises the C++ host API the way a user of the reference's classes would: constructors,
// copy2Device / copy2Host, toOrdering, toDense, fromDense, runKernel dispatch.
//   test_host_api <golden small_32x32_generated dir> [--gpu]
// Without --gpu only host-side paths run (no device needed).  Exit code 0 = all checks passed.
#include <cmath>

#include "engine.hpp"
#include "format.hpp"

using namespace cuspmm;
using Dense = DenseMatrix<float, uint32_t>;

static int failures = 0;
#define CHECK(cond)                                                                    \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            std::fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                                \
        }                                                                              \
    } while (0)

static bool sameData(Dense *a, Dense *b) {
    if (a->numRows != b->numRows || a->numCols != b->numCols) return false;
    return std::memcmp(a->data, b->data, a->numElements() * sizeof(float)) == 0;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    const bool gpu = argc > 2 && std::string(argv[2]) == "--gpu";
    testcase = dir;

    SparseMatrixCSR<float, uint32_t> csr(dir + "/Hamrle1.csr");
    SparseMatrixCOO<float, uint32_t> coo(dir + "/Hamrle1.coo");
    SparseMatrixBSR<float, uint32_t> bsr4(dir + "/Hamrle1_b4.bsr");
    SparseMatrixELL<float, uint32_t> ell(dir + "/Hamrle1_rowind.ell", dir + "/Hamrle1_values_colmajor.ell");
    Dense b(dir + "/dense.in");
    CHECK(csr.numRows == 32 && csr.numCols == 32 && csr.numNonZero == 98);
    CHECK(coo.numNonZero == 98 && coo.isRowSorted());
    CHECK(bsr4.blockRowSize == 4 && bsr4.numBlockRows == 8 && bsr4.numElements == bsr4.numBlocks * 16);
    CHECK(ell.numCols == 32 && ell.numSlots() == 32u * ell.maxColNnz);
    CHECK(b.numRows == 32 && b.numCols == 32 && b.ordering == ORDERING::ROW_MAJOR && !b.onDevice);

    // every format describes the same matrix
    Dense *d0 = csr.toDense(), *d1 = coo.toDense(), *d2 = bsr4.toDense(), *d3 = ell.toDense();
    CHECK(sameData(d0, d1) && sameData(d0, d2) && sameData(d0, d3));
    auto *rebuilt = SparseMatrixBSR<float, uint32_t>::fromDense(d0, 4, 4);
    Dense *d4 = rebuilt->toDense();
    CHECK(rebuilt->numBlocks == bsr4.numBlocks && sameData(d0, d4));
    bool threw = false;
    try { SparseMatrixBSR<float, uint32_t>::fromDense(d0, 5, 5); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);

    // host toOrdering round trip
    Dense copy(&b, false);
    CHECK(copy.toOrdering(ORDERING::COL_MAJOR) && copy.ordering == ORDERING::COL_MAJOR);
    CHECK(copy.data[ColMjIdx(3, 5, 32)] == b.data[RowMjIdx(3, 5, 32)]);
    CHECK(copy.toOrdering(ORDERING::ROW_MAJOR) && sameData(&copy, &b));

    // the four sequential engines agree (fp32 rounding apart) through the dispatch interface
    EngineCSR<float, uint32_t, double> ecsr(dir);
    EngineCOO<float, uint32_t, double> ecoo(dir);
    EngineBSR<float, uint32_t, double> ebsr(dir);
    EngineELL<float, uint32_t, double> eell(dir);
    CHECK(ecsr.numKernels == MISPMM_CSR_NUM_KERNELS && ecsr.fmt == "CSR" && ecsr.SUPPORT_CUSPARSE);
    CHECK(!eell.SUPPORT_CUSPARSE && ebsr.numKernels == MISPMM_BSR_NUM_KERNELS + 1);  // + the zero-skipping kernel 3
    Dense c0(32, 32, false), c1(32, 32, false), c2(32, 32, false), c3(32, 32, false);
    CHECK(ecsr.runKernel(0, &csr, &b, &c0) == &c0);
    ecoo.runKernel(0, &coo, &b, &c1);
    ebsr.runKernel(0, &bsr4, &b, &c2);
    eell.runKernel(0, &ell, &b, &c3);
    CHECK(allclose<float>(c1.data, c0.data, 1024, 1e-6, 1e-6) && allclose<float>(c2.data, c0.data, 1024, 1e-6, 1e-6));
    CHECK(sameData(&c1, &c3));  // COO row-major order == column-major ELL order, both fp32 +=
    threw = false;
    try { ecsr.runKernel(99, &csr, &b, &c0); } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "Not implemented"; }
    CHECK(threw);
    float nanv[1] = {NAN}, one[1] = {1.f}, near1[1] = {1.009f};
    CHECK(!allclose<float>(nanv, nanv, 1, 1e-2, 1e-3) && allclose<float>(near1, one, 1, 1e-2, 1e-3));

    // synthetic operand: pinned to the Python generator by known values (tests/test_formats.py)
    Dense *syn = Dense::synthetic(3, 5);
    CHECK(syn->data[0] >= -1.f && syn->data[0] < 1.f);
    Dense *syn2 = Dense::synthetic(7, 5);
    CHECK(std::memcmp(syn->data, syn2->data, 15 * sizeof(float)) == 0);  // prefix property
    if (argc > 3) syn2->save2File(argv[3]);

    if (gpu) {
        mispmmCheckError(mispmm_set_device(0));
        auto *dcsr = csr.copy2Device();
        auto *dell = ell.copy2Device();
        auto *dbsr = bsr4.copy2Device();
        auto *dcoo = coo.copy2Device();
        Dense *db = b.copy2Device();
        CHECK(dcsr->onDevice && db->onDevice && dell->rowWidth > 0 && dell->rmColIdxs != nullptr);
        // device toOrdering (transpose kernel) round trip
        CHECK(db->toOrdering(ORDERING::COL_MAJOR));
        Dense *hb = db->copy2Host();
        CHECK(hb->ordering == ORDERING::COL_MAJOR && hb->data[ColMjIdx(7, 2, 32)] == b.data[RowMjIdx(7, 2, 32)]);
        // wrappers accept a column-major B and convert it themselves
        for (int k = -1; k <= ecsr.numKernels; ++k) {
            if (k == 0) continue;
            auto *r = reinterpret_cast<Dense *>(ecsr.runKernel(k, dcsr, db, &c0));
            CHECK(r != nullptr && r->onDevice && db->ordering == ORDERING::ROW_MAJOR);
            Dense *hr = r->copy2Host();
            CHECK(sameData(hr, &c0));  // AccT = double: bit-identical to the sequential engine
            delete hr;
            delete r;
        }
        auto checkFmt = [&](EngineBase &e, void *a, Dense &ref) {
            for (int k = 1; k <= e.numKernels; ++k) {
                auto *r = reinterpret_cast<Dense *>(e.runKernel(k, a, db, &ref));
                if (r == nullptr) continue;  // kernel declined the shape
                Dense *hr = r->copy2Host();
                CHECK(allclose<float>(hr->data, ref.data, 1024, 1e-5, 1e-6));
                delete hr;
                delete r;
            }
        };
        checkFmt(ecoo, dcoo, c1);
        checkFmt(ebsr, dbsr, c2);
        checkFmt(eell, dell, c3);
        delete hb; delete db; delete dcsr; delete dell; delete dbsr; delete dcoo;
    }
    delete d0; delete d1; delete d2; delete d3; delete d4; delete rebuilt; delete syn; delete syn2;
    std::fprintf(stderr, failures ? "test_host_api: %d FAILURES\n" : "test_host_api: ok\n", failures);
    return failures ? 1 : 0;
}
