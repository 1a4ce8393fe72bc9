// BSR: sequential CPU engine (kernel 0) and the HIP wrapper.
#include "engine/engine_bsr.hpp"
#include "engine/wrapper_common.hpp"

#include <cstring>
#include <vector>

namespace cuspmm {

// Kernel 0: block rows, then blocks in storage order, then block element (i, j), each adding
// blk[i][j] * B[colBase + j, :] into C[rowBase + i, :] in DT
// (/root/reference/src/spmm/bsr/spmm_bsr.cpp:17-38).  C arrives zero-filled.
template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmBSRCpu(SparseMatrixBSR<DT, MT> *ma, DenseMatrix<DT, MT> *mb, DenseMatrix<DT, MT> *mc) {
    assert(!ma->onDevice && !mb->onDevice && !mc->onDevice);
    mb->toOrdering(ORDERING::ROW_MAJOR);
    const size_t n = mb->numCols;
    const MT bR = ma->blockRowSize, bC = ma->blockColSize;
    for (MT R = 0; R < ma->numBlockRows; ++R) {
        for (MT b = ma->blockRowPtrs[R]; b < ma->blockRowPtrs[R + 1]; ++b) {
            const DT *blk = ma->data + (size_t)b * bR * bC;
            const size_t colBase = (size_t)ma->blockColIdxs[b] * bC;
            for (MT i = 0; i < bR; ++i) {
                DT *crow = mc->data + ((size_t)R * bR + i) * n;
                for (MT j = 0; j < bC; ++j) {
                    const DT v = blk[(size_t)i * bC + j];
                    const DT *brow = mb->data + (colBase + j) * n;
                    for (size_t k = 0; k < n; ++k) {
                        const DT prod = v * brow[k];
                        crow[k] += prod;
                    }
                }
            }
        }
    }
    return mc;
}

template <typename DT, typename MT, typename AccT>
DenseMatrix<DT, MT> *spmmBSRWrapper(int kernelNum, SparseMatrixBSR<DT, MT> *a, DenseMatrix<DT, MT> *b,
                                    DenseMatrix<DT, MT> *ref) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        assert(a->onDevice && b->onDevice);
        b->toOrdering(ORDERING::ROW_MAJOR);
        const double n = b->numCols;
        const WrapperShape shape{"BSR", a->numRows, a->numCols, a->numNonZero, 2.0 * a->numElements * n,
                                 a->numElements * 4.0 + a->numBlocks * 4.0 + (a->numBlockRows + 1.0) * 4 +
                                     a->numCols * n * 4 + a->numRows * n * 4};
        // kernel 2 (fp32 MFMA) has fused numerics only; every other id follows AccT
        const int acc = kernelNum == 2 ? MISPMM_ACC_FAST : accModeOf<AccT>();
        if (kernelNum == 3) {
            // the zero-skipping kernel: the non-zero list built at upload (same bits as the CPU engine unless an explicit
            // zero of A meets an Inf / NaN of B: include/mispmm.h, mispmm_bsr_nonzeros_*); flops and bytes of the list
            const WrapperShape nzShape{"BSR", a->numRows, a->numCols, a->numNonZero, 2.0 * a->nzCount * n,
                                       a->nzCount * 8.0 + (a->numRows + 1.0) * 4 + a->numCols * n * 4 + a->numRows * n * 4};
            return runWrapper<DT, MT>(nzShape, kernelNum, b, ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
                if (a->nzSpans && a->nzSpansLong > 0 && a->nzSpansLong < a->numRows) {  // the long rows on the split shape, the short ones by lane groups, one launch
                    const int st = mispmm_rows_hybrid_f32(stream, a->numRows, a->numCols, a->nzCount, a->nzColIdxs, a->nzVals, a->nzSpans,
                                                          a->numRows, a->nzSpansLong, b->data, b->numCols, b->numCols, c, ldc, acc);
                    if (st != MISPMM_ERR_UNSUPPORTED) return st;
                }
                if (a->nzSpans && !a->nzSpansHybridOnly) {  // long rows: longest first
                    const int st = mispmm_rows_split_f32(stream, a->numRows, a->numCols, a->nzCount, a->nzColIdxs, a->nzVals, a->nzSpans,
                                                         a->numRows, b->data, b->numCols, b->numCols, c, ldc, acc);
                    if (st != MISPMM_ERR_UNSUPPORTED) return st;
                }
                return mispmm_bsr_nonzeros_f32(stream, a->numRows, a->numCols, a->nzCount, a->nzRowPtrs, a->nzColIdxs, a->nzVals,
                                               b->data, b->numCols, b->numCols, c, ldc, acc);
            });
        }
        return runWrapper<DT, MT>(shape, kernelNum, b, ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
            return mispmm_bsr_f32(stream, a->numBlockRows, a->numCols, a->blockRowSize, a->blockColSize, a->numBlocks,
                                  a->blockRowPtrs, a->blockColIdxs, a->data, b->data, b->numCols, b->numCols, c, ldc,
                                  kernelNum, acc);
        });
    }
}

namespace {
// fp32 -> bf16 -> fp32, round to nearest even (finite values; a NaN stays a NaN)
inline float bf16Round(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7F800000u) == 0x7F800000u && (u & 0x007FFFFFu)) u |= 0x00400000u;
    else u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    float r;
    std::memcpy(&r, &u, 4);
    return r;
}
}  // namespace

// `--dtype bf16` (BASELINE.json config 4; the reference has no bf16): A and B rounded to bf16, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16.  The self-check reference is the sequential engine run on the ROUNDED operands, so
// `correct` judges the kernels' arithmetic and not the rounding of the inputs.  Two records with "dtype":"bf16":
// kernelType 6 = column-compacted block rows in 4 step slots, a workgroup per block row (mispmm_bsrc_slots_bf16: config
// 4's kernel from round 3), 4 = the same operand walked by one wave per block row (mispmm_bsrc_bf16), 5 = one B panel
// per block (mispmm_bsr_bf16).  hbmGBps / rooflineFrac of each record count the bytes THAT kernel must move.
template <typename DT, typename MT>
void spmmBSRBf16(SparseMatrixBSR<DT, MT> *a, SparseMatrixBSR<DT, MT> *da, DenseMatrix<DT, MT> *b, DenseMatrix<DT, MT> *db) {
    if constexpr (!std::is_same_v<DT, float>) {
        throw std::runtime_error("Not implemented");
    } else {
        assert(!a->onDevice && da->onDevice && !b->onDevice && db->onDevice);
        if (a->blockRowSize != 16) throw std::runtime_error("--dtype bf16 needs 16-row blocks (got " + std::to_string(a->blockRowSize) + ")");
        b->toOrdering(ORDERING::ROW_MAJOR);
        db->toOrdering(ORDERING::ROW_MAJOR);
        const uint32_t N = b->numCols, K = a->numCols;
        // sequential engine on the rounded operands
        SparseMatrixBSR<DT, MT> ar(a, false);
        for (size_t i = 0; i < ar.numElements; ++i) ar.data[i] = bf16Round(ar.data[i]);
        DenseMatrix<DT, MT> br(b, false);
        for (size_t i = 0; i < br.numElements(); ++i) br.data[i] = bf16Round(br.data[i]);
        DenseMatrix<DT, MT> ref(a->numRows, N, false, ORDERING::ROW_MAJOR);
        spmmBSRCpu<DT, MT, double>(&ar, &br, &ref);
        // device operands: bf16 bits of the blocks and of B, the compacted block rows
        uint16_t *blocks16 = allocateBuffer<uint16_t>(a->numElements ? a->numElements : 1, true);
        uint16_t *b16 = allocateBuffer<uint16_t>((size_t)K * N, true);
        mispmmCheckError(mispmm_f32_to_bf16(nullptr, a->numElements, da->data, blocks16));
        mispmmCheckError(mispmm_f32_to_bf16(nullptr, (size_t)K * N, db->data, b16));
        uint32_t nSteps = 0;
        mispmmCheckError(mispmm_bsr_compact_bf16_host(a->numBlockRows, a->blockRowSize, a->blockColSize, a->numBlocks, a->blockRowPtrs,
                                                      a->blockColIdxs, a->data, &nSteps, nullptr, nullptr, nullptr));
        std::vector<uint32_t> sp((size_t)a->numBlockRows + 1), cl((size_t)(nSteps ? nSteps : 1) * 32);
        std::vector<uint16_t> tl((size_t)(nSteps ? nSteps : 1) * 512);
        mispmmCheckError(mispmm_bsr_compact_bf16_host(a->numBlockRows, a->blockRowSize, a->blockColSize, a->numBlocks, a->blockRowPtrs,
                                                      a->blockColIdxs, a->data, &nSteps, sp.data(), cl.data(), tl.data()));
        uint32_t *dsp = allocateBuffer<uint32_t>(sp.size(), true), *dcl = allocateBuffer<uint32_t>(cl.size(), true);
        uint16_t *dtl = allocateBuffer<uint16_t>(tl.size(), true);
        copyBuffer(dsp, true, sp.data(), false, sp.size() * 4);
        copyBuffer(dcl, true, cl.data(), false, cl.size() * 4);
        copyBuffer(dtl, true, tl.data(), false, tl.size() * 2);
        // the same occupied columns in fixed step slots (4 per block row + extra steps)
        uint32_t nSlotSteps = 0, nUsed = 0;
        mispmmCheckError(mispmm_bsr_compact_slots_bf16_host(a->numBlockRows, a->blockRowSize, a->blockColSize, a->numBlocks, a->blockRowPtrs,
                                                            a->blockColIdxs, a->data, &nSlotSteps, &nUsed, nullptr, nullptr, nullptr));
        std::vector<uint32_t> ep((size_t)a->numBlockRows + 1), scl((size_t)(nSlotSteps ? nSlotSteps : 1) * 32);
        std::vector<uint16_t> stl((size_t)(nSlotSteps ? nSlotSteps : 1) * 512);
        mispmmCheckError(mispmm_bsr_compact_slots_bf16_host(a->numBlockRows, a->blockRowSize, a->blockColSize, a->numBlocks, a->blockRowPtrs,
                                                            a->blockColIdxs, a->data, &nSlotSteps, &nUsed, ep.data(), scl.data(), stl.data()));
        uint32_t *dep = allocateBuffer<uint32_t>(ep.size(), true), *dscl = allocateBuffer<uint32_t>(scl.size(), true);
        uint16_t *dstl = allocateBuffer<uint16_t>(stl.size(), true);
        copyBuffer(dep, true, ep.data(), false, ep.size() * 4);
        copyBuffer(dscl, true, scl.data(), false, scl.size() * 4);
        copyBuffer(dstl, true, stl.data(), false, stl.size() * 2);
        const double n = N;
        const double denseBC = a->numCols * n * 2 + a->numRows * n * 4;  // B bf16 + C fp32
        WrapperShape shape{"BSR", a->numRows, a->numCols, a->numNonZero, 2.0 * da->nzCount * n,
                           nSlotSteps * 128.0 + nUsed * 1024.0 + (a->numBlockRows + 1.0) * 4 + denseBC};
        shape.dtype = "bf16";
        delete runWrapper<DT, MT>(shape, 6, db, &ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
            return mispmm_bsrc_slots_bf16(stream, a->numBlockRows, K, nSlotSteps, dep, dscl, dstl, b16, N, N, c, ldc, 0);
        });
        shape.algorithmicBytes = nSteps * 1152.0 + (a->numBlockRows + 1.0) * 4 + denseBC;
        delete runWrapper<DT, MT>(shape, 4, db, &ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
            return mispmm_bsrc_bf16(stream, a->numBlockRows, K, nSteps, dsp, dcl, dtl, b16, N, N, c, ldc, 0);
        });
        shape.algorithmicBytes = a->numElements * 2.0 + a->numBlocks * 4.0 + (a->numBlockRows + 1.0) * 4 + denseBC;
        if (a->blockColSize == 16) {
            delete runWrapper<DT, MT>(shape, 5, db, &ref, [&](float *c, uint32_t ldc, mispmm_stream_t stream) {
                return mispmm_bsr_bf16(stream, a->numBlockRows, K, 16, 16, a->numBlocks, da->blockRowPtrs, da->blockColIdxs, blocks16, b16,
                                       N, N, c, ldc, 0);
            });
        }
        releaseBuffer(blocks16, true);
        releaseBuffer(b16, true);
        releaseBuffer(dsp, true);
        releaseBuffer(dcl, true);
        releaseBuffer(dtl, true);
        releaseBuffer(dep, true);
        releaseBuffer(dscl, true);
        releaseBuffer(dstl, true);
    }
}

template void spmmBSRBf16<float, uint32_t>(SparseMatrixBSR<float, uint32_t> *, SparseMatrixBSR<float, uint32_t> *,
                                           DenseMatrix<float, uint32_t> *, DenseMatrix<float, uint32_t> *);
template void spmmBSRBf16<double, uint32_t>(SparseMatrixBSR<double, uint32_t> *, SparseMatrixBSR<double, uint32_t> *,
                                            DenseMatrix<double, uint32_t> *, DenseMatrix<double, uint32_t> *);

#define CUSPMM_INST(DT)                                                                                              \
    template DenseMatrix<DT, uint32_t> *spmmBSRCpu<DT, uint32_t, double>(SparseMatrixBSR<DT, uint32_t> *,           \
                                                                         DenseMatrix<DT, uint32_t> *,               \
                                                                         DenseMatrix<DT, uint32_t> *);              \
    template DenseMatrix<DT, uint32_t> *spmmBSRWrapper<DT, uint32_t, double>(int, SparseMatrixBSR<DT, uint32_t> *,  \
                                                                             DenseMatrix<DT, uint32_t> *,           \
                                                                             DenseMatrix<DT, uint32_t> *);
CUSPMM_INST(float)
CUSPMM_INST(double)
#undef CUSPMM_INST

}  // namespace cuspmm
