#include "formats/sparse_bsr.hpp"

#include <vector>

namespace cuspmm {

template <typename DT, typename MT> SparseMatrixBSR<DT, MT>::SparseMatrixBSR(std::string filePath) {
    std::ifstream in(filePath);
    if (!in.is_open()) {
        std::cerr << "File " << filePath << " doesn't exist!" << std::endl;
        throw std::runtime_error("cannot open " + filePath);
    }
    in >> this->numRows >> this->numCols >> this->numNonZero >> this->blockRowSize >> this->blockColSize >>
        this->numBlocks;
    if (in.fail() || this->blockRowSize == 0 || this->blockColSize == 0)
        throw std::runtime_error(filePath + ": malformed .bsr header");
    this->numBlockRows = this->numRows / this->blockRowSize;
    this->numElements = this->numBlocks * this->blockRowSize * this->blockColSize;
    this->allocateSpace(false);
    for (size_t i = 0; i <= this->numBlockRows; ++i) in >> this->blockRowPtrs[i];
    for (size_t i = 0; i < this->numBlocks; ++i) in >> this->blockColIdxs[i];
    for (size_t i = 0; i < this->numElements; ++i) in >> this->data[i];
    if (in.fail()) throw std::runtime_error(filePath + ": truncated or malformed .bsr file");
    this->assertCheck();
}

template <typename DT, typename MT>
SparseMatrixBSR<DT, MT>::SparseMatrixBSR(MT numRows, MT numCols, MT numNonZero, MT blockRowSize, MT blockColSize,
                                         MT numBlocks, bool onDevice) {
    this->numRows = numRows;
    this->numCols = numCols;
    this->numNonZero = numNonZero;
    this->blockRowSize = blockRowSize;
    this->blockColSize = blockColSize;
    this->numBlocks = numBlocks;
    this->numBlockRows = numRows / blockRowSize;
    this->numElements = numBlocks * blockRowSize * blockColSize;
    this->allocateSpace(onDevice);
    this->assertCheck();
}

template <typename DT, typename MT>
SparseMatrixBSR<DT, MT>::SparseMatrixBSR(SparseMatrixBSR<DT, MT> *target, bool onDevice)
    : SparseMatrixBSR(target->numRows, target->numCols, target->numNonZero, target->blockRowSize, target->blockColSize,
                      target->numBlocks, onDevice) {
    this->copyData(target, onDevice);
}

template <typename DT, typename MT> SparseMatrixBSR<DT, MT>::~SparseMatrixBSR() {
    releaseBuffer(this->blockRowPtrs, this->onDevice);
    releaseBuffer(this->blockColIdxs, this->onDevice);
    releaseBuffer(this->data, this->onDevice);
    releaseBuffer(this->nzRowPtrs, true);
    releaseBuffer(this->nzSpans, true);
    releaseBuffer(this->nzColIdxs, true);
    releaseBuffer(this->nzVals, true);
}

template <typename DT, typename MT> bool SparseMatrixBSR<DT, MT>::allocateSpace(bool onDevice) {
    assert(this->data == nullptr && this->blockRowPtrs == nullptr && this->blockColIdxs == nullptr);
    this->blockRowPtrs = allocateBuffer<MT>((size_t)this->numBlockRows + 1, onDevice);
    this->blockColIdxs = allocateBuffer<MT>(this->numBlocks, onDevice);
    this->data = allocateBuffer<DT>(this->numElements, onDevice);
    this->onDevice = onDevice;
    return true;
}

template <typename DT, typename MT> bool SparseMatrixBSR<DT, MT>::copyData(SparseMatrixBSR<DT, MT> *source, bool onDevice) {
    assert(onDevice == this->onDevice);
    (void)onDevice;
    this->assertSameShape(source);
    copyBuffer(this->blockRowPtrs, this->onDevice, source->blockRowPtrs, source->onDevice,
               ((size_t)this->numBlockRows + 1) * sizeof(MT));
    copyBuffer(this->blockColIdxs, this->onDevice, source->blockColIdxs, source->onDevice,
               (size_t)this->numBlocks * sizeof(MT));
    copyBuffer(this->data, this->onDevice, source->data, source->onDevice, (size_t)this->numElements * sizeof(DT));
    return true;
}

template <typename DT, typename MT> SparseMatrixBSR<DT, MT> *SparseMatrixBSR<DT, MT>::copy2Device() {
    assert(!this->onDevice && this->data != nullptr);
    auto *d = new SparseMatrixBSR<DT, MT>(this, true);
    if constexpr (std::is_same_v<DT, float>) {
        // once per upload: the non-zero entries in the reference's order of addition, for the zero-skipping kernel
        uint32_t nz = 0;
        mispmmCheckError(mispmm_bsr_nonzeros_host(this->numBlockRows, this->blockRowSize, this->blockColSize, this->numBlocks,
                                                  this->blockRowPtrs, this->blockColIdxs, this->data, &nz, nullptr, nullptr,
                                                  nullptr));
        std::vector<uint32_t> rp((size_t)this->numRows + 1), ci(nz ? nz : 1);
        std::vector<float> va(nz ? nz : 1);
        mispmmCheckError(mispmm_bsr_nonzeros_host(this->numBlockRows, this->blockRowSize, this->blockColSize, this->numBlocks,
                                                  this->blockRowPtrs, this->blockColIdxs, this->data, &nz, rp.data(), ci.data(),
                                                  va.data()));
        d->nzCount = nz;
        d->nzRowPtrs = allocateBuffer<MT>(rp.size(), true);
        d->nzColIdxs = allocateBuffer<MT>(ci.size(), true);
        d->nzVals = allocateBuffer<DT>(va.size(), true);
        copyBuffer(d->nzRowPtrs, true, rp.data(), false, rp.size() * sizeof(MT));
        copyBuffer(d->nzColIdxs, true, ci.data(), false, ci.size() * sizeof(MT));
        copyBuffer(d->nzVals, true, va.data(), false, va.size() * sizeof(DT));
        bool hybridOnly = false;
        if (wantsRowSpans(this->numRows, rp.data(), hybridOnly)) {
            uint32_t count = 0;
            uint32_t longCount = 0;
            d->nzSpans = uploadRowSpans(this->numRows, rp.data(), 0xFFFFFFFFu, count, &longCount);
            d->nzSpansLong = longCount;
            d->nzSpansHybridOnly = hybridOnly;
        }
    }
    return d;
}

template <typename DT, typename MT> void SparseMatrixBSR<DT, MT>::assertCheck() {
    if (this->blockRowSize == 0 || this->blockColSize == 0 || this->numRows % this->blockRowSize != 0 ||
        this->numCols % this->blockColSize != 0)
        throw std::runtime_error("BSR: matrix shape is not a multiple of the block shape");
}

template <typename DT, typename MT> void SparseMatrixBSR<DT, MT>::assertSameShape(SparseMatrixBSR<DT, MT> *t) {
    assert(this->blockRowSize == t->blockRowSize && this->blockColSize == t->blockColSize &&
           this->numBlocks == t->numBlocks && this->numBlockRows == t->numBlockRows && this->numRows == t->numRows &&
           this->numCols == t->numCols && this->numNonZero == t->numNonZero);
    (void)t;
}

template <typename DT, typename MT>
SparseMatrixBSR<DT, MT> *SparseMatrixBSR<DT, MT>::fromDense(DenseMatrix<DT, MT> *dense, MT bR, MT bC) {
    assert(!dense->onDevice && dense->ordering == ORDERING::ROW_MAJOR);
    if (bR == 0 || bC == 0 || dense->numRows % bR != 0 || dense->numCols % bC != 0)
        throw std::runtime_error("BSR: matrix shape is not a multiple of the block shape");
    const MT nbr = dense->numRows / bR, nbc = dense->numCols / bC;
    std::vector<MT> ptrs(nbr + 1, 0), idxs;
    for (MT R = 0; R < nbr; ++R) {
        for (MT Cb = 0; Cb < nbc; ++Cb) {
            bool any = false;
            for (MT i = 0; i < bR && !any; ++i)
                for (MT j = 0; j < bC && !any; ++j)
                    any = dense->data[RowMjIdx(R * bR + i, Cb * bC + j, dense->numCols)] != DT(0);
            if (any) idxs.push_back(Cb);
        }
        ptrs[R + 1] = (MT)idxs.size();
    }
    const MT nb = (MT)idxs.size();
    auto *m = new SparseMatrixBSR<DT, MT>(dense->numRows, dense->numCols, nb * bR * bC, bR, bC, nb, false);
    std::memcpy(m->blockRowPtrs, ptrs.data(), ptrs.size() * sizeof(MT));
    if (nb) std::memcpy(m->blockColIdxs, idxs.data(), idxs.size() * sizeof(MT));
    for (MT R = 0; R < nbr; ++R)
        for (MT b = ptrs[R]; b < ptrs[R + 1]; ++b)
            for (MT i = 0; i < bR; ++i)
                for (MT j = 0; j < bC; ++j)
                    m->data[((size_t)b * bR + i) * bC + j] =
                        dense->data[RowMjIdx(R * bR + i, idxs[b] * bC + j, dense->numCols)];
    return m;
}

template <typename DT, typename MT> DenseMatrix<DT, MT> *SparseMatrixBSR<DT, MT>::toDense() {
    assert(!this->onDevice);
    auto *dm = new DenseMatrix<DT, MT>(this->numRows, this->numCols, false);
    for (MT R = 0; R < this->numBlockRows; ++R)
        for (MT b = this->blockRowPtrs[R]; b < this->blockRowPtrs[R + 1]; ++b)
            for (MT i = 0; i < this->blockRowSize; ++i)
                for (MT j = 0; j < this->blockColSize; ++j)
                    dm->data[RowMjIdx(R * this->blockRowSize + i, this->blockColIdxs[b] * this->blockColSize + j,
                                      this->numCols)] =
                        this->data[((size_t)b * this->blockRowSize + i) * this->blockColSize + j];
    return dm;
}

template class SparseMatrixBSR<float, uint32_t>;
template class SparseMatrixBSR<double, uint32_t>;

}  // namespace cuspmm
