#include "formats/sparse_ell.hpp"

#include <vector>

namespace cuspmm {

template <typename DT, typename MT>
SparseMatrixELL<DT, MT>::SparseMatrixELL(std::string rowindPath, std::string valuesPath) {
    std::ifstream idx(rowindPath), val(valuesPath);
    if (!idx.is_open()) {
        std::cerr << "File " << rowindPath << " doesn't exist!" << std::endl;
        throw std::runtime_error("cannot open " + rowindPath);
    }
    if (!val.is_open()) {
        std::cerr << "File " << valuesPath << " doesn't exist!" << std::endl;
        throw std::runtime_error("cannot open " + valuesPath);
    }
    idx >> this->numRows >> this->numCols >> this->numNonZero >> this->maxColNnz;
    this->allocateSpace(false);
    const size_t slots = this->numSlots();
    for (size_t i = 0; i < slots; ++i) {
        long long v;  // "-1" marks padding; stored as 0xFFFFFFFF like `istream >> uint32_t` does
        idx >> v;
        this->rowIdxs[i] = (MT)v;
    }
    for (size_t i = 0; i < slots; ++i) val >> this->data[i];
    if (idx.fail() || val.fail()) throw std::runtime_error(rowindPath + ": truncated or malformed ELL files");
}

template <typename DT, typename MT>
SparseMatrixELL<DT, MT>::SparseMatrixELL(MT numRows, MT numCols, MT numNonZero, MT maxColNnz, bool onDevice) {
    this->numRows = numRows;
    this->numCols = numCols;
    this->numNonZero = numNonZero;
    this->maxColNnz = maxColNnz;
    this->allocateSpace(onDevice);
}

template <typename DT, typename MT> SparseMatrixELL<DT, MT>::~SparseMatrixELL() {
    releaseBuffer(this->rowIdxs, this->onDevice);
    releaseBuffer(this->data, this->onDevice);
    releaseBuffer(this->rmColIdxs, true);
    releaseBuffer(this->rmData, true);
    releaseBuffer(this->cpRowPtrs, true);
    releaseBuffer(this->cpColIdxs, true);
    releaseBuffer(this->cpData, true);
    releaseBuffer(this->cpSpans, true);
}

template <typename DT, typename MT> bool SparseMatrixELL<DT, MT>::allocateSpace(bool onDevice) {
    assert(this->data == nullptr && this->rowIdxs == nullptr);
    this->rowIdxs = allocateBuffer<MT>(this->numSlots(), onDevice);
    this->data = allocateBuffer<DT>(this->numSlots(), onDevice);
    this->onDevice = onDevice;
    return true;
}

template <typename DT, typename MT> SparseMatrixELL<DT, MT> *SparseMatrixELL<DT, MT>::copy2Device() {
    assert(!this->onDevice && this->data != nullptr);
    auto *d = new SparseMatrixELL<DT, MT>(this->numRows, this->numCols, this->numNonZero, this->maxColNnz, true);
    copyBuffer(d->rowIdxs, true, this->rowIdxs, false, this->numSlots() * sizeof(MT));
    copyBuffer(d->data, true, this->data, false, this->numSlots() * sizeof(DT));
    // Row-major view for the row-parallel kernel, built once here (layout conversion sits outside
    // the timed sections, like DenseMatrix::toOrdering in the reference's wrappers).
    if constexpr (std::is_same_v<DT, float>) {
        uint32_t width = 0;
        mispmmCheckError(mispmm_ell_colmajor_to_rowmajor_host(this->numRows, this->numCols, this->maxColNnz,
                                                              this->rowIdxs, this->data, &width, nullptr, nullptr));
        const size_t n = (size_t)this->numRows * width;
        std::vector<uint32_t> cols(n ? n : 1);
        std::vector<float> vals(n ? n : 1);
        mispmmCheckError(mispmm_ell_colmajor_to_rowmajor_host(this->numRows, this->numCols, this->maxColNnz,
                                                              this->rowIdxs, this->data, &width, cols.data(),
                                                              vals.data()));
        d->rowWidth = width;
        d->rmColIdxs = allocateBuffer<MT>(n, true);
        d->rmData = allocateBuffer<DT>(n, true);
        copyBuffer(d->rmColIdxs, true, cols.data(), false, n * sizeof(MT));
        copyBuffer(d->rmData, true, vals.data(), false, n * sizeof(DT));
        // mostly padding (an ELL is as wide as its longest row): also keep the list of occupied slots
        uint32_t occupied = 0;
        mispmmCheckError(mispmm_ell_compact_host(this->numRows, width, cols.data(), vals.data(), &occupied, nullptr, nullptr, nullptr));
        if (n != 0 && (size_t)occupied * 2 < n) {
            std::vector<uint32_t> rp((size_t)this->numRows + 1), ci(occupied ? occupied : 1);
            std::vector<float> va(occupied ? occupied : 1);
            mispmmCheckError(mispmm_ell_compact_host(this->numRows, width, cols.data(), vals.data(), &occupied, rp.data(), ci.data(),
                                                     va.data()));
            d->cpRowPtrs = allocateBuffer<MT>((size_t)this->numRows + 1, true);
            d->cpColIdxs = allocateBuffer<MT>(ci.size(), true);
            d->cpData = allocateBuffer<DT>(va.size(), true);
            copyBuffer(d->cpRowPtrs, true, rp.data(), false, rp.size() * sizeof(MT));
            copyBuffer(d->cpColIdxs, true, ci.data(), false, ci.size() * sizeof(MT));
            copyBuffer(d->cpData, true, va.data(), false, va.size() * sizeof(DT));
            d->cpCount = occupied;
            bool hybridOnly = false;
            if (wantsRowSpans(this->numRows, rp.data(), hybridOnly)) {
                uint32_t count = 0;
                uint32_t longCount = 0;
                d->cpSpans = uploadRowSpans(this->numRows, rp.data(), 0xFFFFFFFFu, count, &longCount);
                d->cpSpansLong = longCount;
                d->cpSpansHybridOnly = hybridOnly;
            }
        }
    }
    return d;
}

template <typename DT, typename MT> DenseMatrix<DT, MT> *SparseMatrixELL<DT, MT>::toDense() {
    assert(!this->onDevice);
    auto *dm = new DenseMatrix<DT, MT>(this->numRows, this->numCols, false);
    for (MT c = 0; c < this->numCols; ++c)
        for (MT s = 0; s < this->maxColNnz; ++s) {
            const size_t i = (size_t)c * this->maxColNnz + s;
            if ((int32_t)this->rowIdxs[i] >= 0) dm->data[RowMjIdx(this->rowIdxs[i], c, this->numCols)] = this->data[i];
        }
    return dm;
}

template class SparseMatrixELL<float, uint32_t>;
template class SparseMatrixELL<double, uint32_t>;

}  // namespace cuspmm
