#include "formats/sparse_coo.hpp"

#include <vector>

namespace cuspmm {

template <typename DT, typename MT> SparseMatrixCOO<DT, MT>::SparseMatrixCOO(std::string filePath) {
    std::ifstream in(filePath);
    if (!in.is_open()) {
        std::cerr << "File " << filePath << " doesn't exist!" << std::endl;
        throw std::runtime_error("cannot open " + filePath);
    }
    in >> this->numRows >> this->numCols >> this->numNonZero;
    this->allocateSpace(false);
    for (size_t i = 0; i < this->numNonZero; ++i) in >> this->rowIdxs[i] >> this->colIdxs[i] >> this->data[i];
    if (in.fail()) throw std::runtime_error(filePath + ": truncated or malformed .coo file");
}

template <typename DT, typename MT>
SparseMatrixCOO<DT, MT>::SparseMatrixCOO(MT numRows, MT numCols, MT numNonZero, bool onDevice) {
    this->numRows = numRows;
    this->numCols = numCols;
    this->numNonZero = numNonZero;
    this->allocateSpace(onDevice);
}

template <typename DT, typename MT> SparseMatrixCOO<DT, MT>::~SparseMatrixCOO() {
    releaseBuffer(this->rowIdxs, this->onDevice);
    releaseBuffer(this->colIdxs, this->onDevice);
    releaseBuffer(this->data, this->onDevice);
    releaseBuffer(this->rowBoundsWorkspace, true);
    releaseBuffer(this->rowSpans, true);
}

template <typename DT, typename MT> bool SparseMatrixCOO<DT, MT>::allocateSpace(bool onDevice) {
    assert(this->data == nullptr && this->rowIdxs == nullptr && this->colIdxs == nullptr);
    this->rowIdxs = allocateBuffer<MT>(this->numNonZero, onDevice);
    this->colIdxs = allocateBuffer<MT>(this->numNonZero, onDevice);
    this->data = allocateBuffer<DT>(this->numNonZero, onDevice);
    if (onDevice) this->rowBoundsWorkspace = allocateBuffer<MT>((size_t)this->numRows + 1, true);
    this->onDevice = onDevice;
    return true;
}

template <typename DT, typename MT> SparseMatrixCOO<DT, MT> *SparseMatrixCOO<DT, MT>::copy2Device() {
    assert(!this->onDevice && this->data != nullptr);
    auto *d = new SparseMatrixCOO<DT, MT>(this->numRows, this->numCols, this->numNonZero, true);
    const MT *rows = this->rowIdxs, *cols = this->colIdxs;
    const DT *vals = this->data;
    // The HIP kernel walks rows, so entries must be grouped by row (the reference's atomicAdd kernel accepts any
    // order, spmm_coo_k1.cu:8-27).  A file in another order is put into STABLE row order here, once per upload:
    // each row keeps its storage order, so the sums still round exactly as spmmCOOCpu's do.
    std::vector<MT> sortedRows, sortedCols;
    std::vector<DT> sortedVals;
    if (!this->isRowSorted()) {
        const size_t nnz = this->numNonZero;
        std::vector<size_t> start((size_t)this->numRows + 1, 0);
        for (size_t i = 0; i < nnz; ++i) {
            if (this->rowIdxs[i] >= this->numRows) throw std::runtime_error("COO row index out of range");
            ++start[(size_t)this->rowIdxs[i] + 1];
        }
        for (size_t r = 0; r < this->numRows; ++r) start[r + 1] += start[r];
        sortedRows.resize(nnz);
        sortedCols.resize(nnz);
        sortedVals.resize(nnz);
        for (size_t i = 0; i < nnz; ++i) {
            const size_t o = start[this->rowIdxs[i]]++;
            sortedRows[o] = this->rowIdxs[i];
            sortedCols[o] = this->colIdxs[i];
            sortedVals[o] = this->data[i];
        }
        rows = sortedRows.data();
        cols = sortedCols.data();
        vals = sortedVals.data();
    }
    copyBuffer(d->rowIdxs, true, rows, false, (size_t)this->numNonZero * sizeof(MT));
    copyBuffer(d->colIdxs, true, cols, false, (size_t)this->numNonZero * sizeof(MT));
    copyBuffer(d->data, true, vals, false, (size_t)this->numNonZero * sizeof(DT));
    // long rows: the row boundaries of the sorted entries as spans, longest first (one counting pass per upload)
    if (this->numRows) {
        std::vector<uint32_t> rp((size_t)this->numRows + 1, 0);
        for (size_t i = 0; i < this->numNonZero; ++i) {
            if (rows[i] >= this->numRows) throw std::runtime_error("COO row index out of range");
            ++rp[(size_t)rows[i] + 1];
        }
        for (size_t r = 0; r < this->numRows; ++r) rp[r + 1] += rp[r];
        bool hybridOnly = false;
        if (wantsRowSpans(this->numRows, rp.data(), hybridOnly)) {
            uint32_t count = 0;
            uint32_t longCount = 0;
            d->rowSpans = uploadRowSpans(this->numRows, rp.data(), 0xFFFFFFFFu, count, &longCount);
            d->rowSpansLong = longCount;
            d->rowSpansHybridOnly = hybridOnly;
        }
    }
    return d;
}

template <typename DT, typename MT> DenseMatrix<DT, MT> *SparseMatrixCOO<DT, MT>::toDense() {
    assert(!this->onDevice);
    auto *dm = new DenseMatrix<DT, MT>(this->numRows, this->numCols, false);
    for (size_t i = 0; i < this->numNonZero; ++i)
        dm->data[RowMjIdx(this->rowIdxs[i], this->colIdxs[i], this->numCols)] = this->data[i];
    return dm;
}

template <typename DT, typename MT> bool SparseMatrixCOO<DT, MT>::isRowSorted() const {
    assert(!this->onDevice);
    for (size_t i = 1; i < this->numNonZero; ++i)
        if (this->rowIdxs[i] < this->rowIdxs[i - 1]) return false;
    return true;
}

template class SparseMatrixCOO<float, uint32_t>;
template class SparseMatrixCOO<double, uint32_t>;

}  // namespace cuspmm
