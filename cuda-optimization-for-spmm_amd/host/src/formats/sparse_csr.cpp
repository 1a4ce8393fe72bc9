#include <algorithm>
#include "formats/sparse_csr.hpp"
#include <type_traits>
#include <vector>

namespace cuspmm {

template <typename DT, typename MT> SparseMatrixCSR<DT, MT>::SparseMatrixCSR(std::string filePath) {
    std::ifstream in(filePath);
    if (!in.is_open()) {
        std::cerr << "File " << filePath << " doesn't exist!" << std::endl;
        throw std::runtime_error("cannot open " + filePath);
    }
    in >> this->numRows >> this->numCols >> this->numNonZero;
    this->allocateSpace(false);
    // three whitespace-separated arrays; the text format puts one per line but only the counts matter
    for (size_t i = 0; i <= this->numRows; ++i) in >> this->rowPtrs[i];
    for (size_t i = 0; i < this->numNonZero; ++i) in >> this->colIdxs[i];
    for (size_t i = 0; i < this->numNonZero; ++i) in >> this->data[i];
    if (in.fail()) throw std::runtime_error(filePath + ": truncated or malformed .csr file");
}

template <typename DT, typename MT>
SparseMatrixCSR<DT, MT>::SparseMatrixCSR(MT numRows, MT numCols, MT numNonZero, bool onDevice) {
    this->numRows = numRows;
    this->numCols = numCols;
    this->numNonZero = numNonZero;
    this->allocateSpace(onDevice);
}

template <typename DT, typename MT> SparseMatrixCSR<DT, MT>::~SparseMatrixCSR() {
    releaseBuffer(this->rowPtrs, this->onDevice);
    releaseBuffer(this->colIdxs, this->onDevice);
    releaseBuffer(this->data, this->onDevice);
    releaseBuffer(this->rowSpans, this->onDevice);
    releaseBuffer(this->planRowPtrs, this->onDevice);
    releaseBuffer(this->planColIdxs, this->onDevice);
    releaseBuffer(this->planData, this->onDevice);
    releaseBuffer(this->planRowMap, this->onDevice);
}

template <typename DT, typename MT> bool SparseMatrixCSR<DT, MT>::allocateSpace(bool onDevice) {
    assert(this->data == nullptr && this->rowPtrs == nullptr && this->colIdxs == nullptr);
    this->rowPtrs = allocateBuffer<MT>((size_t)this->numRows + 1, onDevice);
    this->colIdxs = allocateBuffer<MT>(this->numNonZero, onDevice);
    this->data = allocateBuffer<DT>(this->numNonZero, onDevice);
    this->onDevice = onDevice;
    return true;
}

bool wantsRowSpans(uint32_t numRows, const uint32_t *rowPtrsHost, bool &hybridOnly) {
    hybridOnly = false;
    if (numRows == 0 || rowPtrsHost[numRows] == 0) return false;
    if (rowPtrsHost[numRows] / numRows >= 24) return true;
    uint32_t longest = 0;
    for (uint32_t r = 0; r < numRows; ++r) longest = std::max(longest, rowPtrsHost[r + 1] - rowPtrsHost[r]);
    hybridOnly = true;
    return longest >= 64;
}

uint32_t *uploadRowSpans(uint32_t numRows, const uint32_t *rowPtrsHost, uint32_t shareLen, uint32_t &numSpans, uint32_t *numLongSpans) {
    uint32_t count = 0;
    mispmmCheckError(mispmm_csr_spans_by_length_host(numRows, rowPtrsHost, shareLen, &count, nullptr));
    uint32_t *spans = allocateBuffer<uint32_t>((size_t)count * 4, false);
    mispmmCheckError(mispmm_csr_spans_by_length_host(numRows, rowPtrsHost, shareLen, &count, spans));
    uint32_t *dev = allocateBuffer<uint32_t>((size_t)count * 4, true);
    copyBuffer(dev, true, spans, false, (size_t)count * 4 * sizeof(uint32_t));
    if (numLongSpans) mispmmCheckError(mispmm_csr_spans_long_count_host(count, spans, 32, numLongSpans));
    releaseBuffer(spans, false);
    numSpans = count;
    return dev;
}

template <typename DT, typename MT> SparseMatrixCSR<DT, MT> *SparseMatrixCSR<DT, MT>::copy2Device() {
    assert(!this->onDevice && this->rowPtrs != nullptr);
    auto *d = new SparseMatrixCSR<DT, MT>(this->numRows, this->numCols, this->numNonZero, true);
    copyBuffer(d->rowPtrs, true, this->rowPtrs, false, ((size_t)this->numRows + 1) * sizeof(MT));
    copyBuffer(d->colIdxs, true, this->colIdxs, false, (size_t)this->numNonZero * sizeof(MT));
    copyBuffer(d->data, true, this->data, false, (size_t)this->numNonZero * sizeof(DT));
    // structure check, once per upload: do all rows have the same length?
    const MT w = this->numRows ? this->rowPtrs[1] - this->rowPtrs[0] : 0;
    bool uniform = w > 0 && this->rowPtrs[0] == 0;
    for (size_t r = 0; uniform && r <= this->numRows; ++r) uniform = this->rowPtrs[r] == (MT)(r * w);
    d->uniformRowNnz = uniform ? w : 0;
    // long rows: the split kernel wants them longest first (one counting sort per upload)
    bool hybridOnly = false;
    if (wantsRowSpans(this->numRows, this->rowPtrs, hybridOnly)) {
        uint32_t count = 0;
        uint32_t longCount = 0;
        d->spansHybridOnly = hybridOnly;
        d->rowSpans = uploadRowSpans(this->numRows, this->rowPtrs, 0, count, &longCount);
        d->numSpans = count;
        d->numLongSpans = longCount;
    } else if constexpr (std::is_same_v<DT, float>) {
        // short rows: keep a clustered row order when the greedy walk finds one (8 clusters: the best of 4 / 8 / 16 / 64 in the
        // K = 512 A/B, profiles/r3/plan_order.log); the wrapper multiplies from it where the product is bound by what the L2s fetch
        if (this->numRows >= 1024 && this->numNonZero > 0) {
            std::vector<uint32_t> order(this->numRows);
            uint64_t natural = 0, clustered = 0;
            mispmmCheckError(mispmm_csr_cluster_rows_host(this->numRows, this->numCols, this->rowPtrs, this->colIdxs, 8, order.data(),
                                                          &natural, &clustered));
            if (clustered * 10 <= natural * 9) {
                std::vector<uint32_t> ptrs((size_t)this->numRows + 1), cols(this->numNonZero);
                std::vector<float> vals(this->numNonZero);
                mispmmCheckError(mispmm_csr_permute_rows_host(this->numRows, this->rowPtrs, this->colIdxs, this->data, order.data(),
                                                              ptrs.data(), cols.data(), vals.data()));
                d->planRowPtrs = allocateBuffer<MT>((size_t)this->numRows + 1, true);
                d->planColIdxs = allocateBuffer<MT>(this->numNonZero, true);
                d->planData = allocateBuffer<DT>(this->numNonZero, true);
                d->planRowMap = allocateBuffer<MT>(this->numRows, true);
                copyBuffer(d->planRowPtrs, true, ptrs.data(), false, ptrs.size() * sizeof(MT));
                copyBuffer(d->planColIdxs, true, cols.data(), false, cols.size() * sizeof(MT));
                copyBuffer(d->planData, true, vals.data(), false, vals.size() * sizeof(DT));
                copyBuffer(d->planRowMap, true, order.data(), false, order.size() * sizeof(MT));
            }
        }
    }
    return d;
}

// Plan order or storage order for a dense operand of N columns?  Measured once per width on scratch operands
// (mispmm_csr_autotune_plan_f32) and remembered; where nothing can be measured (no plan, not a float matrix, the call fails)
// the footprint rule of round 3: plan where the slice of B one XCD reads does not fit its 4 MiB L2.
template <typename DT, typename MT> bool SparseMatrixCSR<DT, MT>::usePlanFor(uint32_t N, int accMode) {
    if (!this->planRowMap || !this->onDevice) return false;
    const bool rule = N % 512 == 0 && (uint64_t)this->numCols * (N / 8) * 4 > (4ull << 20);
    if constexpr (!std::is_same_v<DT, float>) {
        return rule;
    } else {
        const uint64_t key = (uint64_t)N * 2 + (accMode == MISPMM_ACC_FAST ? 1 : 0);
        auto it = planChoice.find(key);
        if (it != planChoice.end()) return it->second;
        int use = 0;
        float times[2];
        const int st = mispmm_csr_autotune_plan_f32(nullptr, this->numRows, this->numCols, this->numNonZero, this->rowPtrs, this->colIdxs, this->data,
                                                    this->uniformRowNnz, this->planRowPtrs, this->planColIdxs, this->planData, this->planRowMap, N,
                                                    accMode, 0, &use, times);
        if (st != MISPMM_OK) return rule;
        planChoice[key] = use != 0;
        return use != 0;
    }
}

template <typename DT, typename MT> DenseMatrix<DT, MT> *SparseMatrixCSR<DT, MT>::toDense() {
    assert(!this->onDevice);
    auto *dm = new DenseMatrix<DT, MT>(this->numRows, this->numCols, false);
    for (MT r = 0; r < this->numRows; ++r)
        for (MT i = this->rowPtrs[r]; i < this->rowPtrs[r + 1]; ++i)
            dm->data[RowMjIdx(r, this->colIdxs[i], this->numCols)] = this->data[i];
    return dm;
}

template <typename DT, typename MT> std::ostream &operator<<(std::ostream &out, SparseMatrixCSR<DT, MT> &m) {
    out << m.numRows << ' ' << m.numCols << ' ' << m.numNonZero << '\n';
    for (size_t i = 0; i <= m.numRows; ++i) out << m.rowPtrs[i] << (i == m.numRows ? '\n' : ' ');
    for (size_t i = 0; i < m.numNonZero; ++i) out << m.colIdxs[i] << ' ';
    out << '\n';
    for (size_t i = 0; i < m.numNonZero; ++i) out << m.data[i] << ' ';
    out << '\n';
    return out;
}

template class SparseMatrixCSR<float, uint32_t>;
template class SparseMatrixCSR<double, uint32_t>;
template std::ostream &operator<<(std::ostream &, SparseMatrixCSR<float, uint32_t> &);
template std::ostream &operator<<(std::ostream &, SparseMatrixCSR<double, uint32_t> &);

}  // namespace cuspmm
