#include "formats/dense.hpp"

#include <iomanip>
#include <limits>
#include <vector>

namespace cuspmm {

static std::ifstream openOrThrow(const std::string &path) {
    std::ifstream f(path);
    if (!f.is_open()) {
        std::cerr << "File " << path << " doesn't exist!" << std::endl;
        throw std::runtime_error("cannot open " + path);
    }
    return f;
}

template <typename DT, typename MT> DenseMatrix<DT, MT>::DenseMatrix(std::string filePath) {
    std::ifstream in = openOrThrow(filePath);
    std::string line;
    in >> this->numRows >> this->numCols;
    std::getline(in, line);  // rest of the header (the converter appends a non-zero count)
    this->allocateSpace(false);
    for (MT r = 0; r < this->numRows; ++r) {
        if (!std::getline(in, line)) throw std::runtime_error(filePath + ": fewer rows than the header says");
        std::istringstream row(line);
        for (MT c = 0; c < this->numCols; ++c) row >> this->data[RowMjIdx(r, c, this->numCols)];
    }
}

template <typename DT, typename MT>
DenseMatrix<DT, MT>::DenseMatrix(MT numRows, MT numCols, bool onDevice, ORDERING ordering) {
    this->numRows = numRows;
    this->numCols = numCols;
    this->ordering = ordering;
    this->allocateSpace(onDevice);
}

template <typename DT, typename MT> DenseMatrix<DT, MT>::DenseMatrix(DenseMatrix<DT, MT> *source, bool onDevice) {
    this->numRows = source->numRows;
    this->numCols = source->numCols;
    this->ordering = source->ordering;
    this->allocateSpace(onDevice);
    this->copyData(source);
}

template <typename DT, typename MT> DenseMatrix<DT, MT>::~DenseMatrix() { this->freeSpace(); }

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <typename DT, typename MT>
DenseMatrix<DT, MT> *DenseMatrix<DT, MT>::synthetic(MT numRows, MT numCols, uint64_t seed, int mode) {
    auto *m = new DenseMatrix<DT, MT>(numRows, numCols, false, ORDERING::ROW_MAJOR);
    const uint64_t base = seed << 40;
    const size_t n = m->numElements();
    for (size_t i = 0; i < n; ++i) {
        const uint64_t h = splitmix64(base + i);
        if (mode == 0) m->data[i] = (DT)((float)(h >> 40) * 0x1p-23f - 1.0f);
        else m->data[i] = (DT)((float)((int)(h >> 55) - 256) * 0x1p-8f);
    }
    return m;
}

template <typename DT, typename MT> bool DenseMatrix<DT, MT>::copyData(DenseMatrix<DT, MT> *source) {
    this->assertSameShape(source);
    copyBuffer(this->data, this->onDevice, source->data, source->onDevice, this->numElements() * sizeof(DT));
    this->ordering = source->ordering;
    return true;
}

template <typename DT, typename MT> void DenseMatrix<DT, MT>::assertSameShape(DenseMatrix<DT, MT> *target) {
    assert(this->numRows == target->numRows && this->numCols == target->numCols);
    (void)target;
}

template <typename DT, typename MT> DenseMatrix<DT, MT> *DenseMatrix<DT, MT>::copy2Device() {
    assert(!this->onDevice && this->data != nullptr);
    return new DenseMatrix<DT, MT>(this, true);
}

template <typename DT, typename MT> DenseMatrix<DT, MT> *DenseMatrix<DT, MT>::copy2Host() {
    assert(this->onDevice && this->data != nullptr);
    return new DenseMatrix<DT, MT>(this, false);
}

template <typename DT, typename MT> bool DenseMatrix<DT, MT>::toOrdering(ORDERING newOrdering) {
    if (this->ordering == newOrdering) return true;
    if (newOrdering != ORDERING::ROW_MAJOR && newOrdering != ORDERING::COL_MAJOR)
        throw std::runtime_error("Incorrect ordering value");
    // Row-major [R x C] -> column-major is the transpose written as a row-major [C x R] buffer, and back.
    const MT srcRows = this->ordering == ORDERING::ROW_MAJOR ? this->numRows : this->numCols;
    const MT srcCols = this->ordering == ORDERING::ROW_MAJOR ? this->numCols : this->numRows;
    DT *fresh = allocateBuffer<DT>(this->numElements(), this->onDevice);
    if (this->onDevice) {
        if constexpr (std::is_same_v<DT, float>) {
            mispmmCheckError(mispmm_dense_transpose_f32(nullptr, srcRows, srcCols, this->data, fresh));
            mispmmCheckError(mispmm_device_sync());
        } else {
            // no device kernel for this element type: transpose through pinned host memory
            DT *h = allocateBuffer<DT>(this->numElements(), false);
            DT *t = allocateBuffer<DT>(this->numElements(), false);
            copyBuffer(h, false, this->data, true, this->numElements() * sizeof(DT));
            for (MT r = 0; r < srcRows; ++r)
                for (MT c = 0; c < srcCols; ++c) t[(size_t)c * srcRows + r] = h[(size_t)r * srcCols + c];
            copyBuffer(fresh, true, t, false, this->numElements() * sizeof(DT));
            releaseBuffer(h, false);
            releaseBuffer(t, false);
        }
    } else {
        for (MT r = 0; r < srcRows; ++r)
            for (MT c = 0; c < srcCols; ++c) fresh[(size_t)c * srcRows + r] = this->data[(size_t)r * srcCols + c];
    }
    releaseBuffer(this->data, this->onDevice);
    this->data = fresh;
    this->ordering = newOrdering;
    return true;
}

template <typename DT, typename MT> bool DenseMatrix<DT, MT>::save2File(std::string filePath) {
    std::vector<DT> staged;
    const DT *src = this->data;
    if (this->onDevice) {
        staged.resize(this->numElements());
        copyBuffer(staged.data(), false, this->data, true, this->numElements() * sizeof(DT));
        src = staged.data();
    }
    std::ofstream out(filePath);
    if (!out.is_open()) {
        std::cerr << "Cannot open output file " << filePath << std::endl;
        return false;
    }
    out << std::setprecision(std::numeric_limits<DT>::max_digits10);
    if (this->ordering == ORDERING::ROW_MAJOR) {
        out << this->numRows << ' ' << this->numCols << '\n';
        for (MT r = 0; r < this->numRows; ++r) {
            for (MT c = 0; c < this->numCols; ++c) out << src[RowMjIdx(r, c, this->numCols)] << ' ';
            out << '\n';
        }
    } else {
        out << this->numRows << ' ' << this->numCols << " COL_MAJOR\n";
        for (MT c = 0; c < this->numCols; ++c) {
            for (MT r = 0; r < this->numRows; ++r) out << src[ColMjIdx(r, c, this->numRows)] << ' ';
            out << '\n';
        }
    }
    return true;
}

template <typename DT, typename MT> bool DenseMatrix<DT, MT>::allocateSpace(bool onDevice) {
    assert(this->data == nullptr);
    this->data = allocateBuffer<DT>(this->numElements(), onDevice);  // zero-filled on either side
    this->onDevice = onDevice;
    return true;
}

template <typename DT, typename MT> bool DenseMatrix<DT, MT>::freeSpace() {
    releaseBuffer(this->data, this->onDevice);
    this->data = nullptr;
    return true;
}

template class DenseMatrix<float, uint32_t>;
template class DenseMatrix<double, uint32_t>;

}  // namespace cuspmm
