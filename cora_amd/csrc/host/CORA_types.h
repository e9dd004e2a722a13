// Basic types of the C++ host (mirrors the reference's include/CORA/CORA_types.h
// without Eigen: the image has no Eigen, and the host only needs column-major
// dense storage and a CSR container to talk to the C ABI in include/cora_hip.h).
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <iostream>  // the reference's headers bring it in (examples/main.cpp uses std::cout without including it)
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace CORA {

typedef double Scalar;
typedef std::ptrdiff_t Index;

class NotImplementedException : public std::logic_error {
 public:
  explicit NotImplementedException(std::string const &str) : std::logic_error(str + " not implemented") {}
};

// include/CORA/CORA_types.h:23-39
class MatrixShapeException : public std::logic_error {
 public:
  MatrixShapeException(const std::string &func_name, Index exp_rows, Index exp_cols, Index act_rows,
                       Index act_cols)
      : std::logic_error(func_name + ": " + "expected matrix of shape (" + std::to_string(exp_rows) +
                         ", " + std::to_string(exp_cols) + ") but got (" + std::to_string(act_rows) +
                         ", " + std::to_string(act_cols) + ")") {}
};
inline void checkMatrixShape(const std::string &func_name, Index exp_rows, Index exp_cols, Index act_rows,
                             Index act_cols) {
  if (exp_rows != act_rows || exp_cols != act_cols)
    throw MatrixShapeException(func_name, exp_rows, exp_cols, act_rows, act_cols);
}

class SparseMatrix;

/** Column-major dense matrix of doubles (the storage order of Eigen::MatrixXd,
 * so data() can be handed to the C ABI with ld = rows()). */
class Matrix {
  Index rows_ = 0, cols_ = 0;
  std::vector<Scalar> a_;

 public:
  Matrix() = default;
  Matrix(Index r, Index c) : rows_(r), cols_(c), a_(static_cast<size_t>(r * c), 0.0) {}
  /** `CORA::Vector b(n)` (Eigen's VectorXd(n)): n x 1, zero-filled here (Eigen leaves it uninitialised). */
  explicit Matrix(Index n) : Matrix(n, 1) {}
  /** Dense copy of a sparse matrix, implicit as in Eigen (the reference's tests/test.cpp:206 hands a SparseMatrix to
   * blockCholeskySolve's `const Matrix &rhs`). */
  Matrix(const SparseMatrix &S);  // NOLINT(runtime/explicit)
  /** Eigen's comma initialiser, `v << 1.0, 2.0, 3.0;` -- row by row (tests/test.cpp:70,212). */
  class CommaInit {
    Matrix &m_;
    Index k_ = 0;

   public:
    CommaInit(Matrix &m, Scalar first) : m_(m) { put(first); }
    CommaInit &operator,(Scalar v) {
      put(v);
      return *this;
    }

   private:
    void put(Scalar v) {
      if (k_ >= m_.size()) throw std::out_of_range("Matrix <<: more coefficients than entries");
      m_(k_ / m_.cols(), k_ % m_.cols()) = v;
      ++k_;
    }
  };
  CommaInit operator<<(Scalar first) { return CommaInit(*this, first); }
  /** Inverse by Gaussian elimination with partial pivoting (Eigen's MatrixBase::inverse(); small dense matrices of the
   * tests only: tests/test.cpp:56,112,191). */
  Matrix inverse() const;
  static Matrix Zero(Index r, Index c) { return Matrix(r, c); }
  static Matrix Identity(Index r, Index c) {
    Matrix m(r, c);
    for (Index i = 0; i < std::min(r, c); ++i) m(i, i) = 1.0;
    return m;
  }
  /** Uniform in [-1, 1] like Eigen's Matrix::Random (src/CORA_problem.cpp:1026).  Without a seed successive calls return
   * DIFFERENT matrices, as Eigen's do (it draws from rand()): the reference's tests/test_geometry.cpp:53-68 projects one
   * random matrix onto the tangent space at the normalisation of another and requires a non-zero result.  The sequence is
   * the same in every process (a counter, not a clock); everything inside the library passes an explicit seed. */
  static Matrix Random(Index r, Index c) {
    static std::atomic<uint64_t> calls{0};
    return Random(r, c, 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull * calls.fetch_add(1));
  }
  static Matrix Random(Index r, Index c, uint64_t seed) {
    Matrix m(r, c);
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<Scalar> u(-1.0, 1.0);
    for (auto &v : m.a_) v = u(g);
    return m;
  }
  /** The same distribution with a generator per column (seeded seed + column): column j holds the same numbers whatever
   * the number of columns asked for, and the columns are filled by threads (10^6 poses: 54 M numbers, 0.3 s on one). */
  static Matrix RandomColumns(Index r, Index c, uint64_t seed);
  // The vector forms the reference's tests use on Eigen's VectorXd (tests/test_certification.cpp:21,52): n x 1.
  static Matrix Zero(Index n) { return Matrix(n, 1); }
  static Matrix Random(Index n) { return Random(n, 1); }
  Matrix normalized() const {
    const Scalar nrm = norm();
    return nrm > 0 ? (*this) * (1.0 / nrm) : *this;
  }
  /** Sparse copy of the non-zero entries (Eigen's MatrixBase::sparseView()). */
  SparseMatrix sparseView() const;
  /** A 1 x 1 product read as a number, as Eigen allows (`Scalar theta = x.transpose() * S * x`). */
  operator Scalar() const {
    if (rows_ != 1 || cols_ != 1) throw std::logic_error("Matrix: only a 1 x 1 matrix converts to a scalar");
    return a_[0];
  }
  Index rows() const { return rows_; }
  Index cols() const { return cols_; }
  Index size() const { return rows_ * cols_; }
  Scalar *data() { return a_.data(); }
  const Scalar *data() const { return a_.data(); }
  Scalar &operator()(Index i, Index j) { return a_[static_cast<size_t>(j * rows_ + i)]; }
  Scalar operator()(Index i, Index j) const { return a_[static_cast<size_t>(j * rows_ + i)]; }
  Scalar &operator()(Index i) { return a_[static_cast<size_t>(i)]; }
  Scalar operator()(Index i) const { return a_[static_cast<size_t>(i)]; }
  void setZero() { std::fill(a_.begin(), a_.end(), 0.0); }

  /** A writable view of a block (what Eigen's non-const block() / row() / col() give): `M.block(i, j, p, q) = B`,
   * `M.row(i) = v` (a vector is accepted as a row, as Eigen does), and it reads as the Matrix it covers.  The reference's
   * tests/test_construct_problem.cpp:53-60,118-120 builds its expected states this way. */
  class BlockRef {
    Matrix &m_;
    Index r0_, c0_, nr_, nc_;

   public:
    BlockRef(Matrix &m, Index r0, Index c0, Index nr, Index nc) : m_(m), r0_(r0), c0_(c0), nr_(nr), nc_(nc) {}
    operator Matrix() const { return static_cast<const Matrix &>(m_).block(r0_, c0_, nr_, nc_); }
    BlockRef &operator=(const Matrix &b) {
      if (b.rows() == nr_ && b.cols() == nc_) {
        m_.setBlock(r0_, c0_, b);
      } else if (b.rows() == nc_ && b.cols() == nr_ && (nr_ == 1 || nc_ == 1)) {
        m_.setBlock(r0_, c0_, b.transpose());
      } else {
        throw MatrixShapeException("Matrix::block = ", nr_, nc_, b.rows(), b.cols());
      }
      return *this;
    }
    BlockRef &operator=(const BlockRef &o) { return *this = static_cast<Matrix>(o); }
    Matrix operator+(const Matrix &o) const { return static_cast<Matrix>(*this) + o; }
    Matrix operator-(const Matrix &o) const { return static_cast<Matrix>(*this) - o; }
    Matrix operator*(const Matrix &o) const { return static_cast<Matrix>(*this) * o; }
    Matrix transpose() const { return static_cast<Matrix>(*this).transpose(); }
    Scalar norm() const { return static_cast<Matrix>(*this).norm(); }
    Index rows() const { return nr_; }
    Index cols() const { return nc_; }
    Scalar operator()(Index i, Index j) const { return static_cast<const Matrix &>(m_)(r0_ + i, c0_ + j); }
  };
  BlockRef block(Index r0, Index c0, Index nr, Index nc) { return BlockRef(*this, r0, c0, nr, nc); }
  BlockRef row(Index i) { return BlockRef(*this, i, 0, 1, cols_); }
  BlockRef col(Index j) { return BlockRef(*this, 0, j, rows_, 1); }
  Matrix row(Index i) const { return block(i, 0, 1, cols_); }
  Matrix operator-() const { return (*this) * -1.0; }
  void normalize() { *this = normalized(); }
  Matrix block(Index r0, Index c0, Index nr, Index nc) const {
    Matrix b(nr, nc);
    for (Index j = 0; j < nc; ++j)
      for (Index i = 0; i < nr; ++i) b(i, j) = (*this)(r0 + i, c0 + j);
    return b;
  }
  void setBlock(Index r0, Index c0, const Matrix &b) {
    for (Index j = 0; j < b.cols(); ++j)
      for (Index i = 0; i < b.rows(); ++i) (*this)(r0 + i, c0 + j) = b(i, j);
  }
  Matrix col(Index j) const { return block(0, j, rows_, 1); }
  Matrix transpose() const {
    Matrix t(cols_, rows_);
    for (Index j = 0; j < cols_; ++j)
      for (Index i = 0; i < rows_; ++i) t(j, i) = (*this)(i, j);
    return t;
  }
  Matrix operator*(const Matrix &o) const {
    if (cols_ != o.rows_) throw std::invalid_argument("Matrix product: inner dimensions differ");
    Matrix r(rows_, o.cols_);
    for (Index j = 0; j < o.cols_; ++j)
      for (Index k = 0; k < cols_; ++k) {
        const Scalar b = o(k, j);
        if (b == 0.0) continue;
        for (Index i = 0; i < rows_; ++i) r(i, j) += (*this)(i, k) * b;
      }
    return r;
  }
  Matrix operator+(const Matrix &o) const {
    checkMatrixShape("Matrix::operator+", rows_, cols_, o.rows_, o.cols_);
    Matrix r = *this;
    for (size_t i = 0; i < a_.size(); ++i) r.a_[i] += o.a_[i];
    return r;
  }
  Matrix operator-(const Matrix &o) const {
    checkMatrixShape("Matrix::operator-", rows_, cols_, o.rows_, o.cols_);
    Matrix r = *this;
    for (size_t i = 0; i < a_.size(); ++i) r.a_[i] -= o.a_[i];
    return r;
  }
  // (any arithmetic type, matched exactly: with the 1 x 1 conversion below `2 * M` would otherwise also read as int * double)
  template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>
  Matrix operator*(T s) const {
    Matrix r = *this;
    for (auto &v : r.a_) v *= static_cast<Scalar>(s);
    return r;
  }
  Scalar dot(const Matrix &o) const {
    checkMatrixShape("Matrix::dot", rows_, cols_, o.rows_, o.cols_);
    Scalar s = 0;
    for (size_t i = 0; i < a_.size(); ++i) s += a_[i] * o.a_[i];
    return s;
  }
  Scalar norm() const { return std::sqrt(dot(*this)); }
  Scalar trace() const {
    Scalar s = 0;
    for (Index i = 0; i < std::min(rows_, cols_); ++i) s += (*this)(i, i);
    return s;
  }
  /** Eigen's isZero / isApprox (default precision 1e-12): |a_ij| <= prec for every entry; |a - b|_F <= prec min(|a|_F, |b|_F). */
  bool isZero(Scalar prec = 1e-12) const {
    for (Scalar v : a_)
      if (std::fabs(v) > prec) return false;
    return true;
  }
  bool isApprox(const Matrix &o, Scalar prec = 1e-12) const {
    if (rows_ != o.rows_ || cols_ != o.cols_) return false;
    return (*this - o).norm() <= prec * std::min(norm(), o.norm());
  }
  bool hasNaN() const {
    for (Scalar v : a_)
      if (v != v) return true;
    return false;
  }
};
template <typename T, typename = typename std::enable_if<std::is_arithmetic<T>::value>::type>
inline Matrix operator*(T s, const Matrix &m) { return m * s; }
typedef Matrix Vector;  // N x 1

inline Matrix Matrix::inverse() const {
  if (rows_ != cols_) throw std::invalid_argument("Matrix::inverse: not square");
  const Index n = rows_;
  Matrix A = *this, B = Identity(n, n);
  for (Index k = 0; k < n; ++k) {
    Index piv = k;
    for (Index i = k + 1; i < n; ++i)
      if (std::fabs(A(i, k)) > std::fabs(A(piv, k))) piv = i;
    if (A(piv, k) == 0.0) throw std::runtime_error("Matrix::inverse: singular");
    if (piv != k)
      for (Index j = 0; j < n; ++j) {
        std::swap(A(k, j), A(piv, j));
        std::swap(B(k, j), B(piv, j));
      }
    const Scalar inv = 1.0 / A(k, k);
    for (Index i = 0; i < n; ++i) {
      if (i == k) continue;
      const Scalar f = A(i, k) * inv;
      if (f == 0.0) continue;
      for (Index j = k; j < n; ++j) A(i, j) -= f * A(k, j);
      for (Index j = 0; j < n; ++j) B(i, j) -= f * B(k, j);
    }
    for (Index j = k; j < n; ++j) A(k, j) *= inv;
    for (Index j = 0; j < n; ++j) B(k, j) *= inv;
  }
  return B;
}

/** Eigen::VectorXi as the reference uses it (include/CORA/CORA_types.h:44; block sizes of getBlockCholeskyFactorization,
 * tests/test.cpp:50-51,166-167): a vector of ints with Eigen's size constructor, comma initialiser and operator(). */
class VectorXi : public std::vector<int> {
 public:
  VectorXi() = default;
  explicit VectorXi(Index n) : std::vector<int>(static_cast<size_t>(n), 0) {}
  VectorXi(const std::vector<int> &v) : std::vector<int>(v) {}  // NOLINT(runtime/explicit)
  VectorXi(std::initializer_list<int> v) : std::vector<int>(v) {}
  class CommaInit {
    VectorXi &v_;
    size_t k_ = 0;

   public:
    CommaInit(VectorXi &v, int first) : v_(v) { put(first); }
    CommaInit &operator,(int x) {
      put(x);
      return *this;
    }

   private:
    void put(int x) {
      if (k_ >= v_.std::vector<int>::size()) throw std::out_of_range("VectorXi <<: more coefficients than entries");
      v_[k_++] = x;
    }
  };
  CommaInit operator<<(int first) { return CommaInit(*this, first); }
  Index size() const { return static_cast<Index>(std::vector<int>::size()); }
  int &operator()(Index i) { return (*this)[static_cast<size_t>(i)]; }
  int operator()(Index i) const { return (*this)[static_cast<size_t>(i)]; }
  int sum() const {
    int s = 0;
    for (int x : *this) s += x;
    return s;
  }
};

/** Row-major CSR with int32 indices: the layout of the reference's
 * Eigen::SparseMatrix<Scalar, Eigen::RowMajor> (include/CORA/CORA_types.h:70). */
struct Triplet {
  Index r, c;
  Scalar v;
};
class SparseMatrix {
  Index rows_ = 0, cols_ = 0;

 public:
  std::vector<int32_t> outer;  // rows + 1
  std::vector<int32_t> inner;
  std::vector<Scalar> values;

  SparseMatrix() : outer(1, 0) {}
  SparseMatrix(Index r, Index c) : rows_(r), cols_(c), outer(static_cast<size_t>(r) + 1, 0) {}
  Index rows() const { return rows_; }
  Index cols() const { return cols_; }
  Index nonZeros() const { return static_cast<Index>(inner.size()); }
  const int32_t *outerIndexPtr() const { return outer.data(); }
  const int32_t *innerIndexPtr() const { return inner.data(); }
  const Scalar *valuePtr() const { return values.data(); }

  /** Eigen's element-wise interface (the reference's tests/test.cpp:29-47,99-107,153-187 builds its matrices with it).  The
   * matrix is always kept compressed, so insert() moves the entries behind it -- meant for the small matrices of tests --
   * and makeCompressed() has nothing to do.  The reference returned by insert() / coeffRef() is valid until the next one. */
  Scalar &insert(Index i, Index j) {
    int32_t at;
    if (find(i, j, &at)) throw std::invalid_argument("SparseMatrix::insert: the entry exists (use coeffRef)");
    inner.insert(inner.begin() + at, static_cast<int32_t>(j));
    values.insert(values.begin() + at, 0.0);
    for (Index r = i + 1; r <= rows_; ++r) outer[static_cast<size_t>(r)]++;
    return values[static_cast<size_t>(at)];
  }
  Scalar &coeffRef(Index i, Index j) {
    int32_t at;
    return find(i, j, &at) ? values[static_cast<size_t>(at)] : insert(i, j);
  }
  Scalar coeff(Index i, Index j) const {
    int32_t at;
    return find(i, j, &at) ? values[static_cast<size_t>(at)] : 0.0;
  }
  void makeCompressed() {}
  bool isCompressed() const { return true; }
  SparseMatrix operator*(const SparseMatrix &o) const { return times(nullptr, o); }
  /** Eigen's SparseMatrixBase::isApprox: |a - b|_F <= prec min(|a|_F, |b|_F). */
  bool isApprox(const SparseMatrix &o, Scalar prec = 1e-12) const;

  /** Duplicates are summed (in the order they were given) and structural zeros kept, like Eigen's setFromTriplets.
   * Rows are formed by a counting sort, columns sorted inside each row: linear in the number of triplets (the
   * comparison sort of 4.5 M 24-byte triplets was most of updateProblemData at 10^5 poses). */
  void setFromTriplets(std::vector<Triplet> t) {
    outer.assign(static_cast<size_t>(rows_) + 1, 0);
    inner.clear();
    values.clear();
    for (const Triplet &e : t) {
      if (e.r < 0 || e.r >= rows_ || e.c < 0 || e.c >= cols_)
        throw std::invalid_argument("SparseMatrix::setFromTriplets: index out of range");
      outer[static_cast<size_t>(e.r) + 1]++;
    }
    for (Index i = 0; i < rows_; ++i) outer[static_cast<size_t>(i) + 1] += outer[static_cast<size_t>(i)];
    std::vector<int32_t> col(t.size());
    std::vector<Scalar> val(t.size());
    {
      std::vector<int32_t> at(outer.begin(), outer.end() - 1);
      for (const Triplet &e : t) {
        const int32_t w = at[static_cast<size_t>(e.r)]++;
        col[static_cast<size_t>(w)] = static_cast<int32_t>(e.c);
        val[static_cast<size_t>(w)] = e.v;
      }
    }
    t = std::vector<Triplet>();
    inner.reserve(col.size());
    values.reserve(col.size());
    std::vector<int32_t> ord;
    int32_t begin = 0;
    for (Index i = 0; i < rows_; ++i) {
      const int32_t end = outer[static_cast<size_t>(i) + 1];
      const int32_t n = end - begin;
      bool sorted = true;
      for (int32_t q = begin + 1; q < end && sorted; ++q) sorted = col[q - 1] < col[q];
      const int32_t out0 = static_cast<int32_t>(inner.size());
      if (sorted) {  // strictly ascending: no duplicates either
        inner.insert(inner.end(), col.begin() + begin, col.begin() + end);
        values.insert(values.end(), val.begin() + begin, val.begin() + end);
      } else {
        ord.resize(static_cast<size_t>(n));
        for (int32_t q = 0; q < n; ++q) ord[q] = begin + q;
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return col[x] < col[y]; });
        for (int32_t q = 0; q < n; ++q) {
          const int32_t e = ord[q];
          if (q > 0 && col[e] == inner.back()) {
            values.back() += val[e];
          } else {
            inner.push_back(col[e]);
            values.push_back(val[e]);
          }
        }
      }
      begin = end;
      outer[static_cast<size_t>(i)] = out0;
    }
    outer[static_cast<size_t>(rows_)] = static_cast<int32_t>(inner.size());
  }
  std::vector<Triplet> triplets(Index row_off = 0, Index col_off = 0, bool transpose = false) const {
    std::vector<Triplet> t;
    t.reserve(inner.size());
    for (Index i = 0; i < rows_; ++i)
      for (int32_t q = outer[static_cast<size_t>(i)]; q < outer[static_cast<size_t>(i) + 1]; ++q) {
        if (transpose) t.push_back({inner[q] + row_off, i + col_off, values[q]});
        else t.push_back({i + row_off, inner[q] + col_off, values[q]});
      }
    return t;
  }
  SparseMatrix transpose() const {
    SparseMatrix r(cols_, rows_);
    r.setFromTriplets(triplets(0, 0, true));
    return r;
  }
  /** this * diag(w) * other, without pruning (Eigen's conservative product). */
  SparseMatrix times(const std::vector<Scalar> *w, const SparseMatrix &o) const {
    if (cols_ != o.rows_) throw std::invalid_argument("SparseMatrix product: inner dimensions differ");
    std::vector<Triplet> t;
    {
      size_t count = 0;  // one pass for the size: growing a vector of 24-byte triplets copies it several times
      for (size_t q = 0; q < inner.size(); ++q)
        count += static_cast<size_t>(o.outer[static_cast<size_t>(inner[q]) + 1] - o.outer[static_cast<size_t>(inner[q])]);
      t.reserve(count);
    }
    for (Index i = 0; i < rows_; ++i)
      for (int32_t q = outer[static_cast<size_t>(i)]; q < outer[static_cast<size_t>(i) + 1]; ++q) {
        const Index k = inner[q];
        const Scalar a = values[q] * (w ? (*w)[static_cast<size_t>(k)] : 1.0);
        for (int32_t s = o.outer[static_cast<size_t>(k)]; s < o.outer[static_cast<size_t>(k) + 1]; ++s)
          t.push_back({i, o.inner[s], a * o.values[s]});
      }
    SparseMatrix r(rows_, o.cols_);
    r.setFromTriplets(std::move(t));
    return r;
  }
  SparseMatrix plus(const SparseMatrix &o) const {
    auto t = triplets();
    auto u = o.triplets();
    t.insert(t.end(), u.begin(), u.end());
    SparseMatrix r(rows_, cols_);
    r.setFromTriplets(std::move(t));
    return r;
  }
  /** Dense copy (Eigen's SparseMatrix::toDense(): the reference's tests read their dense fixtures through it). */
  Matrix toDense() const {
    Matrix D(rows_, cols_);
    for (Index i = 0; i < rows_; ++i)
      for (int32_t q = outer[static_cast<size_t>(i)]; q < outer[static_cast<size_t>(i) + 1]; ++q) D(i, inner[q]) += values[q];
    return D;
  }

 private:
  /** Position of (i, j) in inner / values, or where it would go. */
  bool find(Index i, Index j, int32_t *at) const {
    if (i < 0 || i >= rows_ || j < 0 || j >= cols_) throw std::out_of_range("SparseMatrix: index out of range");
    const auto b = inner.begin() + outer[static_cast<size_t>(i)], e = inner.begin() + outer[static_cast<size_t>(i) + 1];
    const auto it = std::lower_bound(b, e, static_cast<int32_t>(j));
    *at = static_cast<int32_t>(it - inner.begin());
    return it != e && *it == j;
  }

 public:
  Matrix operator*(const Matrix &X) const {  // CPU product for tiny host-side checks only
    Matrix r(rows_, X.cols());
    for (Index j = 0; j < X.cols(); ++j)
      for (Index i = 0; i < rows_; ++i) {
        Scalar s = 0;
        for (int32_t q = outer[static_cast<size_t>(i)]; q < outer[static_cast<size_t>(i) + 1]; ++q)
          s += values[q] * X(inner[q], j);
        r(i, j) = s;
      }
    return r;
  }
};

inline Matrix::Matrix(const SparseMatrix &S) { *this = S.toDense(); }
inline bool SparseMatrix::isApprox(const SparseMatrix &o, Scalar prec) const {
  if (rows_ != o.rows_ || cols_ != o.cols_) return false;
  return toDense().isApprox(o.toDense(), prec);
}
inline SparseMatrix Matrix::sparseView() const {
  std::vector<Triplet> t;
  for (Index i = 0; i < rows_; ++i)
    for (Index j = 0; j < cols_; ++j)
      if ((*this)(i, j) != 0.0) t.push_back({i, j, (*this)(i, j)});
  SparseMatrix S(rows_, cols_);
  S.setFromTriplets(std::move(t));
  return S;
}
inline Matrix operator*(const Matrix &A, const SparseMatrix &S) { return (S.transpose() * A.transpose()).transpose(); }
inline std::ostream &operator<<(std::ostream &os, const Matrix &M) {
  for (Index i = 0; i < M.rows(); ++i) {
    for (Index j = 0; j < M.cols(); ++j) os << (j ? " " : "") << M(i, j);
    if (i + 1 < M.rows()) os << "\n";
  }
  return os;
}

enum class Formulation { Explicit, Implicit };  // include/CORA/CORA_types.h:50-55

struct CertResults {  // include/CORA/CORA_types.h:58-64
  bool is_certified;
  Scalar theta;
  Vector x;
  Matrix all_eigvecs;
  size_t num_iters;
};

enum class StiefelRetraction { QR, Polar };
enum class ObliqueRetraction { Normalize };
/** include/CORA/CORA_types.h:77 */
enum class Preconditioner { None, Jacobi, BlockCholesky, RegularizedCholesky };
enum class Initialization { Random, Odometry };

}  // namespace CORA
