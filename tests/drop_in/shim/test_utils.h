// Stand-in for the reference's tests/test_utils.h -- TEST INFRASTRUCTURE of the drop-in check
// (tests/test_dropin_reference_tests.py).  The reference's helper is written against Eigen (loadMarket, selfadjointView,
// InnerIterator, cwiseAbs); this one offers the same names with the same meaning on the Eigen-free types of
// include/CORA/CORA_types.h, so that the reference's test translation units compile unmodified:
//   readMatrixMarketFile / getTestDataFpath / getProblem / getRandInit / getGroundTruthState / getRandDX /
//   getExpected{Cost,Egrad,Rgrad,HessProd,RandCertMatrix} / checkSubmatricesAreCorrect / IsApproximatelyEqual[UpToSign]
// (reference tests/test_utils.h:20-137, tests/test_utils.cpp:22-241).  Data files are looked up where the reference looks
// them up: <current directory>/bin/data/<subdir>/<file>.
#pragma once

#include <CORA/CORA_problem.h>
#include <CORA/CORA_types.h>
#include <CORA/pyfg_text_parser.h>

#include <cmath>
#include <sstream>
#include <string>

#include <catch2/catch_test_macros.hpp>
#include <catch2/matchers/catch_matchers.hpp>
#include <catch2/matchers/catch_matchers_floating_point.hpp>

// |expected - actual| <= epsilon entry by entry (dense: every entry; sparse: every entry of the difference)
template <typename MatrixType>
struct EigenMatrixApproxMatcher : Catch::Matchers::MatcherBase<MatrixType> {
  explicit EigenMatrixApproxMatcher(const MatrixType &expected, double epsilon = 1e-6) : expected_(expected), epsilon_(epsilon) {}
  bool match(const MatrixType &actual) const override;
  std::string describe() const override {
    std::ostringstream oss;
    oss << "is approximately equal to the expected " << expected_.rows() << " x " << expected_.cols() << " matrix within epsilon "
        << epsilon_ << " (largest deviation " << worst_ << ")";
    return oss.str();
  }

 private:
  const MatrixType &expected_;
  double epsilon_;
  mutable double worst_ = 0.0;
};

template <>
inline bool EigenMatrixApproxMatcher<CORA::Matrix>::match(const CORA::Matrix &actual) const {
  if (expected_.rows() != actual.rows() || expected_.cols() != actual.cols()) return false;
  worst_ = 0.0;
  for (CORA::Index k = 0; k < actual.size(); ++k) worst_ = std::max(worst_, std::fabs(expected_(k) - actual(k)));
  return worst_ <= epsilon_;
}
template <>
inline bool EigenMatrixApproxMatcher<CORA::SparseMatrix>::match(const CORA::SparseMatrix &actual) const {
  if (expected_.rows() != actual.rows() || expected_.cols() != actual.cols()) return false;
  worst_ = 0.0;
  CORA::SparseMatrix neg = actual;
  for (auto &v : neg.values) v = -v;
  const CORA::SparseMatrix diff = expected_.plus(neg);
  for (double v : diff.values) worst_ = std::max(worst_, std::fabs(v));
  return worst_ <= epsilon_;
}

template <typename MatrixType>
EigenMatrixApproxMatcher<MatrixType> IsApproximatelyEqual(const MatrixType &expected, double epsilon = 1e-6) {
  return EigenMatrixApproxMatcher<MatrixType>(expected, epsilon);
}

template <typename MatrixType>
struct EigenMatrixApproxMatcherUpToSign : Catch::Matchers::MatcherBase<MatrixType> {
  explicit EigenMatrixApproxMatcherUpToSign(const MatrixType &expected, double epsilon = 1e-6) : expected_(expected), epsilon_(epsilon) {}
  bool match(const MatrixType &actual) const override {
    if (expected_.rows() != actual.rows() || expected_.cols() != actual.cols()) return false;
    double minus = 0.0, plus = 0.0;
    for (CORA::Index k = 0; k < actual.size(); ++k) {
      minus = std::max(minus, std::fabs(expected_(k) - actual(k)));
      plus = std::max(plus, std::fabs(expected_(k) + actual(k)));
    }
    return minus <= epsilon_ || plus <= epsilon_;
  }
  std::string describe() const override { return "is approximately equal to the expected matrix up to its sign"; }

 private:
  const MatrixType &expected_;
  double epsilon_;
};
template <typename MatrixType>
EigenMatrixApproxMatcherUpToSign<MatrixType> IsApproximatelyEqualUpToSign(const MatrixType &expected, double epsilon = 1e-6) {
  return EigenMatrixApproxMatcherUpToSign<MatrixType>(expected, epsilon);
}

// The one Eigen type the reference's tests name directly (tests/test_construct_problem.cpp:70-71: a random orthogonal matrix
// from the left singular vectors of a small dense matrix).  Stand-in with the same spelling for the Eigen-free Matrix:
// U = an orthonormal basis whose leading columns span range(A) (Gram-Schmidt, completed when A is rank deficient).
namespace Eigen {
enum { ComputeThinU = 1, ComputeThinV = 2, ComputeFullU = 4, ComputeFullV = 8 };
template <typename MatrixType>
class JacobiSVD {
  MatrixType U_;

 public:
  JacobiSVD(const MatrixType &A, unsigned = 0) {
    const CORA::Index n = A.rows();
    U_ = MatrixType::Identity(n, n);
    // Gram-Schmidt of [A | I]: an orthonormal basis whose leading columns span range(A) -- left singular vectors up to a
    // rotation inside the singular subspaces, which is all the tests use (an orthogonal matrix)
    MatrixType B(n, A.cols() + n);
    for (CORA::Index j = 0; j < A.cols(); ++j)
      for (CORA::Index i = 0; i < n; ++i) B(i, j) = A(i, j);
    for (CORA::Index j = 0; j < n; ++j) B(j, A.cols() + j) = 1.0;
    CORA::Index k = 0;
    for (CORA::Index j = 0; j < B.cols() && k < n; ++j) {
      MatrixType v = static_cast<const MatrixType &>(B).col(j);
      for (CORA::Index q = 0; q < k; ++q) {
        const MatrixType u = static_cast<const MatrixType &>(U_).col(q);
        v = v - u * u.dot(v);
      }
      const double nv = v.norm();
      if (nv > 1e-12) {
        for (CORA::Index i = 0; i < n; ++i) U_(i, k) = v(i) / nv;
        ++k;
      }
    }
  }
  const MatrixType &matrixU() const { return U_; }
};
}  // namespace Eigen

namespace CORA {

SparseMatrix readMatrixMarketFile(const std::string &filename);
std::string getTestDataFpath(const std::string &data_subdir, const std::string &fname);
std::string checkSubmatricesAreCorrect(Problem prob, const std::string &data_subdir);

Problem getProblem(std::string data_subdir);
Matrix getRandInit(std::string data_subdir);
Matrix getGroundTruthState(std::string data_subdir);
Matrix getRandDX(std::string data_subdir);

SparseMatrix getExpectedRandCertMatrix(std::string data_subdir);
Scalar getExpectedCost(std::string data_subdir);
Matrix getExpectedEgrad(std::string data_subdir);
Matrix getExpectedRgrad(std::string data_subdir);
Matrix getExpectedHessProd(std::string data_subdir);

}  // namespace CORA
