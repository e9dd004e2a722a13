#include "CORA_preconditioners.h"

#include <numeric>
#include <stdexcept>
#include <string>

namespace CORA {

CholFactorPtrVector getBlockCholeskyFactorization(const SparseMatrix &A, const VectorXi &block_sizes) {
  const long sum = std::accumulate(block_sizes.begin(), block_sizes.end(), 0L);
  if (sum != A.rows())
    throw std::invalid_argument("The block sizes must sum to A.rows() for the CORA block Cholesky preconditioner. "
                                "Block sizes sum: " + std::to_string(sum) + ", A.rows(): " + std::to_string(A.rows()));
  CholFactorPtrVector factors;
  Index start = 0;
  for (int size : block_sizes) {
    std::vector<Triplet> t;
    for (Index i = 0; i < size; ++i)
      for (int32_t q = A.outer[start + i]; q < A.outer[start + i + 1]; ++q) {
        const Index j = A.inner[q] - start;
        if (j >= 0 && j < size) t.push_back({i, j, A.values[q]});
      }
    SparseMatrix block(size, size);
    block.setFromTriplets(std::move(t));
    std::vector<int32_t> perm(static_cast<size_t>(size));
    std::iota(perm.begin(), perm.end(), 0);
    auto F = std::make_shared<CholeskyFactor>(choleskyFactor(block, size, 0.0, perm));
    if (!F->ok) throw std::runtime_error("getBlockCholeskyFactorization: a diagonal block is not positive definite");
    factors.push_back(std::move(F));
    start += size;
  }
  return factors;
}

Matrix blockCholeskySolve(const CholFactorPtrVector &factors, const Matrix &rhs) {
  Index rows = 0;
  for (const auto &F : factors) rows += F->n;
  const bool same = rhs.rows() == rows, one_more = rhs.rows() == rows + 1;
  if (!same && !one_more)
    throw std::invalid_argument(
        "The number of rows in the right-hand side must be equal to the sum of the number of rows in the block "
        "Cholesky factors or one more row than the sum of the number of rows in the block Cholesky factors.");
  Matrix result(rhs.rows(), rhs.cols());
  Index start = 0;
  for (const auto &F : factors) {
    Matrix b = rhs.block(start, 0, F->n, rhs.cols());
    F->solveInPlace(b);
    result.setBlock(start, 0, b);
    start += F->n;
  }
  return result;  // a trailing extra row stays zero
}

}  // namespace CORA
