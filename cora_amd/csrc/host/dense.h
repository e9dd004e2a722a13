// Small dense helpers for the host side (p x p, 3m x 3m problems).
#pragma once

#include <cmath>
#include <numeric>
#include <vector>

#include "CORA_types.h"

namespace CORA {

/** Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix.
 * evals ascending, evecs column k = eigenvector k. */
inline void symmetricEigen(const Matrix &A_in, Vector &evals, Matrix &evecs) {
  const Index n = A_in.rows();
  Matrix A = A_in;
  Matrix V = Matrix::Identity(n, n);
  for (int sweep = 0; sweep < 100; ++sweep) {
    Scalar off = 0, diag = 0;
    for (Index i = 0; i < n; ++i) {
      diag += A(i, i) * A(i, i);
      for (Index j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
    }
    if (off <= 1e-30 * (diag + off) || off == 0) break;
    for (Index p = 0; p < n - 1; ++p)
      for (Index q = p + 1; q < n; ++q) {
        const Scalar apq = A(p, q);
        if (apq == 0) continue;
        const Scalar zeta = (A(q, q) - A(p, p)) / (2 * apq);
        const Scalar t = (zeta >= 0 ? 1.0 : -1.0) / (std::abs(zeta) + std::sqrt(1 + zeta * zeta));
        const Scalar c = 1 / std::sqrt(1 + t * t), s = c * t;
        for (Index k = 0; k < n; ++k) {
          const Scalar akp = A(k, p), akq = A(k, q);
          A(k, p) = c * akp - s * akq;
          A(k, q) = s * akp + c * akq;
        }
        for (Index k = 0; k < n; ++k) {
          const Scalar apk = A(p, k), aqk = A(q, k);
          A(p, k) = c * apk - s * aqk;
          A(q, k) = s * apk + c * aqk;
        }
        for (Index k = 0; k < n; ++k) {
          const Scalar vkp = V(k, p), vkq = V(k, q);
          V(k, p) = c * vkp - s * vkq;
          V(k, q) = s * vkp + c * vkq;
        }
      }
  }
  std::vector<Index> idx(static_cast<size_t>(n));
  std::iota(idx.begin(), idx.end(), Index{0});
  std::sort(idx.begin(), idx.end(), [&](Index a, Index b) { return A(a, a) < A(b, b); });
  evals = Vector(n, 1);
  evecs = Matrix(n, n);
  for (Index k = 0; k < n; ++k) {
    evals(k) = A(idx[k], idx[k]);
    for (Index i = 0; i < n; ++i) evecs(i, k) = V(i, idx[k]);
  }
}

inline Scalar determinant(const Matrix &M) {  // d <= 3
  if (M.rows() == 2) return M(0, 0) * M(1, 1) - M(0, 1) * M(1, 0);
  return M(0, 0) * (M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) - M(0, 1) * (M(1, 0) * M(2, 2) - M(1, 2) * M(2, 0)) +
         M(0, 2) * (M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0));
}

}  // namespace CORA
