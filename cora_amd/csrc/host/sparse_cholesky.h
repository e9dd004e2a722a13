// Sparse Cholesky on the host: the stand-in for CHOLMOD, which the reference
// reaches through Eigen::CholmodDecomposition (preconditioner,
// include/CORA/CORA_preconditioners.h:24-26, src/CORA_problem.cpp:544-614) and
// Eigen::CholmodSupernodalLLT (PSD test of the certificate,
// src/CORA_utils.cpp:36-51).  SuiteSparse is a system dependency of the
// reference and is not in this image, so this is a from-scratch up-looking
// factorisation (T. A. Davis, "Direct Methods for Sparse Linear Systems", ch. 4)
// with a fill-reducing order tailored to CORA's graphs: nested dissection along
// the pose chain, landmarks last.  What the reference observes -- the solution
// of a solve and whether the factorisation succeeds -- does not depend on the
// ordering.
#pragma once

#include <cstdint>
#include <vector>

#include "CORA_types.h"

namespace CORA {

struct CholeskyFactor {
  int n = 0;
  bool ok = false;            // false <=> not (numerically) positive definite
  int failed_column = -1;     // permuted index of the first non-positive pivot
  std::vector<int32_t> Lp;    // CSC of L, diagonal first in every column
  std::vector<int32_t> Li;
  std::vector<double> Lx;
  std::vector<int32_t> perm;  // new -> old
  std::vector<int32_t> iperm; // old -> new
  std::vector<int32_t> parent;  // elimination tree
  /** When !ok: unit vector z (original order, size = A.rows()) with z^T (A + shift I) z <= 0, built
   * from the failing pivot: z = [-L11^-T l_k; 1; 0] has z^T M z = d_k <= 0. */
  std::vector<double> negative_direction;
  int64_t nnz() const { return static_cast<int64_t>(Li.size()); }
  /** Solve A X = B in place (B: n x k, column-major). */
  void solveInPlace(Matrix &B) const;
};

/** Order of the leading m variables (m = N or N-1) of a CORA problem with d, n
 * poses, r ranges, nt translations: nested dissection over the pose index with
 * single-pose separators (each pose carries its d rotation rows, the range rows
 * hanging off it and its translation), then pose-less range rows and landmarks.
 * Q is only used to attach range rows to poses. Returns new -> old. */
std::vector<int32_t> coraOrdering(int d, int n, int r, int nt, const SparseMatrix &Q, int m, int leaf_poses = 16);

/** LL^T of the leading m x m block of the symmetric matrix A (full pattern,
 * both triangles) plus shift * I, in the order perm (new -> old). */
CholeskyFactor choleskyFactor(const SparseMatrix &A, int m, double shift, const std::vector<int32_t> &perm);

}  // namespace CORA
