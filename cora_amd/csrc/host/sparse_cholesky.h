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
  CholeskyFactor() = default;
  CholeskyFactor(const CholeskyFactor &) = default;
  CholeskyFactor(CholeskyFactor &&) = default;
  CholeskyFactor &operator=(const CholeskyFactor &) = default;
  CholeskyFactor &operator=(CholeskyFactor &&) = default;
  ~CholeskyFactor();  // hands Li / Lx back to the storage pool of choleskyFactor
  /** Solve A X = B in place (B: n x k, column-major). */
  void solveInPlace(Matrix &B) const;
};

/** Order of the leading m variables (m = N or N-1) of a CORA problem with d, n
 * poses, r ranges, nt translations: nested dissection over the pose index with
 * single-pose separators (each pose carries its d rotation rows, the range rows
 * hanging off it and its translation), then pose-less range rows and landmarks.
 * Q is only used to attach range rows to poses. Returns new -> old. */
std::vector<int32_t> coraOrdering(int d, int n, int r, int nt, const SparseMatrix &Q, int m, int leaf_poses = 16);

/** The pattern-only part of a factorisation (permuted structure, elimination tree, column counts: CHOLMOD's analyze
 * step), kept between factorisations of matrices with one pattern and order.  Owned by whoever owns the matrices -- a
 * CORA::Problem keeps one for its preconditioner block and its certificate matrix -- and gone with it; holds the two most
 * recent patterns, keyed by two independent hashes of (pattern, order).  Thread-safe. */
class SymbolicCache {
 public:
  SymbolicCache();
  ~SymbolicCache();
  SymbolicCache(const SymbolicCache &) = delete;
  SymbolicCache &operator=(const SymbolicCache &) = delete;
  struct Impl;
  Impl *impl;
};

/** LL^T of the leading m x m block of the symmetric matrix A (full pattern,
 * both triangles) plus shift * I, in the order perm (new -> old).  cache: where the symbolic analysis is looked up
 * and left (nullptr: analysed afresh every time); the factor is the same bit for bit either way. */
CholeskyFactor choleskyFactor(const SparseMatrix &A, int m, double shift, const std::vector<int32_t> &perm,
                              SymbolicCache *cache = nullptr);

/** The part of choleskyFactor(A, m, *, perm, cache) that depends on the pattern and the order alone, done ahead of
 * time: the symbolic analysis goes into `cache` and storage for the factor is touched once and left in the pool, so the
 * factorisation that follows finds both (CORA::Problem::prepareCertification).  No-op without a cache. */
void choleskyAnalyze(const SparseMatrix &A, int m, const std::vector<int32_t> &perm, SymbolicCache *cache);
/** First touch of storage for a factor of up to `entries` entries, left in the pool the factorisation draws from: a guess
 * made before the symbolic analysis knows the count (run beside it: the touch of 58 MB was 6 ms at the end of the chain
 * the first factorisation waits for).  Too small a guess costs nothing but the touch. */
void choleskyReserveStorage(size_t entries);

/** Incomplete L D L^T of the symmetric (possibly INDEFINITE) matrix A[0:m, 0:m] + shift * I in the order perm --
 * the stand-in for Preconditioners::ILDL (libs/Preconditioners, un-vendored submodule; reference call
 * src/CORA_utils.cpp:140-156 with ILDLOpts{max_fill_factor, drop_tol}).  Left-looking (Crout) elimination with
 * 1x1 pivots: column k keeps its entries above drop_tol * ||column||_1, at most max_fill_factor * (entries of
 * column k of A) of them; a pivot smaller than 1e-10 * ||column|| is moved away from zero.  What comes back is the
 * factor of the POSITIVE-DEFINITE modification the reference applies (ILDL::solve(x, pos_def_mod = true): |D| in
 * place of D), in Cholesky form:  Lx holds  L |D|^(1/2)  (CSC, diagonal first), so that
 * (L |D| L^T)^-1 = (Lx Lx^T)^-1 is applied by the same triangular solves as a Cholesky factor.  `negative_pivots`
 * counts the d_k < 0 (the inertia estimate). */
CholeskyFactor incompleteLDLT(const SparseMatrix &A, int m, double shift, const std::vector<int32_t> &perm,
                              double max_fill_factor, double drop_tol, int *negative_pivots = nullptr);

}  // namespace CORA
