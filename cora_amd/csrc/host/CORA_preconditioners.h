// Block-diagonal Cholesky preconditioner as free functions, the public interface of the reference's
// include/CORA/CORA_preconditioners.h:24-44 (src/CORA_preconditioners.cpp:16-83).  Host code on the
// CHOLMOD stand-in of sparse_cholesky.h; CORA::Problem does not go through these -- its preconditioner is
// factored once on the host and applied on the device (CORA_problem.cpp, trisolve.h) -- they exist so that
// code written against the reference's helpers (tests/test.cpp:149-214) keeps working.
#pragma once

#include <memory>
#include <vector>

#include "CORA_types.h"
#include "sparse_cholesky.h"

namespace CORA {

using CholFactorPtr = std::shared_ptr<CholeskyFactor>;
using CholFactorPtrVector = std::vector<CholFactorPtr>;

/** LL^T of every diagonal block of A; block_sizes must sum to A.rows().  The blocks are the documented
 * ones, block b starting where block b-1 ends (the reference's loop never advances its block_start,
 * src/CORA_preconditioners.cpp:30-41, and factors the first block again: not reproduced). */
CholFactorPtrVector getBlockCholeskyFactorization(const SparseMatrix &A, const VectorXi &block_sizes);

/** Block-wise solve; rhs has as many rows as the factors together, or one more, in which case the last row of
 * the result is zero (the pinned translation, :46-83). */
Matrix blockCholeskySolve(const CholFactorPtrVector &block_chol_factor_ptrs, const Matrix &rhs);

}  // namespace CORA
