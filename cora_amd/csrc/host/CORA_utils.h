// fast_verification / projectToSOd (reference include/CORA/CORA_utils.h:35-64,
// src/CORA_utils.cpp:17-202).
#pragma once

#include <memory>
#include <optional>
#include <vector>

#include "CORA_problem.h"
#include "CORA_types.h"
#include "LOBPCG.h"

namespace CORA {

/**
 * Algorithm 3 of "Accelerating Certifiable Estimation with Preconditioned
 * Eigensolvers" as the reference implements it (src/CORA_utils.cpp:17-186):
 *  1. PSD test of M = S + eta I by sparse Cholesky (success <=> is_certified);
 *  2. if not PSD and n <= 100: dense eigen-decomposition of S (theta = lambda_min);
 *  3. otherwise LOBPCG on M (block X0, nev = 1) until x' S x < -eta / 2.
 * The operator S*X runs on the GPU: `op` when given (the problem's own handle,
 * certificate operator at its current point), else a temporary handle built from S.
 * `perm` (new -> old) is the fill-reducing order for step 1 (natural order if empty).
 * After 1 % of the iterations without a direction of curvature < -eta/2 the reference switches to
 * ILDL-preconditioned LOBPCG (:140-167); so does this: incomplete L D L^T of S + eta I with max_fill_factor /
 * drop_tol (sparse_cholesky.h, incompleteLDLT), applied on the device; `precond`, when given, replaces it.
 */
/** Test switches of step 3 (tests/test_gpu_certification.py): run it without the seed from the failed factorisation
 * and / or without the ILDL preconditioner, and learn whether it ran. */
struct FastVerificationLab {
  bool seed_negative_direction = true, use_ildl = true;
  mutable bool reached_step3 = false;
};

CertResults fast_verification(const SparseMatrix &S, Scalar eta, const Matrix &X0, size_t max_iters = 1000,
                              const std::vector<int32_t> &perm = {}, cora_ctx *ctx = nullptr,
                              const std::optional<DeviceOperator> &S_op = std::nullopt,
                              const std::optional<DeviceOperator> &precond = std::nullopt,
                              Scalar max_fill_factor = 3, Scalar drop_tol = 1e-3, const FastVerificationLab *lab = nullptr,
                              SymbolicCache *symbolic = nullptr);

/** The same with the start block given as pieces of host memory put side by side (columns of the previous level's
 * eigenvectors, cached random columns): nothing of size N x m is assembled on the host. */
CertResults fast_verification(const SparseMatrix &S, Scalar eta, const std::vector<HostColumns> &X0, size_t max_iters = 1000,
                              const std::vector<int32_t> &perm = {}, cora_ctx *ctx = nullptr,
                              const std::optional<DeviceOperator> &S_op = std::nullopt,
                              const std::optional<DeviceOperator> &precond = std::nullopt,
                              Scalar max_fill_factor = 3, Scalar drop_tol = 1e-3, const FastVerificationLab *lab = nullptr,
                              SymbolicCache *symbolic = nullptr, std::shared_ptr<LOBPCGSolver> *keep_block = nullptr);
/* keep_block != nullptr: the Ritz block of the eigensolver is NOT brought to the host (results.all_eigvecs stays empty,
 * results.x is its first column): the solver that holds it on the device is handed over instead, to be the start block
 * of the next certification (HostColumns::device) -- solveCORA needs nothing else from it.  nullptr is handed over when
 * no eigensolver ran (certified, or the dense path of small matrices, which fills all_eigvecs as usual). */

inline CertResults fast_verification(const SparseMatrix &S, Scalar eta, size_t nx, size_t max_iters = 1000) {
  return fast_verification(S, eta, Matrix::Random(S.rows(), static_cast<Index>(nx)), max_iters);
}

/** Nearest rotation to a d x d matrix (src/CORA_utils.cpp:188-202). */
Matrix projectToSOd(const Matrix &M);

}  // namespace CORA
