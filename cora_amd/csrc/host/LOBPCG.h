// Locally optimal block preconditioned conjugate gradient (Knyazev 2001) for the
// algebraically smallest eigenpairs of a symmetric operator.
//
// The reference uses Optimization::LinearAlgebra::LOBPCG from the un-vendored
// submodule libs/Optimization (call sites src/CORA_utils.cpp:113-119,160-167,
// src/CORA_problem.cpp:567-575; sources absent).  This is a from-scratch
// implementation of the published method: the N x m blocks stay resident on the
// GPU, the operator is a device callback, Rayleigh-Ritz Gram matrices and block
// updates are the `cora_gram_dev` / `cora_combine_dev` kernels, and only the
// (<= 3m x 3m) dense eigenproblem runs on the host.
#pragma once

#include <functional>
#include <optional>
#include <vector>

#include "CORA_types.h"

struct cora_ctx;

namespace CORA {

/** dOut = Op(dX) for k-column resident blocks. */
using DeviceOperator = std::function<void(const double *dX, int k, double *dOut)>;
/** Mirrors LOBPCGUserFunction: return true to stop (i = iteration, Theta = Ritz values,
 * dX = current block, k its width). */
using LOBPCGStop = std::function<bool(size_t i, const std::vector<Scalar> &Theta, const double *dX, int k)>;

struct LOBPCGResult {
  Vector Theta;  // m Ritz values, ascending
  Matrix X;      // N x m Ritz vectors (host, API row order)
  size_t num_iters = 0;
  size_t num_converged = 0;
};

/** X0: N x m (host).  nev: wanted pairs; tau: residual tolerance ||r|| <= tau |theta|. */
LOBPCGResult LOBPCG(cora_ctx *ctx, const DeviceOperator &A, const std::optional<DeviceOperator> &T,
                    const Matrix &X0, size_t nev, size_t max_iters, Scalar tau,
                    const std::optional<LOBPCGStop> &stop = std::nullopt);

}  // namespace CORA
