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
  Matrix X;      // N x m Ritz vectors (host, API row order); empty when the run was asked to leave them on the device
  size_t num_iters = 0;
  size_t num_converged = 0;
};

/** Columns of a start block that live in host memory: `cols` columns of N doubles, leading dimension N. */
struct HostColumns {
  const double *data = nullptr;
  int cols = 0;
  // non-null: the piece is the leading `cols` columns of a block that is ALREADY on the device (N rows, `device_width`
  // columns, resident layout) -- e.g. the Ritz block the previous certification left there (LOBPCGSolver::deviceBlock):
  // nothing is uploaded, the columns are picked up where they are
  const double *device = nullptr;
  int device_width = 0;
};

/** One LOBPCG run whose blocks stay on the device until they are asked for.  The start block is the concatenation of up
 * to four pieces (e.g. the previous level's eigenvectors | cached random columns | one seed vector): they are uploaded
 * as they are and put side by side on the device, nothing is assembled on the host.  `block()` / `column(j)` download
 * the Ritz vectors of the last run. */
class LOBPCGSolver {
 public:
  LOBPCGSolver(cora_ctx *ctx, int N);
  ~LOBPCGSolver();
  LOBPCGSolver(const LOBPCGSolver &) = delete;
  LOBPCGSolver &operator=(const LOBPCGSolver &) = delete;
  LOBPCGResult run(const DeviceOperator &A, const std::optional<DeviceOperator> &T, const std::vector<HostColumns> &start,
                   size_t nev, size_t max_iters, Scalar tau, const std::optional<LOBPCGStop> &stop, bool download);
  Matrix block() const;
  Vector column(int j) const;
  int width() const { return m_; }
  /** The Ritz block of the last run where it lives: N rows, width() columns, resident layout (nullptr before a run). */
  const double *deviceBlock() const { return X_; }

 private:
  struct Impl;
  cora_ctx *c_;
  int N_, m_ = 0;
  std::vector<double *> owned_;
  double *X_ = nullptr;
};

/** X0: N x m (host).  nev: wanted pairs; tau: residual tolerance ||r|| <= tau |theta|. */
LOBPCGResult LOBPCG(cora_ctx *ctx, const DeviceOperator &A, const std::optional<DeviceOperator> &T,
                    const Matrix &X0, size_t nev, size_t max_iters, Scalar tau,
                    const std::optional<LOBPCGStop> &stop = std::nullopt);

}  // namespace CORA
