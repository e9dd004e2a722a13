#include "CORA.h"

#include <stdexcept>

#include <algorithm>
#include <chrono>
#include <future>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/cora_hip.h"
#include "dense.h"
#include "../parallel.h"

namespace CORA {

namespace {
void printIfVerbose(bool verbose, const std::string &msg) {
  if (verbose) std::cout << msg << std::endl;
}
Scalar thresholdVal(Scalar v, Scalar lo, Scalar hi) { return v < lo ? lo : (v > hi ? hi : v); }
// CORA_TRACE_BITS=1: one line per stage of the staircase with the BITS of what it produced (FNV-1a over the matrix,
// hex floats), so that two runs can be diffed for the first stage at which they part (tools/determinism_probe.py)
uint64_t bitsOf(const Matrix &m) {
  uint64_t h = 1469598103934665603ull;
  const unsigned char *b = reinterpret_cast<const unsigned char *>(m.data());
  for (size_t i = 0; i < static_cast<size_t>(m.size()) * sizeof(Scalar); ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}
void traceBits(const char *stage, const Matrix &m, double a = 0, double b = 0, long n = 0) {
  static const bool on = std::getenv("CORA_TRACE_BITS") != nullptr;
  if (on) std::printf("[bits] %-14s %016llx  %a  %a  %ld\n", stage, static_cast<unsigned long long>(bitsOf(m)), a, b, n);
}
}  // namespace

CoraResult solveCORA(Problem &problem, const Matrix &x0, int max_relaxation_rank, bool verbose, bool log_iterates,
                     bool show_iterates, CoraSolveInfo *info, const TNTParams *params_override) {
  // resident device vectors carry at most 24 columns and saddleEscape lifts to rank + 1: say so before the
  // staircase starts rather than in the middle of it (the reference has no such cap; it never needs one either --
  // the staircase certifies at rank <= d + a few)
  // -- a cap, not a target: a generous max_relaxation_rank (the reference's default is 20, callers pass 25 or 50) is
  // clamped with a note, and only a staircase that really has to climb past rank 24 fails, when it gets there.
  if (x0.cols() > 24) throw std::invalid_argument("solveCORA: this build supports relaxation ranks up to 24");
  const int requested_max_rank = max_relaxation_rank;
  if (max_relaxation_rank > 23) {
    max_relaxation_rank = 23;
    printIfVerbose(verbose, "solveCORA: max_relaxation_rank " + std::to_string(requested_max_rank) +
                                " clamped to 23 (resident vectors carry at most 24 columns)");
  }
  if (problem.getFormulation() == Formulation::Explicit) {
    checkMatrixShape("solveCora::Explicit", problem.getDataMatrixSize(), x0.cols(), x0.rows(), x0.cols());
  } else {  // src/CORA.cpp:33-40
    std::cout << "Solving problem in translation implicit mode. Make sure that the initial guess only contains "
                 "rotation and range variables."
              << std::endl;
    checkMatrixShape("solveCora::Implicit", problem.rotAndRangeMatrixSize(), x0.cols(), x0.rows(), x0.cols());
  }
  if (log_iterates)
    std::cout << "WARNING: Logging iterates will slow down the optimization process.  This is intended for "
                 "debugging and viz purposes only."
              << std::endl;
  // TNT parameters: src/CORA.cpp:95-109
  TNTParams params;
  if (params_override) params = *params_override;
  params.verbose = show_iterates;
  params.log_iterates = log_iterates;
  // certification parameters: src/CORA.cpp:111-116
  const Scalar MIN_CERT_ETA = 1e-7, MAX_CERT_ETA = 1e-1, REL_CERT_ETA = 5e-6;
  const int LOBPCG_BLOCK_SIZE = 10;
  Scalar eta = 0;

  CoraTntResult result;
  Matrix X = problem.projectToManifold(x0);
  CertResults cert;
  cert.is_certified = false;
  cert.theta = 0;
  Matrix eigvec_bootstrap;
  std::vector<Matrix> iterates;
  bool first_loop = true;
  int levels = 0;
  long hvps = 0;
  double t_tnt = 0, t_cert = 0, t_escape = 0, t_project = 0;
  const auto t_begin = std::chrono::steady_clock::now();
  using clk = std::chrono::steady_clock;
  auto since = [](clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); };
  // What the first certification needs and the point does not decide (order, pattern and symbolic analysis of the
  // certificate matrix, storage of its factor, random start columns) is prepared on a thread of its own while the first
  // TNT solve keeps the device busy.  (The future's destructor joins: an exception below cannot leave the thread behind.)
  std::future<void> cert_prepared;
  if (!std::getenv("CORA_NO_CERT_PREPARE"))
    cert_prepared = std::async(std::launch::async, [&problem, LOBPCG_BLOCK_SIZE] {
      problem.prepareCertification(std::max<Index>(LOBPCG_BLOCK_SIZE, static_cast<Index>(problem.getRelaxationRank()) + 2));
    });
  while (static_cast<int>(problem.getRelaxationRank()) <= max_relaxation_rank) {
    ++levels;
    printIfVerbose(verbose, "\nSolving problem at rank " + std::to_string(problem.getRelaxationRank()));
    auto t0 = clk::now();
    result = TNT(problem, X, params);
    t_tnt += since(t0);
    if (cert_prepared.valid()) cert_prepared.get();
    hvps += result.hessian_vector_products;
    traceBits("TNT", result.x, result.f, result.gradfx_norm, result.hessian_vector_products);
    printIfVerbose(verbose, "Obtained solution with objective value: " + std::to_string(result.f));
    if (log_iterates)
      for (const Matrix &it : result.iterates) iterates.push_back(it);
    eta = thresholdVal(result.f * REL_CERT_ETA, MIN_CERT_ETA, MAX_CERT_ETA);
    if (first_loop) {
      eigvec_bootstrap = result.x;
      if (problem.getFormulation() == Formulation::Implicit)  // src/CORA.cpp:158-164
        eigvec_bootstrap = problem.getTranslationExplicitSolution(eigvec_bootstrap);
      first_loop = false;
    } else {
      eigvec_bootstrap = cert.all_eigvecs;
    }
    t0 = clk::now();
    // (the Ritz block stays on the device from one certification to the next: cert.all_eigvecs comes back empty, and
    // an empty bootstrap tells the next call to start from the block where it is)
    cert = problem.certify_solution_resident(result.x, eta, LOBPCG_BLOCK_SIZE, eigvec_bootstrap);
    t_cert += since(t0);
    traceBits("certify.x", cert.x, cert.theta, eta, static_cast<long>(cert.num_iters));
    traceBits("certify.block", cert.all_eigvecs, cert.theta, eta, cert.is_certified);
    printIfVerbose(verbose, "Result is certified: " + std::to_string(cert.is_certified) + " with eta: " +
                                std::to_string(eta) + " and theta: " + std::to_string(cert.theta));
    if (std::isnan(cert.theta)) throw std::runtime_error("Theta is NaN");
    if (cert.is_certified) {
      X = result.x;
      if (info) {
        info->relaxation_certified = true;
        info->relaxation_rank = static_cast<int>(problem.getRelaxationRank());
      }
      break;
    }
    const Scalar SADDLE_GRAD_TOL = 1e-4, PRECON_SADDLE_GRAD_TOL = 1e-4;
    problem.incrementRank();
    t0 = clk::now();
    X = saddleEscape(problem, result.x, cert.theta, cert.x, SADDLE_GRAD_TOL, PRECON_SADDLE_GRAD_TOL);
    t_escape += since(t0);
    traceBits("saddleEscape", X);
  }
  if (!cert.is_certified && requested_max_rank > 23 && static_cast<int>(problem.getRelaxationRank()) > 23)
    throw std::runtime_error("solveCORA: the staircase is not certified at rank 23 and max_relaxation_rank = " +
                             std::to_string(requested_max_rank) + " asks for more; this build supports ranks up to 24");
  // project to rank d and refine (src/CORA.cpp:198-233)
  if (X.cols() > problem.dim()) {
    printIfVerbose(verbose, "\nProjecting solution to rank " + std::to_string(problem.dim()) + " and refining.");
    auto tp = clk::now();
    X = projectSolution(problem, X, verbose);
    t_project = since(tp);
    traceBits("projectSolution", X);
    problem.setRank(problem.dim());
    auto t0 = clk::now();
    result = TNT(problem, X, params);
    t_tnt += since(t0);
    hvps += result.hessian_vector_products;
    traceBits("TNT final", result.x, result.f, result.gradfx_norm, result.hessian_vector_products);
    printIfVerbose(verbose, "\nObtained FINAL solution with objective value: " + std::to_string(result.f));
    if (log_iterates)
      for (const Matrix &it : result.iterates) iterates.push_back(it);
    printIfVerbose(verbose, "Checking certification of refined solution.");
    eta = thresholdVal(result.f * REL_CERT_ETA, MIN_CERT_ETA, MAX_CERT_ETA);
    t0 = clk::now();
    cert = problem.certify_solution(result.x, eta, LOBPCG_BLOCK_SIZE, Matrix());
    t_cert += since(t0);
  }
  printIfVerbose(verbose, "Final solution is certified: " + std::to_string(cert.is_certified) + " with eta: " +
                              std::to_string(eta) + " and theta: " + std::to_string(cert.theta));
  printIfVerbose(verbose, "Time: TNT " + std::to_string(t_tnt) + " s, certification " + std::to_string(t_cert) +
                              " s, saddle escape " + std::to_string(t_escape) + " s, rounding " + std::to_string(t_project) +
                              " s, all " + std::to_string(since(t_begin)) + " s, Hessian-vector products " + std::to_string(hvps));
  if (info) {
    info->certified = cert.is_certified;
    info->eta = eta;
    info->theta = cert.theta;
    info->final_rank = static_cast<int>(problem.getRelaxationRank());
    info->staircase_levels = levels;
    info->hessian_vector_products = hvps;
    info->tnt_seconds = t_tnt;
    info->certify_seconds = t_cert;
    info->escape_seconds = t_escape;
  }
  return std::make_pair(result, iterates);
}

Matrix saddleEscape(const Problem &problem, const Matrix &Y, Scalar theta, const Vector &v,
                    Scalar gradient_tolerance, Scalar preconditioned_gradient_tolerance) {
  // src/CORA.cpp:245-350
  const size_t r = problem.getRelaxationRank();
  if (static_cast<Index>(r) != Y.cols() + 1)
    throw std::runtime_error("Relaxation rank: " + std::to_string(r) +
                             " should be one greater than the number of columns in Y: " +
                             std::to_string(Y.cols()) +
                             ". This may happen if the relaxation rank is not incremented before attempting "
                             "saddle escape");
  Matrix Y_aug(Y.rows(), static_cast<Index>(r));
  Y_aug.setBlock(0, 0, Y);
  Matrix Ydot(Y.rows(), static_cast<Index>(r));
  for (Index i = 0; i < Y.rows(); ++i) Ydot(i, static_cast<Index>(r) - 1) = v(i);

  // The line search runs on resident vectors: the saddle point and the direction are uploaded once, every trial point
  // is a retraction, an objective and (only when the objective decreased -- the acceptance test is a conjunction) the
  // two gradient norms on the device; one download at the end.  Through the host-matrix interface every trial point
  // cost four round trips of an N x r matrix (0.2 s per trial at 10^6 poses).
  cora_ctx *c = problem.context();
  problem.ensurePreconditionerReady();
  const int p = static_cast<int>(r);
  struct Dev {
    cora_ctx *c;
    std::vector<double *> owned;
    ~Dev() {
      for (double *q : owned) cora_dev_free(c, q);
    }
    void chk(int rc, const char *what) const {
      if (rc != CORA_OK) throw std::runtime_error(std::string("saddleEscape: ") + what + ": " + cora_last_error(c));
    }
    double *alloc(int k) {
      double *q = nullptr;
      chk(cora_dev_alloc(c, k, &q), "cora_dev_alloc");
      owned.push_back(q);
      return q;
    }
  } D{c, {}};
  double *dY = D.alloc(p), *dV = D.alloc(p), *dT = D.alloc(p), *dPg = D.alloc(p);
  Matrix tmp_y, tmp_v;
  const Matrix &Yl = problem.lifted(Y_aug, tmp_y);  // implicit formulation: [Y; 0] on the device
  const Matrix &Vl = problem.lifted(Ydot, tmp_v);
  const int N = static_cast<int>(Yl.rows());
  D.chk(cora_upload(c, Yl.data(), N, p, dY), "cora_upload");
  D.chk(cora_upload(c, Vl.data(), N, p, dV), "cora_upload");
  D.chk(cora_set_point_dev(c, dY), "cora_set_point_dev");
  double FY = 0.0;
  D.chk(cora_point_cost(c, &FY), "cora_point_cost");
  auto download = [&](const double *d) {
    Matrix m(N, p);
    D.chk(cora_download(c, d, p, m.data(), N), "cora_download");
    return problem.lowered(std::move(m));
  };

  const Scalar alpha_min = 1e-6;
  Scalar alpha = std::max(16 * alpha_min, 100 * gradient_tolerance / std::fabs(theta));
  std::vector<double> alphas, fvals;
  while (alpha >= alpha_min) {
    D.chk(cora_retract_dev(c, dV, alpha, dT), "cora_retract_dev");  // R_Y(alpha Ydot), Y the current point
    double FYtest = 0.0;
    D.chk(cora_objective_dev(c, dT, &FYtest), "cora_objective_dev");
    alphas.push_back(alpha);
    fvals.push_back(FYtest);
    if (FYtest < FY) {
      D.chk(cora_set_point_dev(c, dT), "cora_set_point_dev");
      const double *grad = cora_point_rgrad_dev(c);
      D.chk(cora_precondition_projected_dev(c, grad, dPg), "precon");
      const double *A[2] = {grad, dPg};
      const double *B[2] = {grad, dPg};
      double o[2];
      D.chk(cora_dots_dev(c, 2, A, B, o), "cora_dots_dev");
      const Scalar gn = std::sqrt(std::max(o[0], 0.0)), pgn = std::sqrt(std::max(o[1], 0.0));
      if (gn > gradient_tolerance && pgn > preconditioned_gradient_tolerance) return download(dT);
      D.chk(cora_set_point_dev(c, dY), "cora_set_point_dev");  // back to the saddle point for the next retraction
    }
    alpha /= 2;
  }
  const auto it = std::min_element(fvals.begin(), fvals.end());
  const size_t k = static_cast<size_t>(std::distance(fvals.begin(), it));
  if (fvals[k] < FY) {
    D.chk(cora_retract_dev(c, dV, alphas[k], dT), "cora_retract_dev");
    return download(dT);
  }
  std::cout << "WARNING! BACKTRACKING LINE SEARCH FAILED TO ESCAPE FROM SADDLE POINT! (Try decreasing the "
               "preconditioned gradient norm tolerance)"
            << std::endl;
  return Y_aug;
}

Matrix projectSolution(const Problem &problem, const Matrix &Y, bool verbose) {
  // src/CORA.cpp:352-441.  Thin SVD of the N x p iterate through the p x p Gram matrix
  // (Y = U S V^T  =>  U_d S_d = Y V_d): runs once per solve, on the host.
  const int d = problem.dim(), n = problem.numPoses(), r = problem.numRangeMeasurements();
  checkMatrixShape("projectSolution", problem.getExpectedVariableSize(), Y.cols(), Y.rows(), Y.cols());
  const Index p = Y.cols();
  Vector ev;
  Matrix V;
  symmetricEigen(Y.transpose() * Y, ev, V);  // ascending
  Matrix Vd(p, d);
  for (int k = 0; k < d; ++k)
    for (Index i = 0; i < p; ++i) Vd(i, k) = V(i, p - 1 - k);  // d largest singular directions
  if (verbose) {
    printIfVerbose(verbose, "Singular values of Y: ");
    for (Index k = p - 1; k >= 0; --k) printIfVerbose(verbose, std::to_string(std::sqrt(std::max(ev(k), 0.0))));
  }
  Matrix Yd = Y * Vd;
  size_t ng0 = 0;
  {
    const unsigned nth = n < 20000 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<size_t> cnt(nth, 0);
    auto part = [&](unsigned t) {
      for (int i = static_cast<int>(static_cast<int64_t>(n) * t / nth); i < static_cast<int>(static_cast<int64_t>(n) * (t + 1) / nth); ++i)
        if (determinant(Yd.block(static_cast<Index>(i) * d, 0, d, d)) > 0) ++cnt[t];
    };
    cora::parallel_parts(nth, part);
    for (size_t c : cnt) ng0 += c;
  }
  printIfVerbose(verbose, "Out of " + std::to_string(n) + " blocks, " + std::to_string(ng0) +
                              " have positive determinant.");
  if (n > 0 && ng0 < static_cast<size_t>(n) / 2) {
    for (Index i = 0; i < Yd.rows(); ++i) Yd(i, d - 1) = -Yd(i, d - 1);  // Yd * diag(1,..,1,-1)
  }
  {
    // every pose block on its own: shared out over threads (the same numbers, whoever computes them)
    const unsigned nth = n < 20000 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    auto part = [&](unsigned t) {
      for (int i = static_cast<int>(static_cast<int64_t>(n) * t / nth); i < static_cast<int>(static_cast<int64_t>(n) * (t + 1) / nth); ++i)
        Yd.setBlock(static_cast<Index>(i) * d, 0, projectToSOd(Yd.block(static_cast<Index>(i) * d, 0, d, d)));
    };
    cora::parallel_parts(nth, part);
  }
  const Index rot = problem.numPosesDim();
  for (Index j = 0; j < r; ++j) {
    Scalar s = 0;
    for (int c = 0; c < d; ++c) s += Yd(rot + j, c) * Yd(rot + j, c);
    s = std::sqrt(s);
    if (s > 0)
      for (int c = 0; c < d; ++c) Yd(rot + j, c) /= s;
  }
  problem.checkVariablesAreValid(Yd);
  return Yd;
}

}  // namespace CORA
