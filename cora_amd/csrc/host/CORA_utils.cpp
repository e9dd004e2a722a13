#include "CORA_utils.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <thread>

#include "../../../include/cora_hip.h"
#include "dense.h"
#include "sparse_cholesky.h"
#include "../parallel.h"

namespace CORA {

namespace {

// S x for one vector, rows shared out over threads (a row's sum does not depend on who forms it)
Vector sparseTimesVector(const SparseMatrix &S, const Vector &x) {
  const Index n = S.rows();
  Vector y(n, 1);
  auto rows = [&](Index lo, Index hi) {
    for (Index i = lo; i < hi; ++i) {
      Scalar s = 0;
      for (int32_t q = S.outer[static_cast<size_t>(i)]; q < S.outer[static_cast<size_t>(i) + 1]; ++q)
        s += S.values[q] * x(S.inner[q]);
      y(i) = s;
    }
  };
  const unsigned nth = n < 50000 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (nth <= 1) {
    rows(0, n);
    return y;
  }
  cora::parallel_parts(nth, [&](unsigned t) { rows(n * static_cast<Index>(t) / nth, n * static_cast<Index>(t + 1) / nth); });
  return y;
}

}  // namespace

CertResults fast_verification(const SparseMatrix &S, Scalar eta, const Matrix &X0, size_t max_iters,
                              const std::vector<int32_t> &perm_in, cora_ctx *ctx,
                              const std::optional<DeviceOperator> &S_op,
                              const std::optional<DeviceOperator> &precond, Scalar max_fill_factor, Scalar drop_tol,
                              const FastVerificationLab *lab, SymbolicCache *symbolic) {
  if (X0.rows() != S.rows()) throw std::invalid_argument("fast_verification: the start block has the wrong number of rows");
  return fast_verification(S, eta, std::vector<HostColumns>{HostColumns{X0.data(), static_cast<int>(X0.cols())}}, max_iters,
                           perm_in, ctx, S_op, precond, max_fill_factor, drop_tol, lab, symbolic);
}

CertResults fast_verification(const SparseMatrix &S, Scalar eta, const std::vector<HostColumns> &X0, size_t max_iters,
                              const std::vector<int32_t> &perm_in, cora_ctx *ctx,
                              const std::optional<DeviceOperator> &S_op,
                              const std::optional<DeviceOperator> &precond, Scalar max_fill_factor, Scalar drop_tol,
                              const FastVerificationLab *lab, SymbolicCache *symbolic, std::shared_ptr<LOBPCGSolver> *keep_block) {
  if (keep_block) keep_block->reset();
  const Index n = S.rows();
  int x0_cols = 0;
  for (const HostColumns &h : X0) x0_cols += h.cols;
  CertResults results;
  results.theta = 0;
  results.num_iters = 0;
  // STEP 1: Cholesky of M = S + eta I  (src/CORA_utils.cpp:28-51)
  std::vector<int32_t> natural;
  if (perm_in.empty()) {
    natural.resize(static_cast<size_t>(n));
    std::iota(natural.begin(), natural.end(), 0);
  }
  const std::vector<int32_t> &perm = perm_in.empty() ? natural : perm_in;
  const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "    [verify] %-24s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  // The PSD test runs on the host and leaves the device idle; step 2 below -- the unpreconditioned eigensolver for 1 % of
  // the budget -- needs nothing from it.  With the problem's own operator at hand (no handle to build) on one GPU it is
  // started NOW on a thread of its own and joined after the factorisation: a failed certification no longer pays for it
  // (10 ms of 55 at 10^5 poses), a successful one stops it at its next iteration and drops the result.  Same numbers
  // either way.  (Partitioned handles: every rank would have to stop at the same iteration -- not speculated.)
  const double unprecon_iter_frac = .01;
  const int N = static_cast<int>(n);
  LOBPCGStop stopfun = [eta](size_t, const std::vector<Scalar> &Theta, const double *, int) {
    return (Theta[0] - eta) < -eta / 2;  // X orthonormal: x' S x = theta_M - eta
  };
  struct Speculation {
    std::thread th;
    std::atomic<bool> cancel{false};
    std::unique_ptr<LOBPCGSolver> solver;
    LOBPCGResult r;
    std::exception_ptr err;
    bool ran = false;
    ~Speculation() {
      cancel = true;
      if (th.joinable()) th.join();
    }
  } spec;
  if (S_op && ctx && n > 100 && cora_world(ctx) == 1 && std::getenv("CORA_NO_CERT_SPECULATION") == nullptr) {
    spec.ran = true;
    spec.th = std::thread([&, c = ctx] {
      try {
        const DeviceOperator Sop0 = *S_op;
        DeviceOperator Mop0 = [&, c](const double *dX, int k, double *dOut) {
          Sop0(dX, k, dOut);
          if (eta != 0.0 && cora_axpby_cols_dev(c, k, eta, dX, 1.0, dOut) != CORA_OK) throw std::runtime_error(cora_last_error(c));
        };
        LOBPCGStop stop0 = [&](size_t it, const std::vector<Scalar> &Theta, const double *X, int m) {
          return spec.cancel.load() || stopfun(it, Theta, X, m);
        };
        spec.solver = std::make_unique<LOBPCGSolver>(c, N);
        spec.r = spec.solver->run(Mop0, std::nullopt, X0, 1, static_cast<size_t>(unprecon_iter_frac * max_iters), 0.0, stop0, false);
      } catch (...) {
        spec.err = std::current_exception();
      }
    });
  }
  const CholeskyFactor F = choleskyFactor(S, static_cast<int>(n), eta, perm, symbolic);
  tick("Cholesky of S + eta I");
  const bool PSD = F.ok;
  results.is_certified = PSD;
  if (PSD) {
    results.x = Vector::Zero(n, 1);
    results.all_eigvecs = Matrix();
    return results;  // (the speculation, if any, is stopped and joined by its destructor)
  }
  if (spec.th.joinable()) spec.th.join();
  if (spec.err) std::rethrow_exception(spec.err);
  if (n <= 100) {  // dense path (:63-74)
    Matrix D(n, n);
    for (Index i = 0; i < n; ++i)
      for (int32_t q = S.outer[i]; q < S.outer[i + 1]; ++q) D(i, S.inner[q]) += S.values[q];
    Vector ev;
    Matrix V;
    symmetricEigen(D, ev, V);
    results.theta = ev(0);
    results.x = V.col(0);
    results.all_eigvecs = V;
    return results;
  }
  // STEP 2/3: LOBPCG on M with the stopping rule x' S x < -eta/2 (:83-167)
  std::shared_ptr<cora_ctx> tmp;
  DeviceOperator Sop;
  if (S_op) {
    Sop = *S_op;
  } else {
    cora_ctx *c = nullptr;  // any symmetric S is a CORA problem with 0 poses, 0 ranges, n translations
    if (cora_ctx_create(0, 3, 0, 0, static_cast<int>(n), S.outerIndexPtr(), S.innerIndexPtr(), S.valuePtr(), &c) !=
        CORA_OK)
      throw std::runtime_error(std::string("fast_verification: ") + cora_last_error(nullptr));
    tmp.reset(c, cora_ctx_destroy);
    ctx = c;
    Sop = [c](const double *dX, int k, double *dOut) {
      if (cora_spmm_dev(c, dX, k, dOut) != CORA_OK) throw std::runtime_error(cora_last_error(c));
    };
  }
  cora_ctx *c = ctx;
  DeviceOperator Mop = [&, c](const double *dX, int k, double *dOut) {
    Sop(dX, k, dOut);
    if (eta != 0.0 && cora_axpby_cols_dev(c, k, eta, dX, 1.0, dOut) != CORA_OK)
      throw std::runtime_error(cora_last_error(c));
  };
  // STEP 2: unpreconditioned LOBPCG for 1 % of the iteration budget (:104-119).  The blocks of a run stay on the device;
  // what comes back to the host is the Ritz block of the run that ended the search, once.
  std::unique_ptr<LOBPCGSolver> solver;
  LOBPCGResult r;
  if (spec.ran) {  // it ran beside the factorisation
    solver = std::move(spec.solver);
    r = std::move(spec.r);
  } else {
    solver = std::make_unique<LOBPCGSolver>(c, N);
    r = solver->run(Mop, std::nullopt, X0, 1, static_cast<size_t>(unprecon_iter_frac * max_iters), 0.0, stopfun, false);
  }
  size_t iters = r.num_iters;
  tick("LOBPCG, 1 % of the budget");
  if (!(r.Theta(0) - eta < -eta / 2)) {
    // STEP 3 (:129-167): the "hard" case -- a negative eigenvalue of small magnitude.  Preconditioner T: incomplete
    // L D L^T of M = S + eta I (max_fill_factor, drop_tol; libs/Preconditioners is absent, see incompleteLDLT) applied
    // with the positive-definite modification |D| (ILDL::solve(x, true), :152), on the device through the staged
    // triangular solves.  A caller-supplied `precond` takes its place.  The block is also seeded with the direction
    // of non-positive curvature the failed factorisation of step 1 yields for free (z' M z = d_k <= 0): Rayleigh-Ritz
    // can only improve on it.
    if (lab) lab->reached_step3 = true;
    const size_t budget3 = static_cast<size_t>((1.0 - unprecon_iter_frac) * max_iters);
    // (the seed is this build's addition to the reference's order -- same stopping rule, a different x -- and can be
    // switched off: Problem::setVerificationLab / FastVerificationLab per call, CORA_NO_PIVOT_SEED=1 for a whole process;
    // without it the search is the reference's own: bootstrap block, then the ILDL-preconditioned run, :112-167)
    const bool seeded = !F.negative_direction.empty() && (!lab || lab->seed_negative_direction) &&
                        std::getenv("CORA_NO_PIVOT_SEED") == nullptr;
    std::vector<HostColumns> X0s = X0;
    if (seeded) {
      if (x0_cols >= 24) {  // no room for one more column: the seed takes the place of the last one
        int drop = 1;
        while (drop > 0 && !X0s.empty()) {
          if (X0s.back().cols > drop) { X0s.back().cols -= drop; drop = 0; }
          else { drop -= X0s.back().cols; X0s.pop_back(); }
        }
      }
      if (X0s.size() >= 4) throw std::logic_error("fast_verification: too many pieces in the start block");
      X0s.push_back(HostColumns{F.negative_direction.data(), 1});
    }
    bool done = false;
    if (seeded && !precond) {
      // with the seed in the block the first Rayleigh-Ritz step already meets the stopping rule: a few plain
      // iterations, and the factorisation below is only paid for when they do not
      solver = std::make_unique<LOBPCGSolver>(c, N);
      r = solver->run(Mop, std::nullopt, X0s, 1, std::min<size_t>(budget3, 3), 0.0, stopfun, false);
      iters += r.num_iters;
      tick("LOBPCG, seeded");
      done = r.Theta(0) - eta < -eta / 2;
    }
    if (!done) {
      std::optional<DeviceOperator> T = precond;
      // Partitioned handle: the incomplete factor is a sequential recurrence over the whole chain and does not shard,
      // so the preconditioner becomes BLOCK JACOBI OVER THE RANKS like the Cholesky one (Problem::updatePreconditioner):
      // every rank takes the diagonal block of M on its own rows, eliminates them in the order the global `perm` gives
      // them, and applies the factor to its own rows with the same device solve plan.
      if (!T && (!lab || lab->use_ildl)) {
        CholeskyFactor I;
        if (cora_world(c) == 1) {
          I = incompleteLDLT(S, static_cast<int>(n), eta, perm, max_fill_factor, drop_tol);
        } else {
          std::vector<int32_t> map(static_cast<size_t>(n));
          if (cora_row_map(c, map.data()) != CORA_OK) throw std::runtime_error(std::string("fast_verification: ") + cora_last_error(c));
          const int64_t lo = cora_shard_begin(c), hi = lo + cora_shard_rows(c);
          std::vector<int32_t> to_local(static_cast<size_t>(n), -1), own;
          for (Index i = 0; i < n; ++i)
            if (map[static_cast<size_t>(i)] >= lo && map[static_cast<size_t>(i)] < hi) {
              to_local[static_cast<size_t>(i)] = static_cast<int32_t>(own.size());
              own.push_back(static_cast<int32_t>(i));
            }
          std::vector<Triplet> t;
          for (size_t k = 0; k < own.size(); ++k)
            for (int32_t q = S.outer[own[k]]; q < S.outer[own[k] + 1]; ++q) {
              const int32_t j = to_local[static_cast<size_t>(S.inner[q])];
              if (j >= 0) t.push_back({static_cast<Index>(k), static_cast<Index>(j), S.values[q]});
            }
          SparseMatrix B(static_cast<Index>(own.size()), static_cast<Index>(own.size()));
          B.setFromTriplets(std::move(t));
          std::vector<int32_t> lperm;
          lperm.reserve(own.size());
          for (int32_t g : perm)
            if (to_local[static_cast<size_t>(g)] >= 0) lperm.push_back(to_local[static_cast<size_t>(g)]);
          I = incompleteLDLT(B, static_cast<int>(own.size()), eta, lperm, max_fill_factor, drop_tol);
          for (int32_t &q : I.perm) q = own[static_cast<size_t>(q)];
        }
        if (cora_aux_set_cholesky(c, I.n, I.Lp.data(), I.Li.data(), I.Lx.data(), I.perm.data()) != CORA_OK)
          throw std::runtime_error(std::string("fast_verification: ") + cora_last_error(c));
        T = [c](const double *dX, int k, double *dOut) {
          if (cora_aux_solve_dev(c, dX, k, dOut) != CORA_OK) throw std::runtime_error(cora_last_error(c));
        };
      }
      solver = std::make_unique<LOBPCGSolver>(c, N);
      r = solver->run(Mop, T, X0s, 1, budget3, 0.0, stopfun, false);
      iters += r.num_iters;
    }
  }
  if (keep_block) {  // the block stays where it is (the next certification starts from it): one column over the bus
    results.x = solver->column(0);
    results.all_eigvecs = Matrix();
    *keep_block = std::move(solver);
  } else {
    results.all_eigvecs = solver->block();
    results.x = results.all_eigvecs.col(0);
  }
  tick(keep_block ? "first Ritz vector to the host" : "Ritz block to the host");
  // curvature along x, recomputed from S like the reference (:124-127)
  {
    const Vector Sx = sparseTimesVector(S, results.x);
    results.theta = results.x.dot(Sx);
  }
  tick("curvature along x");
  results.num_iters = iters;
  return results;
}

Matrix projectToSOd(const Matrix &M) {
  // polar factor via the eigen-decomposition of M^T M (d <= 3), with the determinant fix of
  // src/CORA_utils.cpp:188-202 (flip the last left singular vector when det(U) det(V) < 0)
  const Index d = M.rows();
  Vector ev;
  Matrix V;
  symmetricEigen(M.transpose() * M, ev, V);  // ascending: smallest singular value first
  // Columns of U from the largest singular value down: u_k = M v_k, orthogonalised against the columns already there
  // and normalised (dividing by sqrt(eigenvalue) alone loses the small singular directions: the eigenvalues of M^T M
  // carry an absolute error of eps * sigma_max^2).  A column that vanishes (rank-deficient block) is completed with
  // the unit vector that keeps the largest remainder -- like the full U of the reference's JacobiSVD, the result is
  // orthogonal in every case.
  Matrix U(d, d);
  Scalar smax = 0;
  for (Index k = d - 1; k >= 0; --k) {
    Vector w = M * V.col(k);
    auto orth = [&](Vector &x) {
      for (int pass = 0; pass < 2; ++pass)
        for (Index j = k + 1; j < d; ++j) {
          Scalar dot = 0;
          for (Index i = 0; i < d; ++i) dot += x(i) * U(i, j);
          for (Index i = 0; i < d; ++i) x(i) -= dot * U(i, j);
        }
      Scalar nrm = 0;
      for (Index i = 0; i < d; ++i) nrm += x(i) * x(i);
      return std::sqrt(nrm);
    };
    Scalar nrm = orth(w);
    if (k == d - 1) smax = nrm;
    if (!(nrm > 1e-10 * std::max(smax, Scalar(1e-300)))) {
      Scalar best = -1;
      Vector wb(d, 1);
      for (Index e = 0; e < d; ++e) {
        Vector x(d, 1);
        for (Index i = 0; i < d; ++i) x(i) = i == e ? 1.0 : 0.0;
        const Scalar ne = orth(x);
        if (ne > best) {
          best = ne;
          wb = x;
        }
      }
      w = wb;
      nrm = best;
    }
    for (Index i = 0; i < d; ++i) U(i, k) = w(i) / nrm;
  }
  if (determinant(U) * determinant(V) < 0)
    for (Index i = 0; i < d; ++i) U(i, 0) = -U(i, 0);  // the column of the SMALLEST singular value
  return U * V.transpose();
}

}  // namespace CORA
