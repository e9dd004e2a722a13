#include "TNT.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <limits>

#include "../../../include/cora_hip.h"

namespace CORA {

std::string toString(TNTStatus s) {
  switch (s) {
    case TNTStatus::Gradient: return "gradient tolerance";
    case TNTStatus::PreconditionedGradient: return "preconditioned gradient tolerance";
    case TNTStatus::RelativeDecrease: return "relative decrease tolerance";
    case TNTStatus::Stepsize: return "stepsize tolerance";
    case TNTStatus::TrustRegion: return "trust-region radius tolerance";
    case TNTStatus::IterationLimit: return "iteration limit";
    case TNTStatus::ElapsedTime: return "elapsed time";
    default: return "user function";
  }
}

namespace {

struct Dev {
  cora_ctx *c;
  int p;
  std::vector<double *> owned;
  explicit Dev(cora_ctx *ctx, int rank) : c(ctx), p(rank) {}
  ~Dev() {
    for (double *q : owned) cora_dev_free(c, q);
  }
  void chk(int rc, const char *what) const {
    if (rc != CORA_OK) {
      const std::string msg = std::string(what) + ": " + cora_last_error(c);
      if (rc == CORA_ERR_NAN) throw std::runtime_error("NaNs in preconditioned vector (" + msg + ")");
      throw std::runtime_error(msg);
    }
  }
  double *alloc() {
    double *q = nullptr;
    chk(cora_dev_alloc(c, p, &q), "cora_dev_alloc");
    owned.push_back(q);
    return q;
  }
  double dot(const double *a, const double *b) const {
    double v;
    chk(cora_dot_dev(c, a, b, p, &v), "cora_dot_dev");
    return v;
  }
  void dots2(const double *a0, const double *b0, const double *a1, const double *b1, double out[2]) const {
    const double *A[2] = {a0, a1};
    const double *B[2] = {b0, b1};
    chk(cora_dots_dev(c, 2, A, B, out), "cora_dots_dev");
  }
  void axpby(double a, const double *x, double b, double *y) const {
    chk(cora_axpby_dev(c, a, x, b, y), "cora_axpby_dev");
  }
};

// Steihaug-Toint truncated preconditioned CG for  min <g,s> + 1/2 <s,Hs>,  ||s||_M <= Delta.
// Returns the number of Hessian-vector products; s is the update step.
// Pg, g_g, g_Pg: the preconditioned gradient and <g, g>, <g, P g> at this point (the outer loop has them).
int STPCG(Dev &D, const double *grad, const double *Pg, double g_g, double g_Pg, double Delta, const TNTParams &prm,
          double *s, double *r, double *v, double *pk, double *Hp, double &step_M_norm) {
  cora_ctx *c = D.c;
  // (partitioned handles without the library's own communication: the host-driven loop below, collective calls)
  if (prm.device_stpcg && cora_stpcg_device_ok(c)) {
    int iters = 0;
    D.chk(cora_stpcg_warm_dev(c, grad, Pg, g_g, g_Pg, Delta, prm.kappa_fgr, prm.theta, prm.max_TPCG_iterations, s, r, v,
                              pk, Hp, &iters, &step_M_norm),
          "cora_stpcg_warm_dev");
    return iters;
  }
  D.axpby(0.0, grad, 0.0, s);   // s = 0
  D.axpby(1.0, grad, 0.0, r);   // r = g
  D.chk(cora_precondition_projected_dev(c, r, v), "precon");
  D.axpby(-1.0, v, 0.0, pk);    // p = -v
  double rr_rv[2];
  D.dots2(r, r, r, v, rr_rv);
  const double r0_norm = std::sqrt(rr_rv[0]);
  double r_v = rr_rv[1];
  const double target = r0_norm * std::min(prm.kappa_fgr, std::pow(r0_norm, prm.theta));
  double sigma_M2 = 0.0, s_Mp = 0.0, p_M2 = r_v;
  int iters = 0;
  while (iters < prm.max_TPCG_iterations) {
    D.chk(cora_hvp_dev(c, pk, Hp), "cora_hvp_dev");
    ++iters;
    const double kappa = D.dot(pk, Hp);
    const double alpha = r_v / kappa;
    const double sigma_next = sigma_M2 + 2 * alpha * s_Mp + alpha * alpha * p_M2;
    if (!(kappa > 0.0) || sigma_next >= Delta * Delta) {
      // negative curvature or the step leaves the trust region: go to the boundary
      const double tau = (-s_Mp + std::sqrt(s_Mp * s_Mp + p_M2 * (Delta * Delta - sigma_M2))) / p_M2;
      D.axpby(tau, pk, 1.0, s);
      step_M_norm = Delta;
      return iters;
    }
    D.chk(cora_axpy2_dev(c, alpha, pk, s, alpha, Hp, r), "cora_axpy2_dev");  // s += alpha p, r += alpha Hp
    sigma_M2 = sigma_next;
    D.chk(cora_precondition_projected_dev(c, r, v), "precon");
    D.dots2(r, r, r, v, rr_rv);
    if (std::sqrt(rr_rv[0]) <= target) break;
    const double beta = rr_rv[1] / r_v;
    r_v = rr_rv[1];
    D.axpby(-1.0, v, beta, pk);  // p = -v + beta p
    s_Mp = beta * (s_Mp + alpha * p_M2);
    p_M2 = r_v + beta * beta * p_M2;
  }
  step_M_norm = std::sqrt(sigma_M2);
  return iters;
}

}  // namespace

TNTResult TNT(const Problem &problem, const Matrix &x0, const TNTParams &prm) {
  using clock = std::chrono::steady_clock;
  const auto t0 = clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(clock::now() - t0).count(); };
  const int p = static_cast<int>(problem.getRelaxationRank());
  checkMatrixShape("TNT::x0", problem.getExpectedVariableSize(), p, x0.rows(), x0.cols());
  cora_ctx *c = problem.context();
  problem.ensurePreconditionerReady();
  Dev D(c, p);
  double *x = D.alloc(), *xprop = D.alloc(), *s = D.alloc(), *r = D.alloc(), *v = D.alloc(), *pk = D.alloc(),
         *Hp = D.alloc(), *Pg = D.alloc();
  // device vectors carry every row of Q; an implicit-formulation iterate is [x; 0] there
  Matrix x0_tmp;
  const Matrix &x0l = problem.lifted(x0, x0_tmp);
  const int N = static_cast<int>(x0l.rows());
  D.chk(cora_upload(c, x0l.data(), N, p, x), "cora_upload");

  TNTResult res;
  auto download = [&](const double *d) {
    Matrix m(N, p);
    D.chk(cora_download(c, d, p, m.data(), N), "cora_download");
    return problem.lowered(std::move(m));
  };
  // f, grad (cached on the device by set_point), preconditioned gradient
  D.chk(cora_set_point_dev(c, x), "cora_set_point_dev");
  double f;
  D.chk(cora_point_cost(c, &f), "cora_point_cost");
  const double *grad = cora_point_rgrad_dev(c);
  double g_g = 0.0, g_Pg = 0.0;  // <g, g> and <g, P g>: where the inner solve starts
  auto gradient_norms = [&](double &gn, double &pgn) {
    D.chk(cora_precondition_projected_dev(c, grad, Pg), "precon");
    // ||P g||, the measure saddleEscape uses (src/CORA.cpp:149 there, CORA.cpp here), so that a point the
    // escape accepted (pgn > tolerance) is not stopped by the first test of the next TNT call; <g, P g> is
    // only the STPCG recurrences' business
    const double *A[3] = {grad, Pg, grad};
    const double *B[3] = {grad, Pg, Pg};
    double o[3];
    D.chk(cora_dots_dev(c, 3, A, B, o), "cora_dots_dev");
    gn = std::sqrt(o[0]);
    pgn = std::sqrt(o[1]);
    g_g = o[0];
    g_Pg = o[2];
  };
  double grad_norm, pgrad_norm;
  gradient_norms(grad_norm, pgrad_norm);
  double Delta = prm.Delta0;
  if (prm.log_iterates) res.iterates.push_back(x0);

  int it = 0;
  for (;; ++it) {
    res.time.push_back(elapsed());
    res.objective_values.push_back(f);
    res.gradient_norms.push_back(grad_norm);
    res.preconditioned_gradient_norms.push_back(pgrad_norm);
    if (prm.verbose)
      std::printf("TNT %3d  f=%.10e  |g|=%.3e  |Pg|=%.3e  Delta=%.3e\n", it, f, grad_norm, pgrad_norm, Delta);
    if (grad_norm < prm.gradient_tolerance) { res.status = TNTStatus::Gradient; break; }
    if (pgrad_norm < prm.preconditioned_gradient_tolerance) { res.status = TNTStatus::PreconditionedGradient; break; }
    if (it >= prm.max_iterations) { res.status = TNTStatus::IterationLimit; break; }
    if (elapsed() > prm.max_computation_time) { res.status = TNTStatus::ElapsedTime; break; }

    double h_M_norm = 0.0;
    const int inner = STPCG(D, grad, Pg, g_g, g_Pg, Delta, prm, s, r, v, pk, Hp, h_M_norm);
    res.hessian_vector_products += inner + 1;
    // model decrease  m(0) - m(h) = -<g,h> - 1/2 <h, H h>,  the trial point and its cost: one wait for all of it
    double o[4];
    D.chk(cora_tnt_trial_dev(c, s, Hp, xprop, o), "cora_tnt_trial_dev");
    const double dm = -o[0] - 0.5 * o[1];
    const double h_norm = std::sqrt(o[2]);
    const double f_prop = o[3];
    const double df = f - f_prop;
    const double rho = df / dm;
    const double rel_dec = df / (std::sqrt(std::numeric_limits<double>::epsilon()) + std::fabs(f));
    const bool accepted = !std::isnan(rho) && rho > prm.eta1 && df > 0;
    res.inner_iterations.push_back(inner);
    res.update_step_norms.push_back(h_norm);
    res.update_step_M_norms.push_back(h_M_norm);
    res.gain_ratios.push_back(rho);
    res.trust_region_radius.push_back(Delta);
    if (accepted) {
      ++res.accepted_steps;
      std::swap(x, xprop);
      // the trial point becomes the current one (its product Q X is kept, not formed again); f, P g and the three
      // inner products of the stopping tests and of the next inner solve arrive in one wait
      double a[4];
      D.chk(cora_tnt_accept_dev(c, x, Pg, a), "cora_tnt_accept_dev");
      f = a[0];
      grad = cora_point_rgrad_dev(c);
      grad_norm = std::sqrt(a[1]);
      pgrad_norm = std::sqrt(a[2]);
      g_g = a[1];
      g_Pg = a[3];
      if (prm.log_iterates) res.iterates.push_back(download(x));
    }
    // trust-region update
    if (std::isnan(rho) || rho < prm.eta1) Delta = prm.alpha1 * h_M_norm;
    else if (rho > prm.eta2 && h_M_norm >= 0.99 * Delta) Delta = std::max(Delta, prm.alpha2 * h_M_norm);
    if (accepted && rel_dec < prm.relative_decrease_tolerance) { res.status = TNTStatus::RelativeDecrease; ++it; break; }
    if (h_norm < prm.stepsize_tolerance) { res.status = TNTStatus::Stepsize; ++it; break; }
    if (Delta < prm.Delta_tolerance) { res.status = TNTStatus::TrustRegion; ++it; break; }
  }
  if (res.objective_values.empty() || res.objective_values.back() != f) {
    res.time.push_back(elapsed());
    res.objective_values.push_back(f);
    res.gradient_norms.push_back(grad_norm);
    res.preconditioned_gradient_norms.push_back(pgrad_norm);
  }
  res.x = download(x);
  res.f = f;
  res.gradfx_norm = grad_norm;
  res.preconditioned_gradfx_norm = pgrad_norm;
  res.elapsed_time = elapsed();
  res.final_trust_region_radius = Delta;
  return res;
}

}  // namespace CORA
