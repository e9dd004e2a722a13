// Riemannian truncated-Newton trust-region method with a Steihaug-Toint
// preconditioned CG inner solver.
//
// The reference calls Optimization::Riemannian::TNT from the un-vendored
// submodule libs/Optimization (src/CORA.cpp:139-140,208-209; parameters
// :95-109), whose sources are NOT in the mount.  This is a from-scratch
// implementation of the published algorithm (Rosen, Carlone, Bandeira, Leonard,
// "SE-Sync", IJRR 2019, Alg. 3-4 / tech report; Conn, Gould, Toint,
// "Trust-Region Methods", Alg. 7.5.1), specialised to CORA: iterates and all
// work vectors stay resident on the GPU and every operator of the loop
// (objective, gradient, Hessian-vector product, preconditioner + projection,
// retraction, inner products, axpys) is a `_dev` call of include/cora_hip.h.
// Library defaults the reference does not override (eta1, eta2, alpha1,
// kappa_fgr) are the standard values documented below.
#pragma once

#include <string>
#include <vector>

#include "CORA_problem.h"

namespace CORA {

enum class TNTStatus {
  Gradient,                // gradient norm below tolerance
  PreconditionedGradient,  // preconditioned gradient norm below tolerance
  RelativeDecrease,
  Stepsize,
  TrustRegion,  // trust-region radius below tolerance
  IterationLimit,
  ElapsedTime,
  UserFunction
};

struct TNTParams {  // values set by the reference: src/CORA.cpp:95-109
  Scalar Delta0 = 5;
  Scalar eta1 = 0.05;  // library default (accept step if rho > eta1)
  Scalar eta2 = 0.9;   // library default (very successful step)
  Scalar alpha1 = 0.25;  // library default (shrink factor)
  Scalar alpha2 = 3.0;
  int max_TPCG_iterations = 80;
  int max_iterations = 250;
  Scalar kappa_fgr = 0.1;  // library default
  /** Run the inner Steihaug-Toint PCG with its scalars on the device (cora_stpcg_dev: no host round
   * trip per iteration).  false: the same iteration driven from the host, one inner product at a time. */
  bool device_stpcg = true;
  Scalar theta = 0.8;
  Scalar preconditioned_gradient_tolerance = 1e-6;
  Scalar gradient_tolerance = 1e-6;
  Scalar Delta_tolerance = 1e-5;
  Scalar relative_decrease_tolerance = 1e-6;
  Scalar stepsize_tolerance = 1e-6;
  Scalar max_computation_time = 20;  // seconds
  bool verbose = false;
  bool log_iterates = false;
};

struct TNTResult {  // fields the reference reads: src/CORA.cpp:141-186, tests/test_cora.cpp:15-38
  Matrix x;
  Scalar f = 0;
  Scalar gradfx_norm = 0;
  Scalar preconditioned_gradfx_norm = 0;
  TNTStatus status = TNTStatus::IterationLimit;
  Scalar elapsed_time = 0;
  std::vector<Scalar> time;
  std::vector<Scalar> objective_values;
  std::vector<Scalar> gradient_norms;
  std::vector<Scalar> preconditioned_gradient_norms;
  std::vector<Scalar> update_step_norms;
  std::vector<Scalar> update_step_M_norms;
  std::vector<Scalar> trust_region_radius;
  std::vector<Scalar> gain_ratios;
  std::vector<int> inner_iterations;
  std::vector<Matrix> iterates;
  long hessian_vector_products = 0;
  Scalar final_trust_region_radius = 0;  // Delta after the last update (a restart from res.x continues with it)
  int accepted_steps = 0;
};

/** Minimise f(Y) = 1/2 tr(Y^T Q Y) over the problem's manifold from x0 (N x p). */
TNTResult TNT(const Problem &problem, const Matrix &x0, const TNTParams &params);

std::string toString(TNTStatus s);

}  // namespace CORA
