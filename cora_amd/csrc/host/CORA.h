// solveCORA / saddleEscape / projectSolution (reference include/CORA/CORA.h:19-37,
// src/CORA.cpp:26-441): the Riemannian staircase driver.  Host control flow over the
// device-resident TNT and certification.
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "CORA_problem.h"
#include "CORA_types.h"
#include "CORA_utils.h"
#include "TNT.h"
#include "pyfg_text_parser.h"

namespace CORA {

using CoraTntResult = TNTResult;
using CoraResult = std::pair<CoraTntResult, std::vector<Matrix>>;

struct CoraSolveInfo {  // extra observability (the reference only prints these)
  bool certified = false;            // the returned (rounded and refined) solution's certificate
  bool relaxation_certified = false; // the staircase stopped because its last level was certified (not at the rank cap)
  int relaxation_rank = 0;           // the rank of that level
  Scalar eta = 0, theta = 0;
  int final_rank = 0;
  int staircase_levels = 0;
  long hessian_vector_products = 0;
  double tnt_seconds = 0, certify_seconds = 0, escape_seconds = 0;  // wall-clock split of the staircase
};

CoraResult solveCORA(Problem &problem, const Matrix &x0, int max_relaxation_rank = 20, bool verbose = false,
                     bool log_iterates = false, bool show_iterates = false, CoraSolveInfo *info = nullptr,
                     const TNTParams *params_override = nullptr);

Matrix saddleEscape(const Problem &problem, const Matrix &Y, Scalar theta, const Vector &v,
                    Scalar gradient_tolerance, Scalar preconditioned_gradient_tolerance);

Matrix projectSolution(const Problem &problem, const Matrix &Y, bool verbose = false);

}  // namespace CORA
