// Synthetic RA-SLAM generator of SURVEY 8(d): SE(d) odometry chain + landmarks +
// pose->landmark ranges, std::mt19937_64 seeded.  Emits an in-memory Problem and,
// optionally, the same graph as a PyFG text file.
#pragma once

#include <cstdint>
#include <string>

#include "CORA_problem.h"

namespace CORA {

struct SyntheticSpec {
  int dim = 3;
  int num_poses = 10000;
  int num_landmarks = 10;
  int num_ranges = 5000;
  int num_loop_closures = 0;
  uint64_t seed = 42;
  double sigma_t = 0.05, sigma_R = 0.01, sigma_range = 0.1;
  /** When set, receives the ground truth as an N x dim point in the explicit layout
   * ([R_i^T]; unit bearings of the ranges; translations).  With all sigmas 0 it has zero cost. */
  Matrix *ground_truth = nullptr;
};

Problem makeSyntheticProblem(const SyntheticSpec &spec, Preconditioner precond = Preconditioner::Jacobi,
                             const std::string &pyfg_out = "");

}  // namespace CORA
