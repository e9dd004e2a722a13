// Odometry initialisation (reference examples/paper_experiments.cpp:375-534,
// getOdomChains / getOdomInitialization): poses by composing the odometry chain of
// each robot, landmarks random, sphere variables from the initial bearing.
#pragma once

#include "CORA_problem.h"

namespace CORA {

/** N x rank initial iterate (on the manifold). `seed` drives the random landmarks and the
 * random orthogonal mixing of the columns. */
Matrix getOdomInitialization(const Problem &problem, uint64_t seed = 7);

}  // namespace CORA
