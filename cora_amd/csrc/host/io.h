// Trajectory writers (reference src/CORA_utils.cpp:204-350: saveSolnToTum / saveSolnToG20).
#pragma once

#include <string>

#include "CORA_problem.h"

namespace CORA {

/** One line per pose, in pose-index order: `index tx ty tz qx qy qz qw` (2-D poses are lifted to
 * the z = 0 plane), from a rank-d, origin-aligned solution. */
void saveSolnToTum(const Problem &problem, const Matrix &soln, const std::string &fpath);

/** g2o VERTEX_SE2 / VERTEX_SE3:QUAT lines for the poses and VERTEX_XY / VERTEX_TRACKXYZ for landmarks. */
void saveSolnToG20(const Problem &problem, const Matrix &soln, const std::string &fpath);

}  // namespace CORA
