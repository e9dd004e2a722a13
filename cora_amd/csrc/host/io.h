// Solution accessors and trajectory writers (reference src/CORA_utils.cpp:204-350:
// getTranslation / getRotation / saveSolnToG20 / saveSolnToTum).
#pragma once

#include <string>
#include <vector>

#include "CORA_problem.h"

namespace CORA {

/** Row of the translation of `sym` in a rank-d solution of the explicit problem (:204-209). */
Matrix getTranslation(const Symbol &sym, const Problem &problem, const Matrix &soln);
/** d x d rotation of pose `sym` (the transpose of its block of rows); throws if it is not a rotation to
 * 1e-6 (:211-233). */
Matrix getRotation(const Symbol &sym, const Problem &problem, const Matrix &soln);

/** One `VERTEX_SE3:QUAT t x y z qx qy qz qw` / `VERTEX_SE2 t x y theta` line per symbol, t = its position in
 * `pose_symbols` (:235-292). */
void saveSolnToG20(const std::vector<Symbol> &pose_symbols, const Problem &problem, const Matrix &soln,
                   const std::string &fpath);
/** One `t x y z qx qy qz qw` line per symbol; 2-D poses are lifted to the z = 0 plane (:294-346). */
void saveSolnToTum(const std::vector<Symbol> &pose_symbols, const Problem &problem, const Matrix &soln,
                   const std::string &fpath);

/** Convenience: every pose of the problem in index order. */
void saveSolnToG20(const Problem &problem, const Matrix &soln, const std::string &fpath);
void saveSolnToTum(const Problem &problem, const Matrix &soln, const std::string &fpath);

}  // namespace CORA
