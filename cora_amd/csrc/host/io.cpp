#include "io.h"

#include <cmath>
#include <cstdio>
#include <stdexcept>

namespace CORA {

namespace {
// rotation block of pose i as a 3x3 matrix.  Rows of the solution hold R_i^T's columns:
// Y_i (d x d) = R_i^T in the reference's convention (Y = [R_1 ... R_n]^T stacked), so R_i = Y_i^T.
void poseRotation(const Problem &p, const Matrix &X, int i, double R[3][3]) {
  const int d = p.dim();
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) R[a][b] = (a == b) ? 1.0 : 0.0;
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) R[a][b] = X(static_cast<Index>(i) * d + b, a);
}
void toQuat(const double R[3][3], double q[4]) {  // x y z w
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[3] = 0.25 * s; q[0] = (R[2][1] - R[1][2]) / s; q[1] = (R[0][2] - R[2][0]) / s; q[2] = (R[1][0] - R[0][1]) / s;
  } else if (R[0][0] > R[1][1] && R[0][0] > R[2][2]) {
    const double s = std::sqrt(1.0 + R[0][0] - R[1][1] - R[2][2]) * 2;
    q[3] = (R[2][1] - R[1][2]) / s; q[0] = 0.25 * s; q[1] = (R[0][1] + R[1][0]) / s; q[2] = (R[0][2] + R[2][0]) / s;
  } else if (R[1][1] > R[2][2]) {
    const double s = std::sqrt(1.0 + R[1][1] - R[0][0] - R[2][2]) * 2;
    q[3] = (R[0][2] - R[2][0]) / s; q[0] = (R[0][1] + R[1][0]) / s; q[1] = 0.25 * s; q[2] = (R[1][2] + R[2][1]) / s;
  } else {
    const double s = std::sqrt(1.0 + R[2][2] - R[0][0] - R[1][1]) * 2;
    q[3] = (R[1][0] - R[0][1]) / s; q[0] = (R[0][2] + R[2][0]) / s; q[1] = (R[1][2] + R[2][1]) / s; q[2] = 0.25 * s;
  }
}
void check(const Problem &p, const Matrix &X) {
  if (X.rows() != p.getDataMatrixSize() || X.cols() != p.dim())
    throw std::invalid_argument("trajectory writers expect a rank-d solution of the explicit problem");
}
}  // namespace

void saveSolnToTum(const Problem &p, const Matrix &X, const std::string &fpath) {
  check(p, X);
  FILE *f = std::fopen(fpath.c_str(), "w");
  if (!f) throw std::runtime_error("Could not open " + fpath);
  const int d = p.dim();
  const Index off = p.rotAndRangeMatrixSize();
  for (int i = 0; i < p.numPoses(); ++i) {
    double R[3][3], q[4], t[3] = {0, 0, 0};
    poseRotation(p, X, i, R);
    toQuat(R, q);
    for (int c = 0; c < d; ++c) t[c] = X(off + i, c);
    std::fprintf(f, "%d %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", i, t[0], t[1], t[2], q[0], q[1], q[2], q[3]);
  }
  std::fclose(f);
}

void saveSolnToG20(const Problem &p, const Matrix &X, const std::string &fpath) {
  check(p, X);
  FILE *f = std::fopen(fpath.c_str(), "w");
  if (!f) throw std::runtime_error("Could not open " + fpath);
  const int d = p.dim();
  const Index off = p.rotAndRangeMatrixSize();
  for (int i = 0; i < p.numPoses(); ++i) {
    double R[3][3], q[4];
    poseRotation(p, X, i, R);
    if (d == 2) {
      std::fprintf(f, "VERTEX_SE2 %d %.9f %.9f %.9f\n", i, X(off + i, 0), X(off + i, 1), std::atan2(R[1][0], R[0][0]));
    } else {
      toQuat(R, q);
      std::fprintf(f, "VERTEX_SE3:QUAT %d %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", i, X(off + i, 0), X(off + i, 1),
                   X(off + i, 2), q[0], q[1], q[2], q[3]);
    }
  }
  for (int j = 0; j < p.numLandmarks(); ++j) {
    const Index row = off + p.numPoses() + j;
    if (d == 2) std::fprintf(f, "VERTEX_XY %d %.9f %.9f\n", p.numPoses() + j, X(row, 0), X(row, 1));
    else std::fprintf(f, "VERTEX_TRACKXYZ %d %.9f %.9f %.9f\n", p.numPoses() + j, X(row, 0), X(row, 1), X(row, 2));
  }
  std::fclose(f);
}

}  // namespace CORA
