#include "io.h"

#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "dense.h"

namespace CORA {

namespace {
void quaternion(const Matrix &rot, int d, double q[4]) {  // x y z w of the rotation padded to 3 x 3
  double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int a = 0; a < d; ++a)
    for (int b = 0; b < d; ++b) R[a][b] = rot(a, b);
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

std::vector<Symbol> allPoses(const Problem &problem) {
  const auto map = problem.getPoseSymbolMap();
  std::vector<Symbol> syms(map.size(), Symbol('A', 0));
  for (const auto &kv : map) syms[static_cast<size_t>(kv.second)] = kv.first;
  return syms;
}
}  // namespace

Matrix getTranslation(const Symbol &sym, const Problem &problem, const Matrix &soln) {
  checkMatrixShape("getTranslation", problem.getDataMatrixSize(), problem.dim(), soln.rows(), soln.cols());
  return soln.block(problem.getTranslationIdx(sym), 0, 1, problem.dim());
}

Matrix getRotation(const Symbol &sym, const Problem &problem, const Matrix &soln) {
  checkMatrixShape("getRotation", problem.getDataMatrixSize(), problem.dim(), soln.rows(), soln.cols());
  const Index d = problem.dim(), start = problem.getRotationIdx(sym);
  const Matrix rot = soln.block(start * d, 0, d, d).transpose();
  const Scalar det = determinant(rot);
  if (std::abs(det - 1) > 1e-6)
    throw std::runtime_error("Rotation matrix determinant is: " + std::to_string(det) + " not 1");
  if ((rot * rot.transpose() - Matrix::Identity(d, d)).norm() > 1e-6)
    throw std::runtime_error("Rotation matrix is not orthogonal");
  return rot;
}

void saveSolnToG20(const std::vector<Symbol> &pose_symbols, const Problem &problem, const Matrix &soln,
                   const std::string &fpath) {
  checkMatrixShape("saveSolnToG20", problem.getDataMatrixSize(), problem.dim(), soln.rows(), soln.cols());
  FILE *f = std::fopen(fpath.c_str(), "w");
  if (!f) throw std::runtime_error("Could not open file " + fpath);
  try {
    const int d = problem.dim();
    for (size_t time = 0; time < pose_symbols.size(); ++time) {
      const Matrix tran = getTranslation(pose_symbols[time], problem, soln);
      const Matrix rot = getRotation(pose_symbols[time], problem, soln);
      const double x = tran(0, 0), y = tran(0, 1), z = d == 2 ? 0.0 : tran(0, 2);
      if (d == 3) {
        double q[4];
        quaternion(rot, d, q);
        std::fprintf(f, "VERTEX_SE3:QUAT %zu %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", time, x, y, z, q[0], q[1], q[2], q[3]);
      } else {
        std::fprintf(f, "VERTEX_SE2 %zu %.9g %.9g %.9g\n", time, x, y, std::atan2(rot(1, 0), rot(0, 0)));
      }
    }
  } catch (...) {
    std::fclose(f);
    throw;
  }
  std::fclose(f);
}

void saveSolnToTum(const std::vector<Symbol> &pose_symbols, const Problem &problem, const Matrix &soln,
                   const std::string &fpath) {
  checkMatrixShape("saveSolnToTum", problem.getDataMatrixSize(), problem.dim(), soln.rows(), soln.cols());
  FILE *f = std::fopen(fpath.c_str(), "w");
  if (!f) throw std::runtime_error("Could not open file " + fpath);
  try {
    const int d = problem.dim();
    for (size_t time = 0; time < pose_symbols.size(); ++time) {
      const Matrix tran = getTranslation(pose_symbols[time], problem, soln);
      const Matrix rot = getRotation(pose_symbols[time], problem, soln);
      double q[4];
      quaternion(rot, d, q);
      std::fprintf(f, "%zu %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", time, tran(0, 0), tran(0, 1), d == 2 ? 0.0 : tran(0, 2),
                   q[0], q[1], q[2], q[3]);
    }
  } catch (...) {
    std::fclose(f);
    throw;
  }
  std::fclose(f);
}

void saveSolnToG20(const Problem &problem, const Matrix &soln, const std::string &fpath) {
  saveSolnToG20(allPoses(problem), problem, soln, fpath);
}
void saveSolnToTum(const Problem &problem, const Matrix &soln, const std::string &fpath) {
  saveSolnToTum(allPoses(problem), problem, soln, fpath);
}

}  // namespace CORA
