#include "pyfg_text_parser.h"

#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

namespace CORA {

namespace {

enum PyFGType {
  POSE_TYPE_2D, POSE_TYPE_3D, POSE_PRIOR_2D, POSE_PRIOR_3D, LANDMARK_TYPE_2D, LANDMARK_TYPE_3D,
  LANDMARK_PRIOR_2D, LANDMARK_PRIOR_3D, REL_POSE_POSE_TYPE_2D, REL_POSE_POSE_TYPE_3D,
  REL_POSE_LANDMARK_TYPE_2D, REL_POSE_LANDMARK_TYPE_3D, RANGE_MEASURE_TYPE,
};

// record names: src/pyfg_text_parser.cpp:122-135
const std::map<std::string, PyFGType> &typeTable() {
  static const std::map<std::string, PyFGType> t{
      {"VERTEX_SE2", POSE_TYPE_2D},          {"VERTEX_SE3:QUAT", POSE_TYPE_3D},
      {"VERTEX_SE2:PRIOR", POSE_PRIOR_2D},   {"VERTEX_SE3:QUAT:PRIOR", POSE_PRIOR_3D},
      {"VERTEX_XY", LANDMARK_TYPE_2D},       {"VERTEX_XYZ", LANDMARK_TYPE_3D},
      {"VERTEX_XY:PRIOR", LANDMARK_PRIOR_2D}, {"VERTEX_XYZ:PRIOR", LANDMARK_PRIOR_3D},
      {"EDGE_SE2", REL_POSE_POSE_TYPE_2D},   {"EDGE_SE3:QUAT", REL_POSE_POSE_TYPE_3D},
      {"EDGE_SE2_XY", REL_POSE_LANDMARK_TYPE_2D}, {"EDGE_SE3_XYZ", REL_POSE_LANDMARK_TYPE_3D},
      {"EDGE_RANGE", RANGE_MEASURE_TYPE}};
  return t;
}

Scalar readScalar(std::istringstream &iss) {
  Scalar v;
  if (iss >> v) return v;
  throw std::runtime_error("Could not read scalar");
}

Vector readVector(std::istringstream &iss, int dim) {
  Vector v(dim, 1);
  for (int i = 0; i < dim; ++i)
    if (!(iss >> v(i))) throw std::runtime_error("Could not read vector");
  return v;
}

Matrix fromAngle(double a) {  // :323-328
  Matrix R(2, 2);
  R(0, 0) = std::cos(a); R(0, 1) = -std::sin(a);
  R(1, 0) = std::sin(a); R(1, 1) = std::cos(a);
  return R;
}

// Eigen::Quaterniond(w,x,y,z).toRotationMatrix() -- no normalisation (:330-338)
Matrix fromQuat(double qx, double qy, double qz, double qw) {
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  Matrix R(3, 3);
  R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
  R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
  return R;
}

Matrix readQuat(std::istringstream &iss) {  // xyzw order, :366-376
  double q[4];
  for (int i = 0; i < 4; ++i)
    if (!(iss >> q[i])) throw std::runtime_error("Could not read quaternion");
  return fromQuat(q[0], q[1], q[2], q[3]);
}

Matrix readSymmetric(std::istringstream &iss, int dim) {  // row-wise upper triangle, :385-401
  Matrix cov(dim, dim);
  double v;
  for (int i = 0; i < dim; ++i)
    for (int j = i; j < dim; ++j) {
      if (!(iss >> v)) throw std::runtime_error("Could not read covariance matrix");
      cov(i, j) = v;
      cov(j, i) = v;
    }
  return cov;
}

PyFGType typeOf(const std::string &line, std::istringstream &iss) {
  std::string item;
  if (!(iss >> item)) throw std::runtime_error("Could not read item type from line " + line);
  auto it = typeTable().find(item);
  if (it == typeTable().end()) throw std::runtime_error("Unknown item type " + item);
  return it->second;
}

}  // namespace

int getDimFromPyfgFirstLine(const std::string &filename) {  // :41-97
  std::ifstream in(filename);
  if (!in.good()) throw std::runtime_error("Could not open file " + filename);
  std::string line;
  std::getline(in, line);
  std::istringstream iss(line);
  switch (typeOf(line, iss)) {
    case POSE_TYPE_2D: case LANDMARK_TYPE_2D: return 2;
    case POSE_TYPE_3D: case LANDMARK_TYPE_3D: return 3;
    default: throw std::runtime_error("Could not determine dimension from first line " + line);
  }
}

Problem parsePyfgTextToProblem(const std::string &filename) {  // :112-321
  const int dim = getDimFromPyfgFirstLine(filename);
  Problem problem(dim, dim, Formulation::Explicit, Preconditioner::RegularizedCholesky);
  std::ifstream in(filename);
  if (!in.good()) throw std::runtime_error("Could not open file " + filename);
  std::string line, s1, s2;
  double ts;
  while (std::getline(in, line)) {
    std::istringstream iss(line);
    switch (typeOf(line, iss)) {
      case POSE_TYPE_2D:
      case POSE_TYPE_3D:  // ground-truth values are ignored (:162-175)
        if (!(iss >> ts >> s1)) throw std::runtime_error("Could not read pose variable from line " + line);
        problem.addPoseVariable(Symbol(s1));
        break;
      case POSE_PRIOR_2D: {
        if (!(iss >> ts >> s1)) throw std::runtime_error("Could not read pose prior from line " + line);
        Vector t = readVector(iss, 2);
        Matrix R = fromAngle(readScalar(iss));
        Matrix cov = readSymmetric(iss, 3);
        problem.addPosePrior(PosePrior(Symbol(s1), R, t, cov));
        break;
      }
      case POSE_PRIOR_3D: {
        if (!(iss >> ts >> s1)) throw std::runtime_error("Could not read pose prior from line " + line);
        Vector t = readVector(iss, 3);
        Matrix R = readQuat(iss);
        Matrix cov = readSymmetric(iss, 6);
        problem.addPosePrior(PosePrior(Symbol(s1), R, t, cov));
        break;
      }
      case LANDMARK_TYPE_2D:
      case LANDMARK_TYPE_3D:  // the only record without a timestamp (:203-214)
        if (!(iss >> s1)) throw std::runtime_error("Could not read landmark variable from line " + line);
        problem.addLandmarkVariable(Symbol(s1));
        break;
      case LANDMARK_PRIOR_2D:
      case LANDMARK_PRIOR_3D: {
        if (!(iss >> ts >> s1)) throw std::runtime_error("Could not read landmark prior from line " + line);
        Vector p = readVector(iss, dim);
        Matrix cov = readSymmetric(iss, dim);
        problem.addLandmarkPrior(LandmarkPrior(Symbol(s1), p, cov));
        break;
      }
      case REL_POSE_POSE_TYPE_2D: {
        if (!(iss >> ts >> s1 >> s2))
          throw std::runtime_error("Could not read relative pose measurement from line " + line);
        Vector t = readVector(iss, 2);
        Matrix R = fromAngle(readScalar(iss));
        Matrix cov = readSymmetric(iss, 3);
        problem.addRelativePoseMeasurement(RelativePoseMeasurement(Symbol(s1), Symbol(s2), R, t, cov));
        break;
      }
      case REL_POSE_POSE_TYPE_3D: {
        if (!(iss >> ts >> s1 >> s2))
          throw std::runtime_error("Could not read relative pose measurement from line " + line);
        Vector t = readVector(iss, 3);
        Matrix R = readQuat(iss);
        Matrix cov = readSymmetric(iss, 6);
        problem.addRelativePoseMeasurement(RelativePoseMeasurement(Symbol(s1), Symbol(s2), R, t, cov));
        break;
      }
      case REL_POSE_LANDMARK_TYPE_2D:
      case REL_POSE_LANDMARK_TYPE_3D: {
        if (!(iss >> ts >> s1 >> s2))
          throw std::runtime_error("Could not read relative pose-landmark measurement from line " + line);
        Vector t = readVector(iss, dim);
        Matrix cov = readSymmetric(iss, dim);
        problem.addRelativePoseLandmarkMeasurement(RelativePoseLandmarkMeasurement(Symbol(s1), Symbol(s2), t, cov));
        break;
      }
      case RANGE_MEASURE_TYPE: {
        if (!(iss >> ts >> s1 >> s2)) throw std::runtime_error("Could not read range measurement from line " + line);
        const Scalar range = readScalar(iss);
        const Scalar cov = readScalar(iss);
        problem.addRangeMeasurement(RangeMeasurement(Symbol(s1), Symbol(s2), range, cov));
        break;
      }
    }
  }
  return problem;
}

}  // namespace CORA
