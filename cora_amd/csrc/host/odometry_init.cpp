#include "odometry_init.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <random>

namespace CORA {

Matrix getOdomInitialization(const Problem &problem, uint64_t seed) {
  const int d = problem.dim();
  const Index N = problem.getDataMatrixSize(), p = static_cast<Index>(problem.getRelaxationRank());
  Matrix x0(N, p);
  std::mt19937_64 g(seed);
  std::uniform_real_distribution<double> U(-1.0, 1.0);

  // odometry chains: per robot character, the measurements (c, k) -> (c, k+1) in index order
  // (examples/paper_experiments.cpp:375-424)
  std::map<unsigned char, std::map<uint64_t, const RelativePoseMeasurement *>> chains;
  for (const RelativePoseMeasurement &m : problem.getRPMs())
    if (m.first_id.chr() == m.second_id.chr() && m.second_id.index() == m.first_id.index() + 1)
      chains[m.first_id.chr()][m.first_id.index()] = &m;

  auto setPose = [&](const Symbol &s, const Matrix &R, const Matrix &t) {
    const Index rot = problem.getRotationIdx(s) * d, tr = problem.getTranslationIdx(s);
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) x0(rot + a, b) = R(b, a);  // block = R^T (:452-455)
    for (int c = 0; c < d; ++c) x0(tr, c) = t(c);
  };
  // poses that no chain reaches keep identity rotation / zero translation
  for (const auto &kv : problem.getPoseSymbolMap()) setPose(kv.first, Matrix::Identity(d, d), Matrix(d, 1));
  bool first = true;
  for (const auto &robot : chains) {
    const auto &edges = robot.second;
    auto it = edges.begin();
    while (it != edges.end()) {
      Matrix R = Matrix::Identity(d, d), t(d, 1);
      if (!first)
        for (int c = 0; c < d; ++c) t(c) = 10.0 * U(g);  // later chains start at a random place
      first = false;
      setPose(it->second->first_id, R, t);
      uint64_t expect = it->first;
      for (; it != edges.end() && it->first == expect; ++it, ++expect) {
        const RelativePoseMeasurement &m = *it->second;
        t = t + R * m.t;  // cur_pose = cur_pose * measure (:461-464)
        R = R * m.R;
        setPose(m.second_id, R, t);
      }
    }
  }
  // landmarks: Random(1, d) * 10 (:482-489)
  for (const auto &kv : problem.getLandmarkSymbolMap()) {
    const Index tr = problem.getTranslationIdx(kv.first);
    for (int c = 0; c < d; ++c) x0(tr, c) = 10.0 * U(g);
  }
  // sphere variables: normalised difference of the endpoints (:491-513)
  const auto &ranges = problem.getRangeMeasurements();
  for (size_t k = 0; k < ranges.size(); ++k) {
    const Index row = problem.numPosesDim() + static_cast<Index>(k);
    const Index a = problem.getTranslationIdx(ranges[k].first_id), b = problem.getTranslationIdx(ranges[k].second_id);
    double nrm = 0;
    for (int c = 0; c < d; ++c) {
      x0(row, c) = x0(b, c) - x0(a, c);
      nrm += x0(row, c) * x0(row, c);
    }
    nrm = std::sqrt(nrm);
    if (nrm < 1e-5) {
      nrm = 0;
      for (int c = 0; c < d; ++c) { x0(row, c) = U(g); nrm += x0(row, c) * x0(row, c); }
      nrm = std::sqrt(nrm);
    }
    for (int c = 0; c < d; ++c) x0(row, c) /= nrm;
  }
  // random p x p rotation so that the iterate is generically dense (:515-531): Gram-Schmidt of a
  // random matrix, determinant fixed to +1
  Matrix Qm(p, p);
  for (Index j = 0; j < p; ++j) {
    for (Index i = 0; i < p; ++i) Qm(i, j) = U(g);
    for (int pass = 0; pass < 2; ++pass)
      for (Index k = 0; k < j; ++k) {
        double dot = 0;
        for (Index i = 0; i < p; ++i) dot += Qm(i, j) * Qm(i, k);
        for (Index i = 0; i < p; ++i) Qm(i, j) -= dot * Qm(i, k);
      }
    double nrm = 0;
    for (Index i = 0; i < p; ++i) nrm += Qm(i, j) * Qm(i, j);
    nrm = std::sqrt(nrm);
    for (Index i = 0; i < p; ++i) Qm(i, j) /= nrm;
  }
  // det +1 (the reference flips the last column): with p = d a reflection would start every rotation block in
  // the other component of O(d), which the solver cannot leave
  {
    Matrix A = Qm;  // sign of det by Gaussian elimination with partial pivoting
    int sign = 1;
    for (Index k = 0; k < p; ++k) {
      Index piv = k;
      for (Index i = k + 1; i < p; ++i)
        if (std::fabs(A(i, k)) > std::fabs(A(piv, k))) piv = i;
      if (piv != k) {
        for (Index j = 0; j < p; ++j) std::swap(A(k, j), A(piv, j));
        sign = -sign;
      }
      if (A(k, k) < 0) sign = -sign;
      for (Index i = k + 1; i < p; ++i) {
        const double f = A(i, k) / A(k, k);
        for (Index j = k; j < p; ++j) A(i, j) -= f * A(k, j);
      }
    }
    if (sign < 0)
      for (Index i = 0; i < p; ++i) Qm(i, p - 1) = -Qm(i, p - 1);
  }
  return x0 * Qm;
}

}  // namespace CORA
