#include "synthetic.h"

#include <cmath>
#include <cstdio>
#include <random>
#include <set>

namespace CORA {

namespace {

Matrix expSO(int d, std::mt19937_64 &g, double sigma) {
  std::normal_distribution<double> n(0.0, sigma);
  if (d == 2) {
    const double a = n(g);
    Matrix R(2, 2);
    R(0, 0) = std::cos(a); R(0, 1) = -std::sin(a);
    R(1, 0) = std::sin(a); R(1, 1) = std::cos(a);
    return R;
  }
  const double w[3] = {n(g), n(g), n(g)};
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  Matrix K(3, 3);
  K(0, 1) = -w[2]; K(0, 2) = w[1];
  K(1, 0) = w[2];  K(1, 2) = -w[0];
  K(2, 0) = -w[1]; K(2, 1) = w[0];
  Matrix R = Matrix::Identity(3, 3);
  if (th < 1e-12) return R + K;
  return R + K * (std::sin(th) / th) + (K * K) * ((1 - std::cos(th)) / (th * th));
}

// rotation matrix -> quaternion (x, y, z, w)
void toQuat(const Matrix &R, double q[4]) {
  const double tr = R(0, 0) + R(1, 1) + R(2, 2);
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[3] = 0.25 * s;
    q[0] = (R(2, 1) - R(1, 2)) / s; q[1] = (R(0, 2) - R(2, 0)) / s; q[2] = (R(1, 0) - R(0, 1)) / s;
  } else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) {
    const double s = std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2;
    q[3] = (R(2, 1) - R(1, 2)) / s; q[0] = 0.25 * s; q[1] = (R(0, 1) + R(1, 0)) / s; q[2] = (R(0, 2) + R(2, 0)) / s;
  } else if (R(1, 1) > R(2, 2)) {
    const double s = std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2;
    q[3] = (R(0, 2) - R(2, 0)) / s; q[0] = (R(0, 1) + R(1, 0)) / s; q[1] = 0.25 * s; q[2] = (R(1, 2) + R(2, 1)) / s;
  } else {
    const double s = std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2;
    q[3] = (R(1, 0) - R(0, 1)) / s; q[0] = (R(0, 2) + R(2, 0)) / s; q[1] = (R(1, 2) + R(2, 1)) / s; q[2] = 0.25 * s;
  }
}

}  // namespace

Problem makeSyntheticProblem(const SyntheticSpec &sp, Preconditioner precond, const std::string &pyfg_out) {
  const int d = sp.dim, n = sp.num_poses, l = sp.num_landmarks;
  if (d != 2 && d != 3) throw std::invalid_argument("synthetic generator: dim must be 2 or 3");
  if (n < 1) throw std::invalid_argument("synthetic generator: need at least one pose");
  if (static_cast<long long>(sp.num_ranges) > static_cast<long long>(n) * std::max(l, 0))
    throw std::invalid_argument("synthetic generator: more ranges than distinct (pose, landmark) pairs");
  std::mt19937_64 g(sp.seed);
  std::normal_distribution<double> nt(0.0, 0.1), unit_t(0.0, 1.0), unit_r(0.0, 1.0);
  auto noise_t = [&](std::mt19937_64 &e) { return sp.sigma_t * unit_t(e); };
  auto noise_r = [&](std::mt19937_64 &e) { return sp.sigma_range * unit_r(e); };
  Problem problem(d, d, Formulation::Explicit, precond);
  FILE *fp = pyfg_out.empty() ? nullptr : std::fopen(pyfg_out.c_str(), "w");
  if (!pyfg_out.empty() && !fp) throw std::runtime_error("Could not open " + pyfg_out);

  std::vector<Matrix> R(n), T(n);
  R[0] = Matrix::Identity(d, d);
  T[0] = Matrix(d, 1);
  const int rdim = d == 3 ? 3 : 1;
  Matrix cov(d + rdim, d + rdim);
  // a zero sigma means "no noise drawn"; the stated covariance then keeps the nominal value
  const SyntheticSpec nominal;
  const double cov_t = sp.sigma_t > 0 ? sp.sigma_t : nominal.sigma_t, cov_R = sp.sigma_R > 0 ? sp.sigma_R : nominal.sigma_R,
               cov_r = sp.sigma_range > 0 ? sp.sigma_range : nominal.sigma_range;
  for (int i = 0; i < d; ++i) cov(i, i) = cov_t * cov_t;
  for (int i = d; i < d + rdim; ++i) cov(i, i) = cov_R * cov_R;

  struct Edge { int i, j; Matrix Rm, tm; };
  std::vector<Edge> edges;
  edges.reserve(n);
  for (int i = 0; i + 1 < n; ++i) {
    const Matrix dR = expSO(d, g, 0.05);
    Matrix dt(d, 1);
    for (int c = 0; c < d; ++c) dt(c) = (c == 0 ? 1.0 : 0.0) + nt(g);
    R[i + 1] = R[i] * dR;
    T[i + 1] = T[i] + R[i] * dt;
    Edge e{i, i + 1, dR * expSO(d, g, sp.sigma_R), dt};
    for (int c = 0; c < d; ++c) e.tm(c) += noise_t(g);
    edges.push_back(std::move(e));
  }
  std::set<std::pair<int, int>> seen;
  std::uniform_int_distribution<int> up(0, n - 1);
  for (int k = 0; k < sp.num_loop_closures && n > 3; ++k) {
    int i = up(g), j = up(g);
    if (i > j) std::swap(i, j);
    if (j - i < 2 || !seen.insert({i, j}).second) continue;
    Edge e{i, j, R[i].transpose() * R[j] * expSO(d, g, sp.sigma_R), R[i].transpose() * (T[j] - T[i])};
    for (int c = 0; c < d; ++c) e.tm(c) += noise_t(g);
    edges.push_back(std::move(e));
  }

  for (int i = 0; i < n; ++i) {
    const Symbol s('A', static_cast<uint64_t>(i));
    problem.addPoseVariable(s);
    if (fp) {
      if (d == 2)
        std::fprintf(fp, "VERTEX_SE2 %d.0 %s %.17g %.17g %.17g\n", i, s.string().c_str(), T[i](0), T[i](1),
                     std::atan2(R[i](1, 0), R[i](0, 0)));
      else {
        double q[4];
        toQuat(R[i], q);
        std::fprintf(fp, "VERTEX_SE3:QUAT %d.0 %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", i,
                     s.string().c_str(), T[i](0), T[i](1), T[i](2), q[0], q[1], q[2], q[3]);
      }
    }
  }
  // landmarks uniform in the trajectory's bounding box +- 20 m
  std::vector<Matrix> L(l, Matrix(d, 1));
  {
    std::vector<double> lo(d, 1e300), hi(d, -1e300);
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < d; ++c) { lo[c] = std::min(lo[c], T[i](c)); hi[c] = std::max(hi[c], T[i](c)); }
    for (int k = 0; k < l; ++k) {
      for (int c = 0; c < d; ++c) L[k](c) = std::uniform_real_distribution<double>(lo[c] - 20, hi[c] + 20)(g);
      const Symbol s('L', static_cast<uint64_t>(k));
      problem.addLandmarkVariable(s);
      if (fp) {
        if (d == 2) std::fprintf(fp, "VERTEX_XY %s %.17g %.17g\n", s.string().c_str(), L[k](0), L[k](1));
        else std::fprintf(fp, "VERTEX_XYZ %s %.17g %.17g %.17g\n", s.string().c_str(), L[k](0), L[k](1), L[k](2));
      }
    }
  }
  for (const Edge &e : edges) {
    const Symbol a('A', static_cast<uint64_t>(e.i)), b('A', static_cast<uint64_t>(e.j));
    problem.addRelativePoseMeasurement(RelativePoseMeasurement(a, b, e.Rm, e.tm, cov));
    if (fp) {
      if (d == 2) {
        std::fprintf(fp, "EDGE_SE2 %d.0 %s %s %.17g %.17g %.17g", e.j, a.string().c_str(), b.string().c_str(),
                     e.tm(0), e.tm(1), std::atan2(e.Rm(1, 0), e.Rm(0, 0)));
      } else {
        double q[4];
        toQuat(e.Rm, q);
        std::fprintf(fp, "EDGE_SE3:QUAT %d.0 %s %s %.17g %.17g %.17g %.17g %.17g %.17g %.17g", e.j,
                     a.string().c_str(), b.string().c_str(), e.tm(0), e.tm(1), e.tm(2), q[0], q[1], q[2], q[3]);
      }
      const int cd = d + rdim;
      for (int i = 0; i < cd; ++i)
        for (int j = i; j < cd; ++j) std::fprintf(fp, " %.17g", cov(i, j));
      std::fprintf(fp, "\n");
    }
  }
  Matrix *gt = sp.ground_truth;
  const Index tb = static_cast<Index>(d) * n + sp.num_ranges;
  if (gt) {
    *gt = Matrix(static_cast<Index>(d + 1) * n + l + sp.num_ranges, d);
    for (int i = 0; i < n; ++i) {
      gt->setBlock(static_cast<Index>(i) * d, 0, R[i].transpose());
      for (int c = 0; c < d; ++c) (*gt)(tb + i, c) = T[i](c);
    }
    for (int k = 0; k < l; ++k)
      for (int c = 0; c < d; ++c) (*gt)(tb + n + k, c) = L[k](c);
  }
  std::set<std::pair<int, int>> used;
  std::uniform_int_distribution<int> ul(0, std::max(l - 1, 0));
  const double rcov = cov_r * cov_r;
  while (static_cast<int>(used.size()) < sp.num_ranges) {
    const int i = up(g), k = ul(g);
    if (!used.insert({i, k}).second) continue;
    double dist = 0;
    for (int c = 0; c < d; ++c) dist += (T[i](c) - L[k](c)) * (T[i](c) - L[k](c));
    if (gt) {
      const Index row = static_cast<Index>(d) * n + static_cast<Index>(used.size()) - 1;
      for (int c = 0; c < d; ++c) (*gt)(row, c) = (T[i](c) - L[k](c)) / std::sqrt(dist);  // Arange has +1 at the pose
    }
    dist = std::abs(std::sqrt(dist) + noise_r(g));
    const Symbol a('A', static_cast<uint64_t>(i)), b('L', static_cast<uint64_t>(k));
    problem.addRangeMeasurement(RangeMeasurement(a, b, dist, rcov));
    if (fp) std::fprintf(fp, "EDGE_RANGE %d.0 %s %s %.17g %.17g\n", i, a.string().c_str(), b.string().c_str(), dist, rcov);
  }
  if (fp) std::fclose(fp);
  return problem;
}

}  // namespace CORA
