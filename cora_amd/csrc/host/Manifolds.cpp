#include "Manifolds.h"

#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/cora_hip.h"

namespace CORA {

namespace {
[[noreturn]] void fail(cora_ctx *c, const char *where) {
  throw std::runtime_error(std::string(where) + ": " + cora_last_error(c));
}
Matrix gaussian(Index rows, Index cols, std::default_random_engine::result_type seed) {
  std::default_random_engine g(seed);
  std::normal_distribution<Scalar> n(0.0, 1.0);
  Matrix R(rows, cols);
  for (Index j = 0; j < cols; ++j)
    for (Index i = 0; i < rows; ++i) R(i, j) = n(g);
  return R;
}
}  // namespace

// A measurement-free problem of the manifold's shape: Q = 0, rows = k * n_frames rotation rows + n_spheres
// range rows (+ one unused translation row per frame: the handle's layout gives every pose one).  Only the
// geometry kernels of the handle are used.
cora_ctx *MatrixManifold::context(int k, int n_frames, int n_spheres, int p) const {
  if (k != 2 && k != 3) throw std::invalid_argument("manifold kernels are built for frames of 2 or 3 vectors");
  if (p < k || p > 24) throw std::invalid_argument("ambient dimension must be between the frame size and 24");
  if (!ctx_) {
    const int64_t N = static_cast<int64_t>(k + 1) * n_frames + n_spheres;
    if (N <= 0) throw std::invalid_argument("empty manifold");
    const std::vector<int32_t> rowptr(static_cast<size_t>(N) + 1, 0);
    cora_ctx *c = nullptr;
    if (cora_ctx_create(0, k, n_frames, n_spheres, n_frames, rowptr.data(), nullptr, nullptr, &c) != CORA_OK)
      throw std::runtime_error(std::string("manifold: cannot create the device handle: ") + cora_last_error(nullptr));
    ctx_ = std::shared_ptr<cora_ctx>(c, [](cora_ctx *q) { cora_ctx_destroy(q); });
  }
  if (cora_get_rank(ctx_.get()) != p && cora_set_rank(ctx_.get(), p) != CORA_OK) fail(ctx_.get(), "manifold rank");
  return ctx_.get();
}

namespace {
// one frame vector / unit vector per ROW, as the device stores points; the handle's unused translation rows are zero
Matrix toDeviceLayout(cora_ctx *c, const Matrix &At) {
  Matrix A = Matrix::Zero(static_cast<Index>(cora_dim(c)), At.rows());
  A.setBlock(0, 0, At.transpose());
  return A;
}
}  // namespace

Matrix MatrixManifold::deviceProject(cora_ctx *c, const Matrix &At) const {
  const Matrix A = toDeviceLayout(c, At);
  Matrix out(A.rows(), A.cols());
  if (cora_project_to_manifold(c, A.data(), static_cast<int>(A.rows()), out.data(), static_cast<int>(out.rows())) != CORA_OK)
    fail(c, "projectToManifold");
  return out.block(0, 0, At.cols(), At.rows()).transpose();
}

Matrix MatrixManifold::deviceTangent(cora_ctx *c, const Matrix &Yt, const Matrix &Vt) const {
  const Matrix Y = toDeviceLayout(c, Yt), V = toDeviceLayout(c, Vt);
  Matrix out(Y.rows(), Y.cols());
  if (cora_tangent_space_projection(c, Y.data(), static_cast<int>(Y.rows()), V.data(), static_cast<int>(V.rows()), out.data(),
                                    static_cast<int>(out.rows())) != CORA_OK)
    fail(c, "projectToTangentSpace");
  return out.block(0, 0, Yt.cols(), Yt.rows()).transpose();
}

// ---- StiefelProduct -------------------------------------------------------------------------------------------
Matrix StiefelProduct::projectToManifold(const Matrix &A) const {
  checkMatrixShape("StiefelProduct::projectToManifold", p_, k_ * n_, A.rows(), A.cols());
  return deviceProject(context(static_cast<int>(k_), static_cast<int>(n_), 0, static_cast<int>(p_)), A);
}

Matrix StiefelProduct::projectToTangentSpace(const Matrix &Y, const Matrix &V) const {
  checkMatrixShape("StiefelProduct::projectToTangentSpace::Y", p_, k_ * n_, Y.rows(), Y.cols());
  checkMatrixShape("StiefelProduct::projectToTangentSpace::V", p_, k_ * n_, V.rows(), V.cols());
  return deviceTangent(context(static_cast<int>(k_), static_cast<int>(n_), 0, static_cast<int>(p_)), Y, V);
}

Matrix StiefelProduct::SymBlockDiagProduct(const Matrix &A, const Matrix &BT, const Matrix &C) const {
  checkMatrixShape("StiefelProduct::SymBlockDiagProduct::A", p_, k_ * n_, A.rows(), A.cols());
  checkMatrixShape("StiefelProduct::SymBlockDiagProduct::BT", k_ * n_, p_, BT.rows(), BT.cols());
  checkMatrixShape("StiefelProduct::SymBlockDiagProduct::C", p_, k_ * n_, C.rows(), C.cols());
  // The tangent projection kernel computes V_i - Y_i sym(Y_i^T V_i) for ANY Y (it never assumes Y orthonormal),
  // so for B = A the product is C - projectToTangentSpace(A, C), on the device.  For B != A the blocks
  // sym(B_i^T C_i) are formed here: k x k products, the helper's only other use in the reference.
  const Matrix B = BT.transpose();
  bool same = true;
  for (Index j = 0; j < A.cols() && same; ++j)
    for (Index i = 0; i < A.rows(); ++i)
      if (A(i, j) != B(i, j)) { same = false; break; }
  if (same) return C - projectToTangentSpace(A, C);
  const Index k = static_cast<Index>(k_), p = static_cast<Index>(p_);
  Matrix R(p, k * static_cast<Index>(n_));
  for (Index f = 0; f < static_cast<Index>(n_); ++f) {
    Matrix S(k, k);
    for (Index a = 0; a < k; ++a)
      for (Index b = 0; b < k; ++b) {
        Scalar s = 0, t = 0;
        for (Index i = 0; i < p; ++i) {
          s += B(i, f * k + a) * C(i, f * k + b);
          t += B(i, f * k + b) * C(i, f * k + a);
        }
        S(a, b) = 0.5 * (s + t);
      }
    for (Index i = 0; i < p; ++i)
      for (Index b = 0; b < k; ++b) {
        Scalar s = 0;
        for (Index a = 0; a < k; ++a) s += A(i, f * k + a) * S(a, b);
        R(i, f * k + b) = s;
      }
  }
  return R;
}

Matrix StiefelProduct::random_sample(const std::default_random_engine::result_type &seed) const {
  return projectToManifold(gaussian(static_cast<Index>(p_), static_cast<Index>(k_ * n_), seed));
}

// ---- ObliqueManifold ------------------------------------------------------------------------------------------
Matrix ObliqueManifold::projectToManifold(const Matrix &A) const {
  checkMatrixShape("ObliqueManifold::projectToManifold", r_, n_, A.rows(), A.cols());
  return deviceProject(context(2, 0, static_cast<int>(n_), static_cast<int>(r_)), A);
}

Matrix ObliqueManifold::projectToTangentSpace(const Matrix &Y, const Matrix &V) const {
  checkMatrixShape("ObliqueManifold::projectToTangentSpace::Y", r_, n_, Y.rows(), Y.cols());
  checkMatrixShape("ObliqueManifold::projectToTangentSpace::V", r_, n_, V.rows(), V.cols());
  return deviceTangent(context(2, 0, static_cast<int>(n_), static_cast<int>(r_)), Y, V);
}

Matrix ObliqueManifold::random_sample(const std::default_random_engine::result_type &seed) const {
  return projectToManifold(gaussian(static_cast<Index>(r_), static_cast<Index>(n_), seed));
}

}  // namespace CORA
