// MatrixManifold / StiefelProduct / ObliqueManifold with the reference's interfaces
// (include/CORA/MatrixManifold.h:20-66, StiefelProduct.h:20-95, ObliqueManifold.h:20-75): points are stored the
// reference's way, one frame / unit vector per COLUMN block (p x kn, r x n).  The geometry runs on the GPU: every
// object owns a device handle for a measurement-free problem of its shape and calls the same kernels as
// CORA::Problem (k_project_manifold, k_tangent_project) through the C ABI -- there is no host implementation
// of the projections.  Frames of k = 2 or 3 vectors in up to 24 ambient dimensions.
#pragma once

#include <cstdint>
#include <memory>
#include <random>

#include "CORA_types.h"

struct cora_ctx;

namespace CORA {

class MatrixManifold {
 protected:
  mutable std::shared_ptr<cora_ctx> ctx_;
  /** device handle for n_frames frames of k vectors and n_spheres unit vectors in p ambient dimensions */
  cora_ctx *context(int k, int n_frames, int n_spheres, int p) const;
  Matrix deviceProject(cora_ctx *c, const Matrix &At) const;                       // transposed layout in and out
  Matrix deviceTangent(cora_ctx *c, const Matrix &Yt, const Matrix &Vt) const;

 public:
  MatrixManifold() = default;
  virtual ~MatrixManifold() = default;
  virtual Matrix projectToManifold(const Matrix &A) const = 0;
  virtual Matrix projectToTangentSpace(const Matrix &Y, const Matrix &V) const = 0;
  /** trace inner product, MatrixManifold.h:49-53 */
  Scalar innerProduct(const Matrix &A, const Matrix &B) const { return A.dot(B); }
  /** projection-based retraction, MatrixManifold.h:55-61 */
  Matrix retract(const Matrix &Y, const Matrix &V) const { return projectToManifold(Y + V); }
};

class StiefelProduct : public MatrixManifold {
  size_t k_ = 0, p_ = 0, n_ = 0;

 public:
  StiefelProduct() = default;
  StiefelProduct(size_t k, size_t p, size_t n) : k_(k), p_(p), n_(n) {}
  void set_k(size_t k) { k_ = k; ctx_.reset(); }
  void set_p(size_t p) { p_ = p; }
  void set_n(size_t n) { n_ = n; ctx_.reset(); }
  void addNewFrame() { n_++; ctx_.reset(); }
  void incrementRank() { p_++; }
  void setRank(size_t p) { p_ = p; }
  size_t get_k() const { return k_; }
  size_t get_p() const { return p_; }
  size_t get_n() const { return n_; }

  /** closest point of St(k,p)^n to A (p x kn): polar factor of every p x k block (src/StiefelProduct.cpp:8-36) */
  Matrix projectToManifold(const Matrix &A) const override;
  /** A * SymBlockDiag(B^T C), src/StiefelProduct.cpp:38-55 (BT is B transposed, as in the reference) */
  Matrix SymBlockDiagProduct(const Matrix &A, const Matrix &BT, const Matrix &C) const;
  /** V - Y SymBlockDiag(Y^T V), StiefelProduct.h:79-81 */
  Matrix projectToTangentSpace(const Matrix &Y, const Matrix &V) const override;
  Matrix random_sample(const std::default_random_engine::result_type &seed =
                           std::default_random_engine::default_seed) const;
};

class ObliqueManifold : public MatrixManifold {
  size_t r_ = 0, n_ = 0;

 public:
  ObliqueManifold() = default;
  ObliqueManifold(size_t r, size_t n) : r_(r), n_(n) {}
  void set_r(size_t r) { r_ = r; }
  void set_n(size_t n) { n_ = n; ctx_.reset(); }
  void addNewSphere() { n_++; ctx_.reset(); }
  void incrementRank() { r_++; }
  void setRank(size_t r) { r_ = r; }

  /** columns normalised to unit length, src/ObliqueManifold.cpp:6-14 */
  Matrix projectToManifold(const Matrix &A) const override;
  /** V - Y diag(colsum(Y o V)), src/ObliqueManifold.cpp:16-27 */
  Matrix projectToTangentSpace(const Matrix &Y, const Matrix &V) const override;
  Matrix random_sample(const std::default_random_engine::result_type &seed =
                           std::default_random_engine::default_seed) const;
};

}  // namespace CORA
