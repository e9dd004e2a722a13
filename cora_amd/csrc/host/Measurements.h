// Measurement records and their scalar precisions
// (reference include/CORA/Measurements.h:21-205).
#pragma once

#include <stdexcept>
#include <utility>

#include "CORA_types.h"
#include "Symbol.h"

namespace CORA {

struct Measurement {
  Symbol id;
  explicit Measurement(const Symbol &id) : id(id) {}
  bool operator==(const Measurement &o) const { return id == o.id; }
};

struct PairMeasurement {
  Symbol first_id;
  Symbol second_id;
  PairMeasurement(const Symbol &a, const Symbol &b) : first_id(a), second_id(b) {}
  SymbolPair getSymbolPair() const { return std::make_pair(first_id, second_id); }
  // unordered pair equality, Measurements.h:38-46
  bool hasSymbolPair(const SymbolPair &p) const {
    return (first_id == p.first && second_id == p.second) || (first_id == p.second && second_id == p.first);
  }
  bool operator==(const PairMeasurement &o) const { return hasSymbolPair(o.getSymbolPair()); }
};

namespace detail {
// Measurements.h:79-93
inline Scalar rotPrecision(const Matrix &cov, const char *who) {
  if (cov.rows() == 6) return 1.5 / (cov(3, 3) + cov(4, 4) + cov(5, 5));
  if (cov.rows() == 3) return 1.0 / cov(2, 2);
  throw std::runtime_error(std::string(who) + "::getRotPrecision() only implemented for 2D and 3D rotations");
}
// Measurements.h:109-112
inline Scalar transPrecision(const Matrix &cov, Index dim) {
  Scalar tr = 0;
  for (Index i = 0; i < dim; ++i) tr += cov(i, i);
  return static_cast<double>(dim) / tr;
}
}  // namespace detail

struct RelativePoseMeasurement : PairMeasurement {
  Matrix R;
  Vector t;
  Matrix cov;  // order: translation, rotation
  RelativePoseMeasurement(const Symbol &a, const Symbol &b, Matrix R_, Vector t_, Matrix cov_)
      : PairMeasurement(a, b), R(std::move(R_)), t(std::move(t_)), cov(std::move(cov_)) {}
  Scalar getRotPrecision() const { return detail::rotPrecision(cov, "RelativePoseMeasurement"); }
  Scalar getTransPrecision() const { return detail::transPrecision(cov, t.size()); }
};

struct RelativePoseLandmarkMeasurement : PairMeasurement {
  Vector t;
  Matrix cov;
  RelativePoseLandmarkMeasurement(const Symbol &a, const Symbol &b, Vector t_, Matrix cov_)
      : PairMeasurement(a, b), t(std::move(t_)), cov(std::move(cov_)) {}
  Scalar getTransPrecision() const { return detail::transPrecision(cov, t.size()); }
};

struct RangeMeasurement : PairMeasurement {
  Scalar r;
  Scalar cov;
  RangeMeasurement(const Symbol &a, const Symbol &b, Scalar r_, Scalar cov_)
      : PairMeasurement(a, b), r(r_), cov(cov_) {}
  Scalar getPrecision() const { return 1.0 / cov; }  // Measurements.h:151
};

struct PosePrior : Measurement {
  Matrix R;
  Vector t;
  Matrix cov;
  PosePrior(const Symbol &id, Matrix R_, Vector t_, Matrix cov_)
      : Measurement(id), R(std::move(R_)), t(std::move(t_)), cov(std::move(cov_)) {}
  Scalar getRotPrecision() const { return detail::rotPrecision(cov, "PosePrior"); }
  Scalar getTransPrecision() const { return detail::transPrecision(cov, t.size()); }
};

struct LandmarkPrior : Measurement {
  Vector p;
  Matrix cov;
  LandmarkPrior(const Symbol &id, Vector p_, Matrix cov_) : Measurement(id), p(std::move(p_)), cov(std::move(cov_)) {}
  Scalar getPrecision() const { return detail::transPrecision(cov, p.size()); }
  Scalar getTransPrecision() const { return getPrecision(); }
};

}  // namespace CORA
