// Stand-in for <catch2/matchers/catch_matchers_floating_point.hpp> (see ../catch_test_macros.hpp): WithinAbs.
#pragma once

#include <cmath>
#include <sstream>
#include <string>

#include "catch_matchers.hpp"

namespace Catch {
namespace Matchers {

struct WithinAbsMatcher : MatcherBase<double> {
  double target, margin;
  WithinAbsMatcher(double t, double m) : target(t), margin(m) {}
  bool match(const double &actual) const override { return std::fabs(actual - target) <= margin; }
  std::string describe() const override {
    std::ostringstream o;
    o.precision(17);
    o << "is within " << margin << " of " << target;
    return o.str();
  }
};
inline WithinAbsMatcher WithinAbs(double target, double margin) { return WithinAbsMatcher(target, margin); }

// |actual - target| <= eps max(|actual|, |target|); Catch2's default eps: 100 machine epsilons
struct WithinRelMatcher : MatcherBase<double> {
  double target, eps;
  WithinRelMatcher(double t, double e) : target(t), eps(e) {}
  bool match(const double &actual) const override {
    return std::fabs(actual - target) <= eps * std::fmax(std::fabs(actual), std::fabs(target));
  }
  std::string describe() const override {
    std::ostringstream o;
    o.precision(17);
    o << "and " << target << " are within " << eps * 100 << "% of each other";
    return o.str();
  }
};
inline WithinRelMatcher WithinRel(double target, double eps = 100 * 2.220446049250313e-16) { return WithinRelMatcher(target, eps); }

}  // namespace Matchers
}  // namespace Catch
