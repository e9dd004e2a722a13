// Stand-in for <catch2/matchers/catch_matchers.hpp> (see ../catch_test_macros.hpp): the base class the reference's
// matchers derive from.
#pragma once

#include <string>

namespace Catch {
namespace Matchers {

template <typename T>
struct MatcherBase {
  virtual ~MatcherBase() = default;
  virtual bool match(const T &actual) const = 0;
  virtual std::string describe() const = 0;
};

}  // namespace Matchers
}  // namespace Catch
