// Minimal stand-in for <catch2/catch_test_macros.hpp> -- TEST INFRASTRUCTURE of the drop-in check
// (tests/test_dropin_reference_tests.py): Catch2 is not in this image, and the reference's own test translation units
// (tests/test_optimizer_helpers.cpp, tests/test_cora.cpp, tests/test_parse_pyfg.cpp) are compiled IN PLACE, unmodified,
// against include/CORA/*.h with this directory first on the include path.  Only what those files use: TEST_CASE (name,
// optional tags), CHECK / REQUIRE / REQUIRE_FALSE, CHECK_THAT / REQUIRE_THAT with a matcher object that has
// match(value) and describe().  catch_main.cpp runs every registered case and returns the number of failed ones.
#pragma once

#include <functional>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace CatchShim {

struct TestCase {
  std::string name, tags;
  std::function<void()> body;
};
inline std::vector<TestCase> &registry() {
  static std::vector<TestCase> r;
  return r;
}
struct Registrar {
  Registrar(const char *name, const char *tags, void (*fn)()) { registry().push_back({name, tags, fn}); }
};
inline bool enter_section(const char *name) {
  std::cout << "  section: " << name << std::endl;
  return true;
}
struct Counters {
  int checks = 0, failed = 0;
};
inline Counters &counters() {
  static Counters c;
  return c;
}
struct RequireFailed : std::runtime_error {
  using std::runtime_error::runtime_error;
};
inline void report(bool ok, bool fatal, const char *expr, const char *file, int line, const std::string &detail = std::string()) {
  ++counters().checks;
  if (ok) return;
  ++counters().failed;
  std::cerr << file << ":" << line << ": FAILED: " << expr << (detail.empty() ? "" : "\n  ") << detail << std::endl;
  if (fatal) throw RequireFailed(expr);
}
template <typename T, typename M>
void check_that(const T &value, const M &matcher, bool fatal, const char *expr, const char *file, int line) {
  const bool ok = matcher.match(value);
  report(ok, fatal, expr, file, line, ok ? std::string() : matcher.describe());
}

}  // namespace CatchShim

#define CATCH_SHIM_CAT2(a, b) a##b
#define CATCH_SHIM_CAT(a, b) CATCH_SHIM_CAT2(a, b)
#define CATCH_SHIM_TEST(fn, name, tags)                                  \
  static void fn();                                                      \
  static CatchShim::Registrar CATCH_SHIM_CAT(fn, _reg)(name, tags, &fn); \
  static void fn()
#define CATCH_SHIM_PICK(_1, _2, NAME, ...) NAME
#define CATCH_SHIM_TEST1(name) CATCH_SHIM_TEST(CATCH_SHIM_CAT(catch_shim_case_, __LINE__), name, "")
#define CATCH_SHIM_TEST2(name, tags) CATCH_SHIM_TEST(CATCH_SHIM_CAT(catch_shim_case_, __LINE__), name, tags)
#define TEST_CASE(...) CATCH_SHIM_PICK(__VA_ARGS__, CATCH_SHIM_TEST2, CATCH_SHIM_TEST1)(__VA_ARGS__)

// SECTION: Catch2 re-runs the test case once per leaf section; the sections of the reference's tests only READ what the case
// set up, so here every section runs once, in order, inside one execution of the case.
#define SECTION(name) if (CatchShim::enter_section(name))
#define CHECK(expr) CatchShim::report(static_cast<bool>(expr), false, #expr, __FILE__, __LINE__)
#define REQUIRE(expr) CatchShim::report(static_cast<bool>(expr), true, #expr, __FILE__, __LINE__)
#define CHECK_FALSE(expr) CatchShim::report(!static_cast<bool>(expr), false, "!(" #expr ")", __FILE__, __LINE__)
#define REQUIRE_FALSE(expr) CatchShim::report(!static_cast<bool>(expr), true, "!(" #expr ")", __FILE__, __LINE__)
// INFO: Catch2 attaches the message to the next failing assertion of the scope; here it is printed when reached (the reference's
// tests/test.cpp:121-124 only reaches its INFO lines after the REQUIRE they explain has already passed or thrown).
#define INFO(msg)                                 \
  do {                                            \
    std::ostringstream catch_shim_info;           \
    catch_shim_info << msg;                       \
    std::cout << "  info: " << catch_shim_info.str() << std::endl; \
  } while (0)
#define CHECK_THAT(value, matcher) CatchShim::check_that(value, matcher, false, #value ", " #matcher, __FILE__, __LINE__)
#define REQUIRE_THAT(value, matcher) CatchShim::check_that(value, matcher, true, #value ", " #matcher, __FILE__, __LINE__)
