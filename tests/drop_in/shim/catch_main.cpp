// main() of a reference test translation unit compiled against the catch2 stand-in (catch2/catch_test_macros.hpp):
// runs every TEST_CASE, prints one line per case, returns the number of failed cases (0 = all passed).
#include <catch2/catch_test_macros.hpp>

int main() {
  int failed_cases = 0;
  for (const CatchShim::TestCase &t : CatchShim::registry()) {
    const int before = CatchShim::counters().failed;
    bool threw = false;
    try {
      t.body();
    } catch (const CatchShim::RequireFailed &) {
    } catch (const std::exception &e) {
      threw = true;
      std::cerr << "  unexpected exception: " << e.what() << std::endl;
    }
    const bool ok = !threw && CatchShim::counters().failed == before;
    std::cout << (ok ? "PASSED " : "FAILED ") << t.name << " " << t.tags << std::endl;
    failed_cases += ok ? 0 : 1;
  }
  std::cout << CatchShim::registry().size() << " test cases, " << failed_cases << " failed, " << CatchShim::counters().checks
            << " assertions" << std::endl;
  return failed_cases;
}
