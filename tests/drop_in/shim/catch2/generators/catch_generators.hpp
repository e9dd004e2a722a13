// Minimal stand-in for Catch2's generators -- TEST INFRASTRUCTURE of the drop-in check (oracle/build_ref_tests.py), written for
// this repository.  The reference uses them once (tests/test.cpp:88-89): `GENERATE(take(1, filter(pred, random(1, 100))))`, one
// odd block size per run.  Catch2 re-runs a test case once per generated value; this stand-in runs a case ONCE and hands out
// the FIRST value of the generator, so take(n > 1) is refused loudly instead of silently testing less.  random() draws from a
// fixed-seed engine (Catch2's default seed is the clock): every run of the binary tests the same sizes.
#pragma once

#include <random>
#include <stdexcept>
#include <utility>

namespace CatchShim {

inline std::mt19937 &generator_engine() {
  static std::mt19937 e(20251017u);
  return e;
}

template <typename T>
struct RandomGen {
  T lo, hi;
  using value_type = T;
  T next() {
    if constexpr (std::is_integral<T>::value) {
      return std::uniform_int_distribution<T>(lo, hi)(generator_engine());
    } else {
      return std::uniform_real_distribution<T>(lo, hi)(generator_engine());
    }
  }
};

template <typename Pred, typename Gen>
struct FilterGen {
  Pred pred;
  Gen gen;
  using value_type = typename Gen::value_type;
  value_type next() {
    for (int tries = 0; tries < 100000; ++tries) {
      value_type v = gen.next();
      if (pred(v)) return v;
    }
    throw std::runtime_error("catch2 stand-in: filter() found no value that passes");
  }
};

template <typename Gen>
struct TakeGen {
  Gen gen;
  using value_type = typename Gen::value_type;
  value_type next() { return gen.next(); }
};

template <typename T>
RandomGen<T> random(T lo, T hi) {
  return RandomGen<T>{lo, hi};
}
template <typename Pred, typename Gen>
FilterGen<Pred, Gen> filter(Pred pred, Gen gen) {
  return FilterGen<Pred, Gen>{std::move(pred), std::move(gen)};
}
template <typename Gen>
TakeGen<Gen> take(int n, Gen gen) {
  if (n != 1) throw std::logic_error("catch2 stand-in: take(n) is only implemented for n = 1 (a case runs once)");
  return TakeGen<Gen>{std::move(gen)};
}
template <typename Gen>
typename Gen::value_type first_value(Gen gen) {
  return gen.next();
}

}  // namespace CatchShim

// Inside a TEST_CASE Catch2 finds take / filter / random through `using namespace Catch::Generators` that GENERATE brings in.
#define GENERATE(...)                      \
  [&] {                                    \
    using namespace CatchShim;             \
    return first_value(__VA_ARGS__);       \
  }()
