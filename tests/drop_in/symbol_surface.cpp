// Written for this repository (test infrastructure): exercises the part of the reference's include/CORA/Symbol.h:33-89 that
// round 5's header lacked -- the conversions to Key and std::string, the comparisons with a Key, symIndex / symChar /
// symbol and the one-letter factories of `namespace shorthand` -- and the Eigen-style surface of CORA_types.h that the
// reference's tests/test.cpp needs.  Header-only for Symbol, so it runs without a GPU.  Returns the number of failures.
#include <CORA/CORA_types.h>
#include <CORA/Symbol.h>

#include <iostream>
#include <map>
#include <string>

static int failures = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      ++failures;                                                           \
      std::cerr << __FILE__ << ":" << __LINE__ << ": " #cond << std::endl;  \
    }                                                                       \
  } while (0)

int main() {
  using namespace CORA::shorthand;  // NOLINT
  const CORA::Symbol s('x', 7);
  const CORA::Key k = s;                 // operator Key
  const std::string name = s;            // operator std::string
  EXPECT(name == "x7" && s.string() == "x7");
  EXPECT(k == X(7) && k == s.key());
  EXPECT(s == k && !(s != k));           // Symbol vs Key
  EXPECT(CORA::Symbol('a', 1) < A(2) && !(CORA::Symbol('b', 0) < A(5)));
  EXPECT(CORA::symIndex(k) == 7 && CORA::symChar(k) == 'x' && CORA::symbol('x', 7) == k);
  EXPECT(CORA::Symbol(L(12)).chr() == 'l' && CORA::Symbol(L(12)).index() == 12);
  EXPECT(CORA::Symbol(Z((1ull << 56) - 1)).index() == (1ull << 56) - 1);   // 56 bits of index
  EXPECT(CORA::Symbol(std::string("A123")) == CORA::Symbol('A', 123));
  std::map<CORA::Symbol, int> m;         // ordered container, as Problem keeps its variables
  m[CORA::Symbol(B(2))] = 1;
  m[CORA::Symbol(A(9))] = 2;
  EXPECT(m.begin()->second == 2);

  // Eigen-style surface of CORA_types.h (tests/test.cpp)
  CORA::SparseMatrix S(3, 3);
  S.insert(0, 0) = 2.0;
  S.insert(2, 1) = -1.0;
  S.insert(0, 2) = 0.5;
  S.coeffRef(0, 0) += 1.0;
  S.coeffRef(1, 1) = 4.0;
  S.makeCompressed();
  EXPECT(S.nonZeros() == 4 && S.coeff(0, 0) == 3.0 && S.coeff(2, 1) == -1.0 && S.coeff(1, 0) == 0.0);
  bool threw = false;
  try {
    S.insert(0, 0) = 1.0;
  } catch (const std::invalid_argument &) {
    threw = true;
  }
  EXPECT(threw);
  CORA::Matrix D = S;                    // dense from sparse
  EXPECT(D(0, 2) == 0.5 && D.rows() == 3);
  CORA::Matrix M(2, 2);
  M << 4.0, 7.0, 2.0, 6.0;               // row by row
  EXPECT(M(0, 1) == 7.0 && M(1, 0) == 2.0);
  EXPECT((M * M.inverse()).isApprox(CORA::Matrix::Identity(2, 2)));
  CORA::VectorXi b(2);
  b << 3, 4;
  EXPECT(b.size() == 2 && b(1) == 4 && b.sum() == 7);
  CORA::Vector v(3);
  v << 1.0, 2.0, 3.0;
  EXPECT(v.rows() == 3 && v.cols() == 1 && v(2) == 3.0);
  EXPECT((S * S).isApprox((D * D).sparseView()));
  std::cout << (failures ? "FAILED" : "ok") << std::endl;
  return failures;
}
