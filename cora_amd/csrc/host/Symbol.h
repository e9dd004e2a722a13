// GTSAM-style symbol: one character + 56-bit index (reference include/CORA/Symbol.h:20-89,
// src/Symbol.cpp:28-40): the class with its conversions to Key and std::string and its comparisons with a Key (:33-52),
// symIndex / symChar / symbol (:55-58) and the one-letter factories of `namespace shorthand` (:60-89).
#pragma once

#include <climits>
#include <cstdint>
#include <functional>
#include <string>
#include <utility>

namespace CORA {

typedef uint64_t Key;

class Symbol {
 protected:
  unsigned char c_;
  uint64_t j_;
  static constexpr size_t kKeyBits = sizeof(Key) * 8;
  static constexpr size_t kChrBits = sizeof(unsigned char) * 8;
  static constexpr size_t kIndexBits = kKeyBits - kChrBits;
  static constexpr Key kChrMask = Key(UCHAR_MAX) << kIndexBits;
  static constexpr Key kIndexMask = ~kChrMask;

 public:
  Symbol(unsigned char c, uint64_t j) : c_(c), j_(j) {}
  explicit Symbol(const std::string &s) : c_(static_cast<unsigned char>(s.at(0))), j_(std::stoull(s.substr(1))) {}
  explicit Symbol(Key key) : c_(static_cast<unsigned char>(key >> kIndexBits)), j_(key & kIndexMask) {}
  uint64_t index() const { return j_; }
  unsigned char chr() const { return c_; }
  Key key() const { return (Key(c_) << kIndexBits) | j_; }
  std::string string() const { return std::string(1, static_cast<char>(c_)) + std::to_string(j_); }
  // conversions, implicit as in the reference (:33-40)
  operator Key() const { return key(); }          // NOLINT(runtime/explicit)
  operator std::string() const { return string(); }  // NOLINT(runtime/explicit)
  bool operator==(const Symbol &o) const { return c_ == o.c_ && j_ == o.j_; }
  bool operator!=(const Symbol &o) const { return !(*this == o); }
  bool operator<(const Symbol &o) const { return c_ < o.c_ || (c_ == o.c_ && j_ < o.j_); }
  bool operator==(const Key &o) const { return key() == o; }
  bool operator!=(const Key &o) const { return key() != o; }
  bool operator<(const Key &o) const { return key() < o; }
};

using SymbolPair = std::pair<Symbol, Symbol>;
inline uint64_t symIndex(Key key) { return Symbol(key).index(); }
inline unsigned char symChar(Key key) { return Symbol(key).chr(); }
inline Key symbol(unsigned char c, uint64_t j) { return Symbol(c, j).key(); }

/** `using namespace CORA::shorthand; Key k = X(3);` -- a lower-case letter per factory, as GTSAM's symbol_shorthand. */
namespace shorthand {
#define CORA_SYMBOL_SHORTHAND(NAME, CHR) \
  inline Key NAME(std::uint64_t j) { return Symbol(CHR, j); }
CORA_SYMBOL_SHORTHAND(A, 'a') CORA_SYMBOL_SHORTHAND(B, 'b') CORA_SYMBOL_SHORTHAND(C, 'c') CORA_SYMBOL_SHORTHAND(D, 'd')
CORA_SYMBOL_SHORTHAND(E, 'e') CORA_SYMBOL_SHORTHAND(F, 'f') CORA_SYMBOL_SHORTHAND(G, 'g') CORA_SYMBOL_SHORTHAND(H, 'h')
CORA_SYMBOL_SHORTHAND(I, 'i') CORA_SYMBOL_SHORTHAND(J, 'j') CORA_SYMBOL_SHORTHAND(K, 'k') CORA_SYMBOL_SHORTHAND(L, 'l')
CORA_SYMBOL_SHORTHAND(M, 'm') CORA_SYMBOL_SHORTHAND(N, 'n') CORA_SYMBOL_SHORTHAND(O, 'o') CORA_SYMBOL_SHORTHAND(P, 'p')
CORA_SYMBOL_SHORTHAND(Q, 'q') CORA_SYMBOL_SHORTHAND(R, 'r') CORA_SYMBOL_SHORTHAND(S, 's') CORA_SYMBOL_SHORTHAND(T, 't')
CORA_SYMBOL_SHORTHAND(U, 'u') CORA_SYMBOL_SHORTHAND(V, 'v') CORA_SYMBOL_SHORTHAND(W, 'w') CORA_SYMBOL_SHORTHAND(X, 'x')
CORA_SYMBOL_SHORTHAND(Y, 'y') CORA_SYMBOL_SHORTHAND(Z, 'z')
#undef CORA_SYMBOL_SHORTHAND
}  // namespace shorthand

}  // namespace CORA
