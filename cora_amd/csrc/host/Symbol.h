// GTSAM-style symbol: one character + 56-bit index (reference include/CORA/Symbol.h:20-52,
// src/Symbol.cpp:28-40).
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
  bool operator==(const Symbol &o) const { return c_ == o.c_ && j_ == o.j_; }
  bool operator!=(const Symbol &o) const { return !(*this == o); }
  bool operator<(const Symbol &o) const { return c_ < o.c_ || (c_ == o.c_ && j_ < o.j_); }
};

using SymbolPair = std::pair<Symbol, Symbol>;

}  // namespace CORA
