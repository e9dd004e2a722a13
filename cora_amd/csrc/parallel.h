// One helper for every "share this loop out over a few threads" of the host code (format / plan builders, the sparse
// Cholesky, solution rounding): the pieces run on nth threads (piece 0 on the caller's), EVERY thread is joined before
// the call returns whatever happens, and the first exception thrown by any piece is rethrown on the caller's thread --
// so a bad_alloc in a worker, or an exception on the calling thread while workers are still running, ends in the
// caller's catch (the C ABI's guarded()) instead of std::terminate.
#pragma once

#include <exception>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace cora {

template <class Body>
void parallel_parts(unsigned nth, Body &&body) {
  if (nth <= 1) {
    body(0u);
    return;
  }
  std::exception_ptr first;
  std::mutex m;
  auto piece = [&](unsigned t) {
    try {
      body(t);
    } catch (...) {
      std::lock_guard<std::mutex> lk(m);
      if (!first) first = std::current_exception();
    }
  };
  std::vector<std::thread> pool;
  struct Join {
    std::vector<std::thread> &p;
    ~Join() {
      for (std::thread &t : p)
        if (t.joinable()) t.join();
    }
  } join{pool};
  pool.reserve(nth - 1);
  for (unsigned t = 1; t < nth; ++t) pool.emplace_back(piece, t);  // (a failed start throws: Join collects the ones that run)
  piece(0u);
  for (std::thread &t : pool) t.join();
  if (first) std::rethrow_exception(first);
}

}  // namespace cora
