#include "sparse_cholesky.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

#include <algorithm>
#include <cmath>
#include <functional>
#include <numeric>
#include <stdexcept>
#include "../parallel.h"

namespace CORA {

std::vector<int32_t> coraOrdering(int d, int n, int r, int nt, const SparseMatrix &Q, int m, int leaf_poses) {
  const int64_t dn = static_cast<int64_t>(d) * n, tb = dn + r, N = dn + r + nt;
  if (m != N && m != N - 1) throw std::invalid_argument("coraOrdering: m must be N or N-1");
  // range row k hangs off the first pose translation it touches (flat arrays, a counting sort by pose: a vector per pose
  // was 2 x 10^5 allocations at 10^5 poses, and this function is on the set-up's critical path)
  std::vector<int32_t> range_pose(static_cast<size_t>(r), -1), pose_cnt(static_cast<size_t>(n) + 1, 0);
  std::vector<int32_t> loose;
  for (int k = 0; k < r; ++k) {
    int pose = -1;
    for (int32_t q = Q.outer[dn + k]; q < Q.outer[dn + k + 1]; ++q) {
      const int64_t c = Q.inner[q];
      if (c >= tb && c - tb < n) { pose = static_cast<int>(c - tb); break; }
    }
    range_pose[static_cast<size_t>(k)] = pose;
    if (pose >= 0) pose_cnt[static_cast<size_t>(pose) + 1]++;
    else loose.push_back(static_cast<int32_t>(dn + k));
  }
  std::vector<int32_t> perm;
  perm.reserve(static_cast<size_t>(N));
  // Range rows first: a range row touches only the two translations it connects, so it is a leaf of
  // the elimination tree and eliminating it adds no fill (the t-t coupling is already in Q33).  With
  // the ranges out of the way every pose is a clean chain [rotation rows, translation] and
  // consecutive poses of a leaf merge into one supernode of the device triangular solves.
  {
    for (int i = 0; i < n; ++i) pose_cnt[static_cast<size_t>(i) + 1] += pose_cnt[static_cast<size_t>(i)];
    const size_t base = perm.size(), attached = static_cast<size_t>(pose_cnt[static_cast<size_t>(n)]);
    perm.resize(base + attached);
    std::vector<int32_t> at(pose_cnt.begin(), pose_cnt.end() - 1);
    for (int k = 0; k < r; ++k)  // (ascending k within a pose, as before)
      if (range_pose[static_cast<size_t>(k)] >= 0) perm[base + static_cast<size_t>(at[static_cast<size_t>(range_pose[static_cast<size_t>(k)])]++)] = static_cast<int32_t>(dn + k);
  }
  for (int32_t row : loose) perm.push_back(row);
  auto emit_pose = [&](int i) {
    for (int a = 0; a < d; ++a) perm.push_back(static_cast<int32_t>(static_cast<int64_t>(i) * d + a));
    perm.push_back(static_cast<int32_t>(tb + i));
  };
  // The pose graph: poses i ~ j when their translations are coupled in Q33 (an odometry edge, a loop closure, a range
  // between two poses).  On a single chain in index order -- every synthetic graph, plaza, single_drone -- nested
  // dissection is a bisection of the index range with one pose as separator.  Anything else (several robots whose
  // chains follow each other in the index order and are tied together by inter-robot ranges: tiers, MR.CLAM) gets a
  // nested dissection by BFS level sets below: bisecting THEIR index range leaves every inter-robot edge across the
  // cut, and the fill (tiers: 22 s of factorisation, an explicit inverse of 134 s) is what the chain order avoids.
  std::vector<std::vector<int32_t>> adj;
  bool index_chain = true;
  for (int i = 0; i < n && index_chain; ++i)
    for (int32_t q = Q.outer[tb + i]; q < Q.outer[tb + i + 1]; ++q) {
      const int64_t c = Q.inner[q] - tb;
      if (c < 0 || c >= n || c == i) continue;
      if (c != i - 1 && c != i + 1) { index_chain = false; break; }
    }
  if (!index_chain) {  // (the adjacency lists are only walked by the level-set dissection)
    adj.resize(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i)
      for (int32_t q = Q.outer[tb + i]; q < Q.outer[tb + i + 1]; ++q) {
        const int64_t c = Q.inner[q] - tb;
        if (c < 0 || c >= n || c == i) continue;
        adj[static_cast<size_t>(i)].push_back(static_cast<int32_t>(c));
      }
  }
  if (index_chain) {
    std::function<void(int, int)> nd = [&](int lo, int hi) {  // poses [lo, hi)
      if (hi - lo <= leaf_poses) {
        for (int i = lo; i < hi; ++i) emit_pose(i);
        return;
      }
      const int mid = lo + (hi - lo) / 2;
      nd(lo, mid);
      nd(mid + 1, hi);
      emit_pose(mid);
    };
    nd(0, n);
  } else {
    // Nested dissection by level sets: BFS from a pseudo-peripheral pose of the piece, the level that splits it in
    // half is the separator (for robots that move side by side and range each other at equal times a level is one
    // pose per robot), both halves recurse, the separator is eliminated last.  Pieces of at most leaf_poses poses are
    // emitted in BFS order (chain neighbours stay next to each other).  Deterministic: ties by pose index.
    std::vector<int32_t> member(static_cast<size_t>(n), 0), level(static_cast<size_t>(n), -1);
    int32_t stamps = 0;
    // BFS over the vertices whose member[] is `stamp`, from `root`: sets level[], returns the order of visit
    auto bfs = [&](int32_t root, int32_t stamp, std::vector<int32_t> &order) {
      order.clear();
      order.push_back(root);
      level[static_cast<size_t>(root)] = 0;
      for (size_t h = 0; h < order.size(); ++h) {
        const int32_t u = order[h];
        for (int32_t v : adj[static_cast<size_t>(u)])
          if (member[static_cast<size_t>(v)] == stamp && level[static_cast<size_t>(v)] < 0) {
            level[static_cast<size_t>(v)] = level[static_cast<size_t>(u)] + 1;
            order.push_back(v);
          }
      }
    };
    std::function<void(std::vector<int32_t> &)> dissect = [&](std::vector<int32_t> &S) {
      if (S.empty()) return;
      std::sort(S.begin(), S.end());
      const int32_t stamp = ++stamps;
      for (int32_t v : S) {
        member[static_cast<size_t>(v)] = stamp;
        level[static_cast<size_t>(v)] = -1;
      }
      std::vector<int32_t> comp;
      for (int32_t seed : S) {  // connected components, in order of their smallest pose
        // (vertices handed to a recursive call carry a later stamp; emitted ones and separators keep level >= 0)
        if (member[static_cast<size_t>(seed)] != stamp || level[static_cast<size_t>(seed)] >= 0) continue;
        bfs(seed, stamp, comp);
        for (int pass = 0; pass < 2; ++pass) {  // pseudo-peripheral root: restart from the last vertex reached
          const int32_t far = comp.back();
          for (int32_t v : comp) level[static_cast<size_t>(v)] = -1;
          bfs(far, stamp, comp);
        }
        const int depth = level[static_cast<size_t>(comp.back())] + 1;
        if (static_cast<int>(comp.size()) <= leaf_poses || depth < 3) {
          for (int32_t v : comp) emit_pose(v);
          continue;
        }
        // the level at which the running count passes half of the component (never the first or the last level)
        std::vector<int64_t> count(static_cast<size_t>(depth), 0);
        for (int32_t v : comp) count[static_cast<size_t>(level[static_cast<size_t>(v)])]++;
        int cut = 1;
        int64_t run = count[0];
        while (cut < depth - 2 && 2 * (run + count[static_cast<size_t>(cut)]) < static_cast<int64_t>(comp.size())) run += count[static_cast<size_t>(cut++)];
        std::vector<int32_t> A, B, sep;
        for (int32_t v : comp) {
          const int l = level[static_cast<size_t>(v)];
          (l < cut ? A : (l > cut ? B : sep)).push_back(v);
        }
        dissect(A);
        dissect(B);
        for (int32_t v : sep) emit_pose(v);
      }
    };
    std::vector<int32_t> all(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) all[static_cast<size_t>(i)] = i;
    dissect(all);
  }
  for (int j = n; j < nt; ++j) perm.push_back(static_cast<int32_t>(tb + j));
  if (m == N - 1) {  // drop the pinned last variable (src/CORA_problem.cpp:602-609)
    std::vector<int32_t> q;
    q.reserve(perm.size());
    for (int32_t v : perm)
      if (v != N - 1) q.push_back(v);
    perm.swap(q);
  }
  if (static_cast<int64_t>(perm.size()) != m) throw std::logic_error("coraOrdering: size mismatch");
  return perm;
}


namespace {

// pattern-only part of a factorisation (see choleskyFactor)
struct Symbolic {
  uint64_t key = 0, key2 = 0;
  int n = 0;
  size_t nnzA = 0;
  std::vector<int32_t> Cp, Ci;  // upper triangle of P A P^T by columns
  std::vector<int32_t> Cmap;    // entry -> position in A.values (-1: structurally missing diagonal, value 0)
  std::vector<int32_t> Cdiag;   // position of the diagonal entry of every column
  std::vector<int32_t> parent, cnt, Lp;
  int64_t tot = 0;
};

uint64_t hashWords(uint64_t h, const int32_t *p, size_t count) {
  uint64_t a = h ^ 0x9E3779B97F4A7C15ull, b = h + 0xC2B2AE3D27D4EB4Full, c = ~h, d = h * 0xD6E8FEB86659FD93ull + 1;
  size_t i = 0;
  for (; i + 8 <= count; i += 8) {
    uint64_t w[4];
    std::memcpy(w, p + i, 32);
    a = (a ^ w[0]) * 0xFF51AFD7ED558CCDull; a ^= a >> 29;
    b = (b ^ w[1]) * 0xC4CEB9FE1A85EC53ull; b ^= b >> 31;
    c = (c ^ w[2]) * 0x9FB21C651E98DF25ull; c ^= c >> 30;
    d = (d ^ w[3]) * 0xD6E8FEB86659FD93ull; d ^= d >> 32;
  }
  for (; i < count; ++i) { a = (a ^ static_cast<uint32_t>(p[i])) * 0xFF51AFD7ED558CCDull; a ^= a >> 29; }
  uint64_t r = a;
  r = (r ^ b) * 0xFF51AFD7ED558CCDull; r ^= r >> 32;
  r = (r ^ c) * 0xC4CEB9FE1A85EC53ull; r ^= r >> 29;
  r = (r ^ d) * 0x9FB21C651E98DF25ull; r ^= r >> 32;
  return r;
}

// runs body(thread index) on nth threads
template <class Body>
void parallelRun(unsigned nth, Body body) {
  cora::parallel_parts(nth, body);  // (joins everything and forwards the first exception: parallel.h)
}


}  // namespace

struct SymbolicCache::Impl {
  std::mutex m;
  // the most recent patterns, most recent first: preconditioner block, certificate matrix, the implicit formulation's M,
  // a shard's block -- two slots were one too few once prepareCertification() ran beside the first TNT solve and the
  // three patterns met in a nondeterministic order (round-4 advice)
  static constexpr int kSlots = 4;
  std::shared_ptr<const Symbolic> slot[kSlots];
};
SymbolicCache::SymbolicCache() : impl(new Impl) {}
SymbolicCache::~SymbolicCache() { delete impl; }

namespace {

std::shared_ptr<const Symbolic> symbolicFor(const SparseMatrix &A, int n, const std::vector<int32_t> &perm,
                                            const std::vector<int32_t> &iperm, bool *hit, SymbolicCache *cache) {
  *hit = false;
  // two independent 64-bit hashes of (outer, inner, perm) -- a stale hit would write a factor through the wrong column
  // counts, so the pair has to collide, not one word (round-2 advice).  The 17 M words are hashed in eight pieces on
  // threads; the pieces' hashes are chained in order.
  uint64_t key = static_cast<uint64_t>(n) * 0x100000001B3ull + static_cast<uint64_t>(A.rows());
  uint64_t key2 = 0x243F6A8885A308D3ull ^ static_cast<uint64_t>(perm.size());
  {
    const int32_t *arr[3] = {A.outer.data(), A.inner.data(), perm.data()};
    const size_t len[3] = {A.outer.size(), A.inner.size(), perm.size()};
    constexpr unsigned kPieces = 8;
    uint64_t h1[3][kPieces], h2[3][kPieces];
    const unsigned nt = A.inner.size() < (1u << 20) ? 1u : std::min(kPieces, std::max(1u, std::thread::hardware_concurrency()));
    parallelRun(nt, [&](unsigned t) {
      for (unsigned pc = t; pc < kPieces; pc += nt)
        for (int a = 0; a < 3; ++a) {
          const size_t lo = len[a] * pc / kPieces, hi = len[a] * (pc + 1) / kPieces;
          h1[a][pc] = hashWords(0x9E3779B97F4A7C15ull * (pc + 1) + a, arr[a] + lo, hi - lo);
          h2[a][pc] = hashWords(0xC2B2AE3D27D4EB4Full * (pc + 3) + 7 * a, arr[a] + lo, hi - lo);
        }
    });
    for (int a = 0; a < 3; ++a)
      for (unsigned pc = 0; pc < kPieces; ++pc) {
        key = (key ^ h1[a][pc]) * 0xFF51AFD7ED558CCDull;
        key ^= key >> 29;
        key2 = (key2 + h2[2 - a][pc]) * 0xC4CEB9FE1A85EC53ull;
        key2 ^= key2 >> 31;
      }
  }
  const bool use_cache = cache != nullptr && std::getenv("CORA_CHOL_NO_SYMBOLIC_CACHE") == nullptr;
  if (use_cache) {
    std::lock_guard<std::mutex> lock(cache->impl->m);
    auto &slot = cache->impl->slot;
    for (int e = 0; e < SymbolicCache::Impl::kSlots; ++e)
      if (slot[e] && slot[e]->key == key && slot[e]->key2 == key2 && slot[e]->n == n && slot[e]->nnzA == A.inner.size()) {
        std::rotate(slot, slot + e, slot + e + 1);  // to the front, the others keep their order
        *hit = true;
        return slot[0];
      }
  }
  const bool timing_ = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick_ = [t_prev = std::chrono::steady_clock::now(), timing_](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing_) std::fprintf(stderr, "      [symbolic] %-20s %.4f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  auto S = std::make_shared<Symbolic>();
  S->key = key;
  S->key2 = key2;
  S->n = n;
  S->nnzA = A.inner.size();
  // upper triangle of P A P^T by columns == rows of A restricted to iperm <= k
  std::vector<int32_t> &Cp = S->Cp, &Ci = S->Ci, &Cmap = S->Cmap;
  Cp.assign(static_cast<size_t>(n) + 1, 0);
  unsigned np_ = n < 20000 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *e = std::getenv("CORA_SYMBOLIC_THREADS")) np_ = static_cast<unsigned>(std::max(1, std::atoi(e)));
  parallelRun(np_, [&](unsigned t) {  // entries per column (columns are independent)
    for (int k = static_cast<int>(static_cast<int64_t>(n) * t / np_); k < static_cast<int>(static_cast<int64_t>(n) * (t + 1) / np_); ++k) {
      const int old = perm[k];
      int cnt = 0;
      bool diag = false;
      for (int32_t q = A.outer[old]; q < A.outer[old + 1]; ++q) {
        const int i = iperm[A.inner[q]];
        if (i >= 0 && i <= k) { ++cnt; diag |= (i == k); }
      }
      if (!diag) ++cnt;  // room for the shift on a structurally missing diagonal
      Cp[k + 1] = cnt;
    }
  });
  for (int k = 0; k < n; ++k) Cp[k + 1] += Cp[k];
  Ci.resize(static_cast<size_t>(Cp[n]));
  Cmap.resize(static_cast<size_t>(Cp[n]));
  S->Cdiag.assign(static_cast<size_t>(n), -1);
  parallelRun(np_, [&](unsigned t) {
    for (int k = static_cast<int>(static_cast<int64_t>(n) * t / np_); k < static_cast<int>(static_cast<int64_t>(n) * (t + 1) / np_); ++k) {
      const int old = perm[k];
      int32_t w = Cp[k];
      for (int32_t q = A.outer[old]; q < A.outer[old + 1]; ++q) {
        const int i = iperm[A.inner[q]];
        if (i >= 0 && i <= k) {
          Ci[w] = i;
          Cmap[w] = q;
          // (a duplicated diagonal entry cannot occur: setFromTriplets sums duplicates)
          if (i == k) S->Cdiag[k] = w;
          ++w;
        }
      }
      if (S->Cdiag[k] < 0) { Ci[w] = k; Cmap[w] = -1; S->Cdiag[k] = w; ++w; }
    }
  });
  tick_("pattern of P A P'");
  // elimination tree
  std::vector<int32_t> &parent = S->parent;
  parent.assign(static_cast<size_t>(n), -1);
  std::vector<int32_t> anc(static_cast<size_t>(n), -1);
  for (int k = 0; k < n; ++k)
    for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) {
      int i = Ci[q];
      while (i != -1 && i < k) {
        const int nxt = anc[i];
        anc[i] = k;
        if (nxt == -1) parent[i] = k;
        i = nxt;
      }
    }
  tick_("elimination tree");
  // column counts (symbolic up-looking pass)
  // (rows are independent given the tree: each thread walks its rows with marks and counters of its own; the counters
  // are added up at the end -- integer adds, the same totals in any order)
  std::vector<int32_t> &cnt = S->cnt;
  cnt.assign(static_cast<size_t>(n), 0);
  {
    std::vector<std::vector<int32_t>> acc(np_);
    parallelRun(np_, [&](unsigned t) {
      std::vector<int32_t> flag(static_cast<size_t>(n), -1), &mine = acc[t];
      mine.assign(static_cast<size_t>(n), 0);
      for (int k = static_cast<int>(t); k < n; k += static_cast<int>(np_)) {  // interleaved: rows near the root are the long walks
        flag[k] = k;
        mine[k]++;
        for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) {
          int i = Ci[q];
          while (i != -1 && i < k && flag[i] != k) {
            mine[i]++;
            flag[i] = k;
            i = parent[i];
          }
        }
      }
    });
    parallelRun(np_, [&](unsigned t) {
      for (int k = static_cast<int>(static_cast<int64_t>(n) * t / np_); k < static_cast<int>(static_cast<int64_t>(n) * (t + 1) / np_); ++k) {
        int32_t tot_k = 0;
        for (unsigned u = 0; u < np_; ++u) tot_k += acc[u][k];
        cnt[k] = tot_k;
      }
    });
  }
  tick_("column counts");
  S->Lp.assign(static_cast<size_t>(n) + 1, 0);
  int64_t tot = 0;
  for (int k = 0; k < n; ++k) {
    S->Lp[k] = static_cast<int32_t>(tot);
    tot += cnt[k];
    if (tot > 2000000000LL) throw std::runtime_error("choleskyFactor: factor too large for int32 indexing");
  }
  S->Lp[n] = static_cast<int32_t>(tot);
  S->tot = tot;
  if (use_cache) {
    std::lock_guard<std::mutex> lock(cache->impl->m);
    auto &slot = cache->impl->slot;
    std::rotate(slot, slot + SymbolicCache::Impl::kSlots - 1, slot + SymbolicCache::Impl::kSlots);  // the oldest falls out
    slot[0] = S;
  }
  return S;
}

// Per-thread scratch of the up-looking row solve, and the factor's storage, are kept from call to call: a fresh
// allocation of this size costs more in page faults than the arithmetic that runs in it (10^5 poses: 7 MB per thread
// times 16 threads, 60 MB of factor).  Invariant of a pooled Work: x is all zeros.
struct Work {
  std::vector<double> x;
  std::vector<int32_t> flag, stack;
};
std::mutex g_pool_mutex;
std::vector<std::unique_ptr<Work>> g_work_pool;
std::vector<std::vector<int32_t>> g_pool_i;
std::vector<std::vector<double>> g_pool_x;
constexpr size_t kWorkPoolMax = 64, kStoragePoolMax = 4;

// (what the pools may keep alive between calls: at 10^6 poses one Work is 72 MB and a factor 570 MB -- the pools
// serve the sizes where page faults matter and let the big ones go)
constexpr size_t kWorkPoolBytes = size_t(512) << 20, kStoragePoolBytes = size_t(1) << 30;
size_t g_work_pool_bytes = 0;

std::unique_ptr<Work> acquireWork(int n) {
  std::unique_ptr<Work> W;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    if (!g_work_pool.empty()) {
      W = std::move(g_work_pool.back());
      g_work_pool.pop_back();
      g_work_pool_bytes -= W->x.capacity() * sizeof(double) + (W->flag.capacity() + W->stack.capacity()) * sizeof(int32_t);
    }
  }
  if (!W) W = std::make_unique<Work>();
  const size_t m = static_cast<size_t>(n);
  if (W->x.size() < m) {
    W->x.assign(m, 0.0);
    W->flag.resize(m);
    W->stack.resize(m);
  }
  std::fill(W->flag.begin(), W->flag.begin() + static_cast<std::ptrdiff_t>(m), -1);
  return W;
}

void releaseWork(std::unique_ptr<Work> W) {
  const size_t bytes = W->x.capacity() * sizeof(double) + (W->flag.capacity() + W->stack.capacity()) * sizeof(int32_t);
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  if (g_work_pool.size() < kWorkPoolMax && g_work_pool_bytes + bytes <= kWorkPoolBytes) {
    g_work_pool_bytes += bytes;
    g_work_pool.push_back(std::move(W));
  }
}

// A reservation under way (choleskyReserveStorage: buffers being touched on a thread beside the ordering) has not handed its
// buffers to the pool yet: whoever asks the pool meanwhile waits for it instead of allocating a second full-size pair that would
// stay pooled beside the first (round-5 advice; 10^6 poses: hundreds of MB).  The reserving threads themselves do not wait.
std::atomic<int> g_reserving{0};
thread_local bool t_reserving = false;

template <class T>
std::vector<T> takeStorage(std::vector<std::vector<T>> &pool, size_t count) {
  std::vector<T> v;
  if (!t_reserving)
    while (g_reserving.load(std::memory_order_acquire) > 0) std::this_thread::yield();
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    for (size_t e = 0; e < pool.size(); ++e)
      if (pool[e].capacity() >= count) {
        v = std::move(pool[e]);
        pool.erase(pool.begin() + static_cast<std::ptrdiff_t>(e));
        break;
      }
  }
  v.resize(count);  // (entries the factorisation does not write are never read: see `next`)
  return v;
}

}  // namespace

CholeskyFactor::~CholeskyFactor() {
  // the storage of a factor goes back to the pool (the certificate's factor lives for one PSD test)
  if (Lx.capacity() < (1u << 20)) return;
  // (what goes is freed outside the lock)
  std::vector<std::vector<double>> drop_x;
  std::vector<std::vector<int32_t>> drop_i;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    const size_t mine = Lx.capacity() * sizeof(double) + Li.capacity() * sizeof(int32_t);
    if (mine > kStoragePoolBytes) return;
    auto held = [&] {
      size_t h = 0;
      for (const auto &v : g_pool_x) h += v.capacity() * sizeof(double);
      for (const auto &v : g_pool_i) h += v.capacity() * sizeof(int32_t);
      return h;
    };
    // The pool keeps the LARGEST buffers: a smaller one that stands in the way goes (10^6 poses: the preconditioner's
    // factor is 570 MB, the certificate's 580 MB -- with the first kept and the second let go, every PSD test of the
    // staircase paid 0.12 s of page faults for a fresh one).
    auto evict_smaller = [&](auto &pool, auto &drop, size_t cap) {
      size_t best = pool.size();
      for (size_t e = 0; e < pool.size(); ++e)
        if (pool[e].capacity() < cap && (best == pool.size() || pool[e].capacity() < pool[best].capacity())) best = e;
      if (best == pool.size()) return false;
      drop.push_back(std::move(pool[best]));
      pool.erase(pool.begin() + static_cast<std::ptrdiff_t>(best));
      return true;
    };
    while (held() + mine > kStoragePoolBytes || g_pool_x.size() >= kStoragePoolMax || g_pool_i.size() >= kStoragePoolMax) {
      const bool a = evict_smaller(g_pool_x, drop_x, Lx.capacity());
      const bool b = evict_smaller(g_pool_i, drop_i, Li.capacity());
      if (!a && !b) return;  // everything kept is at least as large: this one goes
    }
    g_pool_x.push_back(std::move(Lx));
    g_pool_i.push_back(std::move(Li));
  }
}

void choleskyReserveStorage(size_t entries) {
  if (entries < (1u << 20)) return;  // (small factors are not pooled)
  g_reserving.fetch_add(1, std::memory_order_acq_rel);
  struct Done {  // (released when the buffers are in the pool: after F's destructor, declared below, has run)
    ~Done() { g_reserving.fetch_sub(1, std::memory_order_acq_rel); }
  } done;
  CholeskyFactor F;  // (its destructor hands the storage to the pool, where choleskyAnalyze / choleskyFactor find it)
  t_reserving = true;
  std::thread idx([&] {
    t_reserving = true;
    try {
      F.Li = takeStorage(g_pool_i, entries);
    } catch (...) {  // (a reservation that fails is no reservation)
    }
  });
  try {
    F.Lx = takeStorage(g_pool_x, entries);
  } catch (...) {
  }
  idx.join();
  t_reserving = false;
}

void choleskyAnalyze(const SparseMatrix &A, int m, const std::vector<int32_t> &perm, SymbolicCache *cache) {
  if (!cache || m <= 0) return;
  std::vector<int32_t> iperm(static_cast<size_t>(A.rows()), -1);
  for (int i = 0; i < m; ++i) iperm[perm[i]] = i;
  bool hit = false;
  const std::shared_ptr<const Symbolic> sym = symbolicFor(A, m, perm, iperm, &hit, cache);
  CholeskyFactor F;  // (its destructor hands the storage to the pool)
  // first touch of 58 MB at 10^5 poses: the two arrays on a thread each (10 -> 6 ms)
  std::exception_ptr idx_error;
  std::thread idx([&] {
    try {
      F.Li = takeStorage(g_pool_i, static_cast<size_t>(sym->tot));
    } catch (...) {
      idx_error = std::current_exception();
    }
  });
  try {
    F.Lx = takeStorage(g_pool_x, static_cast<size_t>(sym->tot));
  } catch (...) {
    idx.join();
    throw;
  }
  idx.join();
  if (idx_error) std::rethrow_exception(idx_error);
}

CholeskyFactor choleskyFactor(const SparseMatrix &A, int m, double shift, const std::vector<int32_t> &perm,
                              SymbolicCache *cache) {
  CholeskyFactor F;
  const int n = m;
  F.n = n;
  if (static_cast<int>(perm.size()) != n) throw std::invalid_argument("choleskyFactor: bad permutation size");
  const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "    [cholesky] %-24s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  F.perm = perm;
  F.iperm.assign(static_cast<size_t>(A.rows()), -1);
  for (int i = 0; i < n; ++i) F.iperm[perm[i]] = i;

  // Everything that depends only on the pattern of A and on the order -- the permuted upper triangle's structure, the
  // elimination tree, the column counts -- is kept from one call to the next (`Symbolic`): the certificate matrix
  // S + eta I is factorised up to three times per staircase on one pattern, like CHOLMOD's analyze / factorize split
  // behind Eigen's analyzePattern.
  bool sym_hit = false;
  const std::shared_ptr<const Symbolic> sym = symbolicFor(A, n, perm, F.iperm, &sym_hit, cache);
  const std::vector<int32_t> &Cp = sym->Cp, &Ci = sym->Ci, &cnt = sym->cnt;
  const int64_t tot = sym->tot;
  std::vector<double> Cx(static_cast<size_t>(Cp[n]));
  {
    const std::vector<int32_t> &Cmap = sym->Cmap;
    const size_t nc = Cx.size();
    const unsigned nv = nc < (1u << 20) ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    parallelRun(nv, [&](unsigned th) {
      for (size_t w = nc * th / nv; w < nc * (th + 1) / nv; ++w) Cx[w] = Cmap[w] >= 0 ? A.values[Cmap[w]] : 0.0;
    });
    for (int k = 0; k < n; ++k) Cx[sym->Cdiag[k]] += shift;
  }
  tick(sym_hit ? "values into the cached pattern" : "permuted upper triangle + symbolic");
  F.parent = sym->parent;
  F.Lp = sym->Lp;
  F.Li = takeStorage(g_pool_i, static_cast<size_t>(tot));
  F.Lx = takeStorage(g_pool_x, static_cast<size_t>(tot));
  tick("factor storage");
  std::vector<int32_t> next(F.Lp.begin(), F.Lp.end() - 1);
  F.ok = true;
  // Row k of L (up-looking: a sparse triangular solve over the columns of k's elimination-tree descendants).  Touches
  // only columns of the subtree rooted at k, so disjoint subtrees can be factorised by different threads; the
  // arithmetic of a row does not depend on who runs it, so the factor is the sequential one bit for bit.
  auto process_row = [&](int k, Work &W) -> bool {
    std::vector<double> &x = W.x;
    std::vector<int32_t> &flag = W.flag, &stack = W.stack;
    int top = n;
    flag[k] = k;
    double dk = 0.0;
    for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) {
      int i = Ci[q];
      if (i == k) { dk += Cx[q]; continue; }
      x[i] += Cx[q];
      int len = 0;
      while (flag[i] != k) {
        stack[len++] = i;
        flag[i] = k;
        i = F.parent[i];
      }
      while (len > 0) stack[--top] = stack[--len];
    }
    for (; top < n; ++top) {
      const int i = stack[top];
      const double lki = x[i] / F.Lx[F.Lp[i]];
      x[i] = 0.0;
      for (int32_t q = F.Lp[i] + 1; q < next[i]; ++q) x[F.Li[q]] -= F.Lx[q] * lki;
      dk -= lki * lki;
      const int32_t w = next[i]++;
      F.Li[w] = k;
      F.Lx[w] = lki;
    }
    if (!(dk > 0.0)) return false;  // CHOLMOD's "not positive definite" (quick_return_if_not_posdef)
    const int32_t w = next[k]++;
    F.Li[w] = k;
    F.Lx[w] = std::sqrt(dk);
    return true;
  };
  int first_failure = n;  // sequential semantics: the smallest k whose pivot is not positive
  unsigned nth = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *e = std::getenv("CORA_CHOL_THREADS")) nth = static_cast<unsigned>(std::max(1, std::atoi(e)));
  std::vector<int32_t> task;  // row -> subtree task (-1: a separator above the tasks)
  int ntask = 0;
  std::vector<int32_t> tptr, trows;
  if (n < 20000 || nth <= 1) {
    std::unique_ptr<Work> W = acquireWork(n);
    for (int k = 0; k < n; ++k)
      if (!process_row(k, *W)) { first_failure = k; break; }
    releaseWork(std::move(W));
  } else {
    // tasks: maximal elimination-tree subtrees whose share of the factor's entries is below 1 / (8 threads)
    std::vector<int64_t> weight(cnt.begin(), cnt.end());
    for (int v = 0; v < n; ++v)
      if (F.parent[v] >= 0) weight[F.parent[v]] += weight[v];  // children come before parents
    const int64_t target = std::max<int64_t>(tot / (8 * static_cast<int64_t>(nth)), 1);
    task.assign(static_cast<size_t>(n), -1);
    std::vector<std::pair<int64_t, int32_t>> roots;  // (weight, root)
    for (int v = n - 1; v >= 0; --v) {
      const int p = F.parent[v];
      if (p >= 0 && task[p] >= 0) task[v] = task[p];
      else if (weight[v] <= target) {
        task[v] = static_cast<int32_t>(roots.size());
        roots.push_back({weight[v], v});
      }
    }
    ntask = static_cast<int>(roots.size());
    tptr.assign(static_cast<size_t>(ntask) + 1, 0);
    trows.resize(static_cast<size_t>(n));
    for (int v = 0; v < n; ++v)
      if (task[v] >= 0) tptr[task[v] + 1]++;
    for (int t = 0; t < ntask; ++t) tptr[t + 1] += tptr[t];
    {
      std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
      for (int v = 0; v < n; ++v)
        if (task[v] >= 0) trows[fill[task[v]]++] = v;  // ascending inside a task
    }
    std::vector<int32_t> order(static_cast<size_t>(ntask));
    for (int t = 0; t < ntask; ++t) order[t] = t;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return roots[a].first > roots[b].first; });
    std::atomic<int> next_task{0}, min_fail{n};
    parallelRun(nth, [&](unsigned) {
      std::unique_ptr<Work> W = acquireWork(n);
      for (;;) {
        const int slot = next_task.fetch_add(1);
        if (slot >= ntask) break;
        const int t = order[slot];
        for (int32_t q = tptr[t]; q < tptr[t + 1]; ++q) {
          const int k = trows[q];
          if (k >= min_fail.load(std::memory_order_relaxed)) break;  // rows past a failure are never looked at
          if (!process_row(k, *W)) {
            int cur = min_fail.load();
            while (k < cur && !min_fail.compare_exchange_weak(cur, k)) {}
            break;
          }
        }
      }
      releaseWork(std::move(W));
    });
    tick("numeric: subtrees (threads)");
    first_failure = min_fail.load();
    // What is left -- the separators above the tasks and, last, the rows every part of the graph reaches (the
    // landmarks of a CORA problem: ten rows of L that are nearly dense and take nine tenths of this phase at 10^5
    // poses when they run one after the other).  The trailing rows [k0, n) with a long row of A are taken together:
    //   A  every row solves against the columns below k0 on its own thread.  Columns are read up to where the rows
    //      before k0 left them and nothing is appended yet; the row's values stay in its scratch vector.
    //   B  what the trailing rows contribute to each other through those columns -- the running value
    //      x[k'] -= L[k', i] L[k, i] over row k's columns in their elimination order -- one pair (k, k') per work item;
    //   C  the small trailing triangle, row after row (the pivots are tested here);
    //   D  the rows' values go to the ends of their columns, row after row, every thread on a range of columns.
    // Each number is produced by the operations of the sequential elimination in their sequential order, so the
    // factor is the same bit for bit.  That relies on the row's elimination order visiting every column below k0
    // before the first trailing one; a row for which this does not hold sends the whole group down the plain path.
    int k0 = n;
    if (std::getenv("CORA_CHOL_NO_TRAILING_GROUP") == nullptr)
      while (k0 > 0 && n - k0 < 32 && task[k0 - 1] < 0 && Cp[k0] - Cp[k0 - 1] >= 64) --k0;
    if (n - k0 < 2) k0 = n;
    {
      std::unique_ptr<Work> W = acquireWork(n);
      for (int k = 0; k < std::min(first_failure, k0); ++k)
        if (task[k] < 0 && !process_row(k, *W)) { first_failure = k; break; }
      releaseWork(std::move(W));
    }
    tick("numeric: separators");
    if (k0 < n && first_failure >= k0) {
      const int ns = n - k0;
      std::vector<std::unique_ptr<Work>> rows(static_cast<size_t>(ns));
      std::vector<int32_t> top_of(static_cast<size_t>(ns), n), split_of(static_cast<size_t>(ns), n);
      std::vector<double> dk_of(static_cast<size_t>(ns), 0.0);
      std::vector<char> unordered(static_cast<size_t>(ns), 0);
      for (int r = 0; r < ns; ++r) rows[r] = acquireWork(n);
      const std::vector<int32_t> next0(next.begin(), next.begin() + k0);
      std::atomic<int> next_row{0};
      parallelRun(std::min<unsigned>(nth, static_cast<unsigned>(ns)), [&](unsigned) {
        for (;;) {
          const int r = next_row.fetch_add(1);
          if (r >= ns) break;
          const int k = k0 + r;
          Work &W = *rows[r];
          std::vector<double> &x = W.x;
          std::vector<int32_t> &flag = W.flag, &stack = W.stack;
          int top = n;
          flag[k] = k;
          double dk = 0.0;
          for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) {
            int i = Ci[q];
            if (i == k) { dk += Cx[q]; continue; }
            x[i] += Cx[q];
            int len = 0;
            while (flag[i] != k) {
              stack[len++] = i;
              flag[i] = k;
              i = F.parent[i];
            }
            while (len > 0) stack[--top] = stack[--len];
          }
          int split = top;
          while (split < n && stack[split] < k0) ++split;
          for (int t = split; t < n; ++t)
            if (stack[t] < k0) unordered[r] = 1;
          top_of[r] = top;
          split_of[r] = split;
          if (unordered[r]) continue;
          for (int t = top; t < split; ++t) {
            const int i = stack[t];
            const double lki = x[i] / F.Lx[F.Lp[i]];
            for (int32_t q = F.Lp[i] + 1; q < next0[i]; ++q) x[F.Li[q]] -= F.Lx[q] * lki;
            x[i] = lki;  // the row's value stays here until step D
            dk -= lki * lki;
          }
          dk_of[r] = dk;
        }
      });
      tick("  trailing A (rows)");
      bool plain = false;
      for (int r = 0; r < ns; ++r) plain |= unordered[r] != 0;
      if (plain) {
        for (int r = 0; r < ns; ++r) {  // undo step A's bookkeeping (no row went past the scatter of A's entries)
          Work &W = *rows[r];
          const int k = k0 + r;
          if (!unordered[r])
            for (int t = top_of[r]; t < n; ++t) W.x[W.stack[t]] = 0.0;
          for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) W.x[Ci[q]] = 0.0;
        }
        std::unique_ptr<Work> W = acquireWork(n);
        for (int k = k0; k < n; ++k)
          if (!process_row(k, *W)) { first_failure = k; break; }
        releaseWork(std::move(W));
      } else {
        // B: pairs (k, k'), k' < k
        std::vector<std::pair<int32_t, int32_t>> pairs;
        for (int r = ns - 1; r >= 1; --r)
          for (int r2 = 0; r2 < r; ++r2) pairs.push_back({r, r2});
        std::atomic<int> next_pair{0};
        parallelRun(std::min<unsigned>(nth, static_cast<unsigned>(pairs.size())), [&](unsigned) {
          for (;;) {
            const int e = next_pair.fetch_add(1);
            if (e >= static_cast<int>(pairs.size())) break;
            const int r = pairs[e].first, r2 = pairs[e].second;
            const Work &Wk = *rows[r], &Wo = *rows[r2];
            const int k2 = k0 + r2;
            double v = Wk.x[k2];
            const int32_t *st = Wk.stack.data();
            const int32_t *fo = Wo.flag.data();
            const double *xk = Wk.x.data(), *xo = Wo.x.data();
            for (int t = top_of[r]; t < split_of[r]; ++t) {
              const int i = st[t];
              if (fo[i] == k2) v -= xo[i] * xk[i];
            }
            rows[r]->x[k2] = v;  // (nobody else reads or writes this element during step B)
          }
        });
        tick("  trailing B (pairs)");
        // C: the trailing triangle
        int last_row = n - 1;
        for (int r = 0; r < ns; ++r) {
          const int k = k0 + r;
          Work &W = *rows[r];
          double dk = dk_of[r];
          for (int t = split_of[r]; t < n; ++t) {
            const int i = W.stack[t];
            const double lki = W.x[i] / F.Lx[F.Lp[i]];
            W.x[i] = 0.0;
            for (int32_t q = F.Lp[i] + 1; q < next[i]; ++q) W.x[F.Li[q]] -= F.Lx[q] * lki;
            dk -= lki * lki;
            const int32_t w = next[i]++;
            F.Li[w] = k;
            F.Lx[w] = lki;
          }
          if (!(dk > 0.0)) {
            first_failure = k;
            last_row = k;
            break;
          }
          const int32_t w = next[k]++;
          F.Li[w] = k;
          F.Lx[w] = std::sqrt(dk);
        }
        tick("  trailing C (triangle)");
        // D: values to the ends of their columns; scratch vectors back to zero
        const unsigned nd = nth;
        parallelRun(nd, [&](unsigned th) {
          const int lo = static_cast<int>(static_cast<int64_t>(k0) * th / nd), hi = static_cast<int>(static_cast<int64_t>(k0) * (th + 1) / nd);
          for (int r = 0; r < ns; ++r) {
            const int k = k0 + r;
            Work &W = *rows[r];
            const int32_t *fl = W.flag.data();
            double *x = W.x.data();
            const bool keep = k <= last_row;
            for (int i = lo; i < hi; ++i)
              if (fl[i] == k) {
                if (keep) {
                  const int32_t w = next[i]++;
                  F.Li[w] = k;
                  F.Lx[w] = x[i];
                }
                x[i] = 0.0;
              }
          }
        });
        for (int r = 0; r < ns; ++r)
          for (int i = k0; i < n; ++i) rows[r]->x[i] = 0.0;
      }
      for (int r = 0; r < ns; ++r) releaseWork(std::move(rows[r]));
    }
  }
  tick("numeric: trailing rows");
  if (first_failure < n) {
    const int k = first_failure;
    F.ok = false;
    F.failed_column = k;
    // The storage comes from a pool: what the stopped factorisation did not write is set to zero.  Direction of
    // non-positive curvature from the failing pivot: solve L11^T y = l_k.  Row k is the last one that was appended to
    // a column, so its value in column j, if it has one, is the column's last entry.  y[j] needs y of j's ancestors
    // only: the separators first, then every subtree task on its own (same numbers, whoever computes them).
    std::vector<double> y(static_cast<size_t>(k) + 1, 0.0);
    const unsigned nz = n < 20000 ? 1u : nth;
    parallelRun(nz, [&](unsigned th) {
      const int lo = static_cast<int>(static_cast<int64_t>(n) * th / nz), hi = static_cast<int>(static_cast<int64_t>(n) * (th + 1) / nz);
      for (int j = lo; j < hi; ++j) {
        for (int32_t q = next[j]; q < F.Lp[j + 1]; ++q) {
          F.Li[q] = 0;
          F.Lx[q] = 0.0;
        }
        if (j < k && next[j] > F.Lp[j] + 1 && F.Li[next[j] - 1] == k) y[j] = F.Lx[next[j] - 1];
      }
    });
    auto back = [&](int j) {
      double s = y[j];
      for (int32_t q = F.Lp[j] + 1; q < next[j]; ++q)
        if (F.Li[q] < k) s -= F.Lx[q] * y[F.Li[q]];
      y[j] = s / F.Lx[F.Lp[j]];
    };
    tick("  unwritten entries, row k");
    if (ntask == 0) {
      for (int j = k - 1; j >= 0; --j) back(j);
    } else {
      for (int j = k - 1; j >= 0; --j)
        if (task[j] < 0) back(j);
      tick("  back: separators");
      std::atomic<int> next_task{0};
      parallelRun(nth, [&](unsigned) {
        for (;;) {
          const int t = next_task.fetch_add(1);
          if (t >= ntask) break;
          for (int32_t q = tptr[t + 1] - 1; q >= tptr[t]; --q)
            if (trows[q] < k) back(trows[q]);
        }
      });
    }
    tick("  back: tasks");
    F.negative_direction.assign(static_cast<size_t>(A.rows()), 0.0);
    double nrm = 1.0;
    for (int j = 0; j < k; ++j) nrm += y[j] * y[j];
    nrm = std::sqrt(nrm);
    for (int j = 0; j < k; ++j) F.negative_direction[perm[j]] = -y[j] / nrm;
    F.negative_direction[perm[k]] = 1.0 / nrm;
    tick("direction of non-positive curvature");
  }
  return F;
}

CholeskyFactor incompleteLDLT(const SparseMatrix &A, int m, double shift, const std::vector<int32_t> &perm,
                              double max_fill_factor, double drop_tol, int *negative_pivots) {
  CholeskyFactor F;
  const int n = m;
  F.n = n;
  if (static_cast<int>(perm.size()) != n) throw std::invalid_argument("incompleteLDLT: bad permutation size");
  F.perm = perm;
  F.iperm.assign(static_cast<size_t>(A.rows()), -1);
  for (int i = 0; i < n; ++i) F.iperm[perm[i]] = i;
  // columns of L (unit diagonal implied while eliminating) as growing lists; `first[j]`: position in column j of its
  // first entry with row >= the current column; `list[k]`: columns j < k with an entry in row k (linked)
  std::vector<std::vector<int32_t>> Lrow(static_cast<size_t>(n));
  std::vector<std::vector<double>> Lval(static_cast<size_t>(n));
  std::vector<double> d(static_cast<size_t>(n), 0.0);
  std::vector<int32_t> first(static_cast<size_t>(n), 0), next_in_row(static_cast<size_t>(n), -1), head(static_cast<size_t>(n), -1);
  std::vector<double> w(static_cast<size_t>(n), 0.0);
  std::vector<char> mark(static_cast<size_t>(n), 0);
  std::vector<int32_t> pattern;
  int neg = 0;
  for (int k = 0; k < n; ++k) {
    // w = (A + shift I)[k:, k] in the permuted order
    pattern.clear();
    const int old = perm[k];
    int a_cnt = 0;
    double dk = shift;
    for (int32_t q = A.outer[old]; q < A.outer[old + 1]; ++q) {
      const int i = F.iperm[A.inner[q]];
      if (i < 0) continue;
      ++a_cnt;
      if (i == k) dk += A.values[q];
      else if (i > k) {
        if (!mark[i]) { mark[i] = 1; pattern.push_back(i); }
        w[i] += A.values[q];
      }
    }
    // minus the contributions of the columns j < k with L_kj != 0
    for (int j = head[k]; j >= 0;) {
      const int nj = next_in_row[j];
      const std::vector<int32_t> &rj = Lrow[j];
      const std::vector<double> &vj = Lval[j];
      const int32_t f = first[j];  // rj[f] == k
      const double lkj_d = vj[f] * d[j];
      dk -= vj[f] * lkj_d;
      for (size_t t = static_cast<size_t>(f) + 1; t < rj.size(); ++t) {
        const int i = rj[t];
        if (!mark[i]) { mark[i] = 1; pattern.push_back(i); }
        w[i] -= vj[t] * lkj_d;
      }
      // column j moves on to its next row
      first[j] = f + 1;
      if (static_cast<size_t>(f) + 1 < rj.size()) {
        const int r2 = rj[f + 1];
        next_in_row[j] = head[r2];
        head[r2] = j;
      }
      j = nj;
    }
    double col1 = std::fabs(dk);
    for (int32_t i : pattern) col1 += std::fabs(w[i]);
    if (std::fabs(dk) < 1e-10 * col1 || dk == 0.0) dk = (dk < 0 ? -1.0 : 1.0) * std::max(1e-10 * col1, 1e-300);
    if (dk < 0) ++neg;
    d[k] = dk;
    // dropping: small entries, then all but the largest `keep`
    const double tol = drop_tol * col1;
    std::vector<std::pair<double, int32_t>> ent;
    ent.reserve(pattern.size());
    for (int32_t i : pattern) {
      const double l = w[i] / dk;
      if (std::fabs(w[i]) > tol) ent.push_back({std::fabs(l), i});
      w[i] = l;  // keep the scaled value for the survivors below
    }
    const size_t keep = static_cast<size_t>(std::ceil(max_fill_factor * std::max(a_cnt, 1)));
    if (ent.size() > keep) {
      std::nth_element(ent.begin(), ent.begin() + static_cast<std::ptrdiff_t>(keep), ent.end(),
                       [](const auto &x, const auto &y) { return x.first > y.first; });
      ent.resize(keep);
    }
    std::sort(ent.begin(), ent.end(), [](const auto &x, const auto &y) { return x.second < y.second; });
    std::vector<int32_t> &rk = Lrow[k];
    std::vector<double> &vk = Lval[k];
    rk.reserve(ent.size());
    vk.reserve(ent.size());
    for (const auto &e : ent) {
      rk.push_back(e.second);
      vk.push_back(w[e.second]);
    }
    for (int32_t i : pattern) { w[i] = 0.0; mark[i] = 0; }
    first[k] = 0;
    if (!rk.empty()) {
      next_in_row[k] = head[rk[0]];
      head[rk[0]] = k;
    }
  }
  if (negative_pivots) *negative_pivots = neg;
  // Cholesky form of L |D| L^T: column k = sqrt|d_k| * [1; L[k+1:, k]]
  F.Lp.assign(static_cast<size_t>(n) + 1, 0);
  int64_t tot = 0;
  for (int k = 0; k < n; ++k) {
    F.Lp[k] = static_cast<int32_t>(tot);
    tot += 1 + static_cast<int64_t>(Lrow[k].size());
    if (tot > 2000000000LL) throw std::runtime_error("incompleteLDLT: factor too large for int32 indexing");
  }
  F.Lp[n] = static_cast<int32_t>(tot);
  F.Li.resize(static_cast<size_t>(tot));
  F.Lx.resize(static_cast<size_t>(tot));
  F.parent.assign(static_cast<size_t>(n), -1);
  for (int k = 0; k < n; ++k) {
    const double sk = std::sqrt(std::fabs(d[k]));
    int32_t at = F.Lp[k];
    F.Li[at] = k;
    F.Lx[at++] = sk;
    for (size_t t = 0; t < Lrow[k].size(); ++t) {
      F.Li[at] = Lrow[k][t];
      F.Lx[at++] = Lval[k][t] * sk;
    }
    if (!Lrow[k].empty()) F.parent[k] = Lrow[k][0];
  }
  F.ok = true;
  return F;
}

void CholeskyFactor::solveInPlace(Matrix &B) const {
  if (!ok) throw std::runtime_error("CholeskyFactor::solve: factorisation failed");
  std::vector<double> y(static_cast<size_t>(n));
  for (Index c = 0; c < B.cols(); ++c) {
    for (int i = 0; i < n; ++i) y[i] = B(perm[i], c);
    for (int j = 0; j < n; ++j) {
      y[j] /= Lx[Lp[j]];
      const double yj = y[j];
      for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) y[Li[q]] -= Lx[q] * yj;
    }
    for (int j = n - 1; j >= 0; --j) {
      double s = y[j];
      for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) s -= Lx[q] * y[Li[q]];
      y[j] = s / Lx[Lp[j]];
    }
    for (int i = 0; i < n; ++i) B(perm[i], c) = y[i];
  }
}

}  // namespace CORA
