#include "trisolve.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <functional>
#include <future>
#include <thread>
#include <utility>
#include "parallel.h"

namespace cora {

namespace {
// Tunables below marked "env" are read ONCE, when the library is loaded (lab sweeps set them per process: tools/plan_sweep.sh).
// They were process-wide variables that every build_tri_plan call rewrote from the environment -- a data race between the
// rank threads that build their plans at the same time (round-4 advice).
int64_t env_i64(const char *name, int64_t dflt, int64_t lo, int64_t hi) {
  const char *e = std::getenv(name);
  return e ? std::min(hi, std::max(lo, static_cast<int64_t>(std::atoll(e)))) : dflt;
}
constexpr int kBorderRowNnz = 4096;  // rows of L longer than this (landmarks) always belong to the last stage
constexpr int kFirstCap = 64;        // rows of a stage-0 subtree: what one wavefront holds (a pair of
                                     // nested-dissection leaves, their separator and range rows); 32 / 40 / 48 /
                                     // 64 measured 154 / 155 / 162 / 151 us per apply at 10^5 poses
constexpr int kCapGrowth = 16;       // stage k subtrees hold up to kFirstCap * kCapGrowth^k rows
constexpr int kTopCap = 1536;        // stop cutting once this few rows are left: they form the last stage
const int64_t kTopInverseNnz = env_i64("CORA_TRI_TOP_INV", 2500000, 0, INT64_MAX);  // ... or once the inverse of what is left has this few entries:
                                             // a launch floor (~5 us) is worth ~25 MB of traffic, so small
                                             // factors are applied as ONE explicit inverse (2 products)
const int kShortRow = static_cast<int>(env_i64("CORA_TRI_SHORT_ROW", 64, 8, 1 << 30));  // entries: <= this -> 8 lanes per row
const int kWaveRow = static_cast<int>(env_i64("CORA_TRI_WAVE_ROW", 1024, 64, 1 << 30));  // entries: <= this -> one wavefront per row, else chunked
const int kChunk = static_cast<int>(env_i64("CORA_TRI_CHUNK", 512, 64, 1 << 30));
constexpr int kDenseBlock = 64;      // stage-0 blocks up to this many rows use the dense wavefront kernel
const int kSubRows = static_cast<int>(env_i64("CORA_TRI_SUB_ROWS", 512, 32, 2048));  // workgroup blocks (SubBlockOpHost): rows of a block (its tile of right-hand sides sits in LDS:
const int kSubEnt = static_cast<int>(env_i64("CORA_TRI_SUB_ENT", 5000, 100, 1 << 30));  // 512 rows x 24 columns = 96 KB) and entries of L per block (streamed since round 2: a cap on a block's work, not on LDS)
constexpr int kSnCapChain = 4;  // rows of a supernode of the substitution blocks (a 3-D pose: 3 rotation rows + translation)
const int kLaneEntries = static_cast<int>(env_i64("CORA_TRI_LANE_ENTRIES", 8, 1, 8));  // entries one lane of a row walks through (<= kSubNpl of the kernel: they sit in registers)
const int kLevelLanes = static_cast<int>(env_i64("CORA_TRI_LEVEL_LANES", 256, 64, 256));  // rows x lanes per row of one level (<= 256 = kSubThreads of the kernel)
constexpr int kSubWaves = 4, kWaveLanes = 64;  // wavefronts of a substitution block's workgroup (kSubThreads of the kernel / 64)
const int kSplitMinRows = static_cast<int>(env_i64("CORA_TRI_SPLIT_MIN_ROWS", 1 << 20, 1, 1 << 30));  // a chunk of a level is closed early when the next rows are half as long, from this many rows on.
                              // Never, since round 4: closing early saves padding (tile reads 15.4 M instead of 16.3 M at 10^5
                              // poses) and costs barrier levels (25.0 k instead of 20.0 k); measured: 10^5 poses 117.6 / 118.4 us
                              // per iteration (16 / never), 10^4 poses 62.9 / 58.4, tiers 83.1 / 78.6 us per product, mrclam3b 85.7 / 77.0
constexpr int kMinBlock = 8;         // smaller subtrees are left to the next stage (a wavefront per block would idle)

std::atomic<int64_t> g_seg_waves{0}, g_seg_reads{0}, g_seg_real{0}, g_seg_levels{0}, g_seg_sublevels{0}, g_cur_reads{0}, g_cur_sublevels{0};  // (timing mode)

struct RowList {  // rows of one product before they are sorted into length classes
  std::vector<int32_t> out, ptr{0}, col;
  std::vector<double> val;
  void begin_row(int32_t out_row) { out.push_back(out_row); }
  void add(int32_t c, double v) { col.push_back(c); val.push_back(v); }
  void end_row() { ptr.push_back(static_cast<int32_t>(col.size())); }
};

void finalize(const RowList &R, RowOpHost &op) {
  const int n = static_cast<int>(R.out.size());
  std::vector<int32_t> s8, s64, sl;
  for (int i = 0; i < n; ++i) {
    const int len = R.ptr[i + 1] - R.ptr[i];
    (len <= kShortRow ? s8 : (len <= kWaveRow ? s64 : sl)).push_back(i);
  }
  op = RowOpHost();
  op.n8 = static_cast<int32_t>(s8.size());
  op.n64 = static_cast<int32_t>(s64.size());
  op.col.reserve(R.col.size());
  op.val.reserve(R.val.size());
  auto copy_entries = [&](int i) {
    op.col.insert(op.col.end(), R.col.begin() + R.ptr[i], R.col.begin() + R.ptr[i + 1]);
    op.val.insert(op.val.end(), R.val.begin() + R.ptr[i], R.val.begin() + R.ptr[i + 1]);
  };
  for (const auto *cls : {&s8, &s64})
    for (int32_t i : *cls) {
      op.out_row.push_back(R.out[i]);
      op.begin.push_back(static_cast<int32_t>(op.col.size()));
      copy_entries(i);
      op.end.push_back(static_cast<int32_t>(op.col.size()));
    }
  op.long_chunk_ptr.push_back(0);
  for (int32_t i : sl) {
    op.long_out.push_back(R.out[i]);
    const int32_t b = static_cast<int32_t>(op.col.size());
    copy_entries(i);
    const int32_t e = static_cast<int32_t>(op.col.size());
    for (int32_t c = b; c < e; c += kChunk) {
      op.chunk_begin.push_back(c);
      op.chunk_end.push_back(std::min(e, c + kChunk));
    }
    op.long_chunk_ptr.push_back(static_cast<int32_t>(op.chunk_begin.size()));
  }
}
}  // namespace

void build_tri_plan(int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                    const std::vector<int32_t> &row_of, int32_t zero_row, TriPlan &P,
                    const std::vector<int32_t> *group, int32_t aux_base) {
  int kSnCap = kSnCapChain;  // (local: plans are built from several rank threads at once)
  if (const char *e = std::getenv("CORA_TRI_SN_CAP")) kSnCap = std::min(32, std::max(1, std::atoi(e)));
  const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  [tri plan] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  P = TriPlan();
  P.m = m;
  P.zero_row = zero_row;
  if (m <= 0) return;
  P.nnzL = Lp[m];
  // ---- elimination tree and row lengths, CSR of the strictly lower part (rows of L, columns ascending).
  // Threads take ascending ranges of columns (equal shares of the entries): they check their columns, count their
  // entries per row, and the counts of the threads before give each its place in the row.
  std::vector<int32_t> parent(m, -1), rcount(m, 0);
  bool sorted = true;  // columns with ascending row indices: the rows of a column's own block come first
  for (int i = 0; i < m; ++i)
    if (row_of[i] < 0) throw std::runtime_error("cora: factor row outside the handle");
  const unsigned hw0 = std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt0 = Lp[m] < (1 << 20) ? 1u : std::min(8u, hw0);
  std::vector<int> cut(nt0 + 1, m);
  cut[0] = 0;
  for (unsigned t = 1; t < nt0; ++t) {
    const int32_t want = static_cast<int32_t>(static_cast<int64_t>(Lp[m]) * t / nt0);
    cut[t] = static_cast<int>(std::lower_bound(Lp, Lp + m, want) - Lp);
  }
  std::vector<std::vector<int32_t>> at(nt0);
  std::vector<const char *> bad(nt0, nullptr);
  std::vector<char> unsorted(nt0, 0);
  auto run0 = [&](auto body) {
    cora::parallel_parts(nt0, body);
  };
  run0([&](unsigned t) {
    std::vector<int32_t> &a = at[t];
    a.assign(static_cast<size_t>(m), 0);
    for (int j = cut[t]; j < cut[t + 1]; ++j) {
      if (Li[Lp[j]] != j) { bad[t] = "cora: Cholesky factor must store the diagonal first in each column"; return; }
      for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) {
        if (Li[q] <= j || Li[q] >= m) { bad[t] = "cora: Cholesky factor has an entry above the diagonal"; return; }
        if (parent[j] < 0 || Li[q] < parent[j]) parent[j] = Li[q];
        if (q > Lp[j] + 1 && Li[q] < Li[q - 1]) unsorted[t] = 1;
        a[Li[q]]++;
      }
    }
  });
  for (unsigned t = 0; t < nt0; ++t) {
    if (bad[t]) throw std::runtime_error(bad[t]);
    if (unsorted[t]) sorted = false;
  }
  run0([&](unsigned t) {
    for (int i = static_cast<int>(static_cast<int64_t>(m) * t / nt0); i < static_cast<int>(static_cast<int64_t>(m) * (t + 1) / nt0); ++i) {
      int32_t n = 0;
      for (unsigned u = 0; u < nt0; ++u) n += at[u][i];
      rcount[i] = n;
    }
  });
  std::vector<int32_t> rptr(static_cast<size_t>(m) + 1, 0);
  for (int i = 0; i < m; ++i) rptr[i + 1] = rptr[i] + rcount[i];
  std::vector<int32_t> rcol(static_cast<size_t>(rptr[m]));
  std::vector<double> rval(static_cast<size_t>(rptr[m]));
  run0([&](unsigned t) {  // counts -> start positions
    for (int i = static_cast<int>(static_cast<int64_t>(m) * t / nt0); i < static_cast<int>(static_cast<int64_t>(m) * (t + 1) / nt0); ++i) {
      int32_t pos = rptr[i];
      for (unsigned u = 0; u < nt0; ++u) {
        const int32_t n_u = at[u][i];
        at[u][i] = pos;
        pos += n_u;
      }
    }
  });
  run0([&](unsigned t) {
    std::vector<int32_t> &a = at[t];
    for (int j = cut[t]; j < cut[t + 1]; ++j)
      for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) {
        const int32_t w = a[Li[q]]++;
        rcol[w] = j;
        rval[w] = Lx[q];
      }
  });
  at.clear();
  tick("rows of L");
  // ---- elimination tree of the factor's own pattern (Liu's algorithm with path compression).  For a complete
  // Cholesky factor this is "parent = first sub-diagonal row of the column"; an INCOMPLETE factor (dropped entries)
  // only keeps the property the stages below rely on -- L_ij != 0 implies that i is an ancestor of j -- with the
  // tree of its actual pattern.
  // A complete factor is recognised by its pattern being closed under elimination -- the rows of column j below its
  // first one are rows of that first one's column -- and then the first sub-diagonal rows found above ARE the tree
  // (checked column by column on threads, sorted columns: one merge walk each); anything else takes Liu's algorithm.
  bool closed = sorted;
  if (closed) {
    std::vector<char> open_(nt0, 0);
    run0([&](unsigned t) {
      for (int j = static_cast<int>(static_cast<int64_t>(m) * t / nt0); j < static_cast<int>(static_cast<int64_t>(m) * (t + 1) / nt0) && !open_[t]; ++j) {
        const int pj = parent[j];
        if (pj < 0) continue;
        int32_t a = Lp[j] + 2, b = Lp[pj] + 1;  // rows of column j after its first one; rows of column pj after the diagonal
        const int32_t ae = Lp[j + 1], be = Lp[pj + 1];
        while (a < ae) {
          while (b < be && Li[b] < Li[a]) ++b;
          if (b == be || Li[b] != Li[a]) { open_[t] = 1; break; }
          ++a;
        }
      }
    });
    for (unsigned t = 0; t < nt0; ++t) closed = closed && !open_[t];
  }
  const bool check_tree = closed && std::getenv("CORA_TRI_CHECK_ETREE") != nullptr;  // test hook: both ways, must agree
  const std::vector<int32_t> parent_closed = check_tree ? parent : std::vector<int32_t>();
  if (!closed || check_tree) {
    std::fill(parent.begin(), parent.end(), -1);
    std::vector<int32_t> anc(static_cast<size_t>(m), -1);
    for (int i = 0; i < m; ++i)
      for (int32_t q = rptr[i]; q < rptr[i + 1]; ++q) {
        int j = rcol[q];
        while (j != -1 && j < i) {
          const int nxt = anc[j];
          anc[j] = i;
          if (nxt == -1) parent[j] = i;
          j = nxt;
        }
      }
  }
  if (check_tree && parent != parent_closed) throw std::logic_error("cora: the closed-pattern elimination tree differs from Liu's");
  // Supernodes of the substitution blocks: 4 rows (one 3-D pose) on the shallow trees of single chains; graphs of several
  // robots that range each other dissect into separators of one pose PER ROBOT -- chains of 10 to 15 rows in a tree
  // ten times as high for its size (MR.CLAM 3b: 560 levels at 20 k rows, a chain of 450 k rows: 77) -- and there 8 rows
  // per supernode take 10-15 % off an iteration (mrclam6 115 -> 103, tiers 98 -> 84 us per product end to end), while
  // on the chains they cost 8 % (rows get longer).  16 exceeds what a level of the kernel holds.
  if (!std::getenv("CORA_TRI_SN_CAP")) {
    std::vector<int32_t> depth(static_cast<size_t>(m), 1);
    int height = 0;
    for (int i = 0; i < m; ++i) {
      if (parent[i] >= 0) depth[parent[i]] = std::max(depth[parent[i]], depth[i] + 1);
      height = std::max(height, depth[i]);
    }
    kSnCap = height > 8.0 * std::log2(static_cast<double>(std::max(m, 2))) ? 8 : 4;
  }
  tick("elimination tree");
  // ---- stages: repeatedly peel the maximal subtrees of the remaining forest that fit the cap
  int first_border = m;  // trailing run of long rows: forced into the last stage
  while (first_border > 0 && rcount[first_border - 1] > kBorderRowNnz) --first_border;
  std::vector<int32_t> stage(m, -1), blk(m, -1), sz(m, 0);
  int nstage = 0;
  int64_t remaining = first_border, cap = kFirstCap;
  std::vector<int32_t> depth(m, 0);
  auto inverse_nnz_of_rest = [&]() {  // entries of the explicit inverse of everything not taken yet
    int64_t inv_nnz = 0;
    for (int v = m - 1; v >= 0; --v)
      if (stage[v] < 0) {
        depth[v] = 1 + (parent[v] >= 0 && stage[parent[v]] < 0 ? depth[parent[v]] : 0);
        inv_nnz += depth[v];
      }
    return inv_nnz;
  };
  // ---- two-stage form: workgroup blocks solved by substitution + ONE explicit inverse of what is left
  bool sub0 = false;
  {
    const char *e = std::getenv("CORA_TRI_SUB");
    const bool want = aux_base >= 0 && !(e && std::atoi(e) == 0);
    if (want && remaining + (m - first_border) > kTopCap && inverse_nnz_of_rest() > kTopInverseNnz) {
      std::vector<int64_t> esz(m, 0);
      for (int v = 0; v < first_border; ++v) { sz[v] = 1; esz[v] = Lp[v + 1] - Lp[v] - 1; }
      for (int v = 0; v < first_border; ++v) {
        const int p = parent[v];
        if (p >= 0 && p < first_border) { sz[p] += sz[v]; esz[p] += esz[v]; }
      }
      // Height of every subtree in rows (a block's barrier levels grow with it: a supernode of <= kSnCap rows per level).
      // A launch of the sweep lasts as long as its DEEPEST block -- on the reference's mid-size data sets a block is alone
      // on its CU (mrclam6: 17 levels where the mean is 7.8, profiles/r05_kernel_evolution.md step 13) -- so subtrees much
      // taller than the typical block are not taken whole: their top rows go to the last stage and what hangs below them
      // becomes blocks of its own.  Two passes: the blocks the size caps alone would give, the median of their heights,
      // then the same selection with heights capped at CORA_TRI_LEVEL_CAP times that median.  OFF by default (0): measured
      // in round 6 (profiles/r06_kernel_evolution.md) at 1.25 / 1.5 / 2.0 -- mrclam6 85.7-87.6 -> 88.8-91.6 / 88.0-89.1 /
      // 85.8-87.5 us per product end to end, tiers 68.1-68.5 -> 70.6 / 68.8-69.2 / 67.8-68.4, mrclam3b 65.8-66.3 -> 65.3-67.2 /
      // 70.7-74.7 / 66.9-67.1, 10^5 poses unchanged: what the deepest blocks lose the last stage's products gain (the rows cut
      // off the tall subtrees join the explicit inverse).  Kept as a switch.
      std::vector<int32_t> hgt(static_cast<size_t>(first_border), 1);
      for (int v = 0; v < first_border; ++v) {
        const int p = parent[v];
        if (p >= 0 && p < first_border) hgt[p] = std::max(hgt[p], hgt[v] + 1);
      }
      int64_t taken = 0;
      int nblocks = 0;
      auto select = [&](int32_t max_height, std::vector<int32_t> *heights) {
        taken = 0;
        nblocks = 0;
        for (int v = first_border - 1; v >= 0; --v) {
          const int p = parent[v];
          if (p >= 0 && p < first_border && stage[p] == 0) {
            stage[v] = 0;
            blk[v] = blk[p];
            ++taken;
          } else if (sz[v] <= kSubRows && esz[v] <= kSubEnt && hgt[v] <= max_height &&
                     (sz[v] >= kMinBlock || p < 0 || p >= first_border) &&
                     !(group && p >= 0 && (*group)[v] >= 0 && (*group)[v] == (*group)[p])) {
            stage[v] = 0;
            blk[v] = v;
            ++nblocks;
            ++taken;
            if (heights) heights->push_back(hgt[v]);
          }
        }
      };
      const double level_cap = [] { const char *e = std::getenv("CORA_TRI_LEVEL_CAP"); return e ? std::atof(e) : 0.0; }();
      std::vector<int32_t> heights;
      select(INT32_MAX, &heights);
      if (level_cap > 0.0 && heights.size() >= 8) {
        std::nth_element(heights.begin(), heights.begin() + heights.size() / 2, heights.end());
        const int32_t median = heights[heights.size() / 2];
        const int32_t max_height = static_cast<int32_t>(std::ceil(level_cap * median));
        if (*std::max_element(heights.begin(), heights.end()) > max_height) {
          if (timing) std::fprintf(stderr, "  [tri plan] block heights: median %d rows, cap %d\n", median, max_height);
          std::fill(stage.begin(), stage.end(), -1);
          std::fill(blk.begin(), blk.end(), -1);
          select(max_height, nullptr);
        }
      }
      // what is left above the blocks is applied as one explicit inverse: worth it while its entries stay a fraction of
      // the factor's (10^5 poses: 0.2 M of 4.8 M; 10^6 poses: the separators of 10^4 blocks no longer fit the fixed
      // cap meant for SMALL factors, and the plan fell back to the three explicit stages of round 1: 1.3 ms per apply)
      if (taken > 0 && inverse_nnz_of_rest() <= std::max<int64_t>(kTopInverseNnz, P.nnzL / 2)) {
        sub0 = true;
        nstage = 1;
        remaining -= taken;
        P.stages.resize(1);
        P.stages[0].blocks = nblocks;
      } else {
        std::fill(stage.begin(), stage.end(), -1);
        std::fill(blk.begin(), blk.end(), -1);
      }
    }
  }
  while (!sub0 && remaining + (m - first_border) > kTopCap && cap < 4LL * m) {
    // entries of the explicit inverse of everything not taken yet = sum over its rows of the number of
    // remaining ancestors (the inverse of a Cholesky factor is non-zero exactly along tree paths)
    int64_t inv_nnz = 0;
    for (int v = m - 1; v >= 0; --v)
      if (stage[v] < 0) {
        depth[v] = 1 + (parent[v] >= 0 ? depth[parent[v]] : 0);
        inv_nnz += depth[v];
      }
    if (inv_nnz <= kTopInverseNnz) break;
    for (int v = 0; v < first_border; ++v) sz[v] = stage[v] < 0 ? 1 : 0;
    for (int v = 0; v < first_border; ++v) {
      const int p = parent[v];
      if (stage[v] < 0 && p >= 0 && p < first_border) sz[p] += sz[v];  // children come before parents
    }
    int64_t taken = 0;
    for (int v = first_border - 1; v >= 0; --v) {
      if (stage[v] >= 0) continue;
      const int p = parent[v];
      if (p >= 0 && p < first_border && stage[p] == nstage) {  // inside a subtree taken in this round
        stage[v] = nstage;
        blk[v] = blk[p];
        ++taken;
      } else if (sz[v] <= cap && (nstage > 0 || sz[v] >= kMinBlock || p < 0 || p >= first_border) &&
                 !(group && p >= 0 && (*group)[v] >= 0 && (*group)[v] == (*group)[p])) {
        // (a group -- the d rotation rows of a pose -- is never cut: its rows share a block)
        // maximal: its parent (if any) was visited and did not fit.  Tiny stage-0 subtrees hanging off a
        // bigger remainder (single range rows of separator poses) stay with that remainder.
        stage[v] = nstage;
        blk[v] = v;
        P.stages.resize(static_cast<size_t>(nstage) + 1);
        P.stages[nstage].blocks++;
        ++taken;
      }
    }
    cap *= kCapGrowth;
    if (taken == 0) continue;
    remaining -= taken;
    ++nstage;
  }
  for (int v = 0; v < m; ++v)
    if (stage[v] < 0) {
      stage[v] = nstage;
      blk[v] = m;  // one block: the top of the tree and the long rows
    }
  const int K = nstage + 1;
  P.height = K;
  P.stages.resize(static_cast<size_t>(K));
  P.stages[K - 1].blocks = 1;
  for (int v = 0; v < m; ++v) P.stages[stage[v]].rows++;
  // for L_ij != 0 the column j is a descendant of the row i, and every round takes whole subtrees of
  // what is left, so stage[j] <= stage[i]
  // ---- stage 0 in dense form when there is more than one stage and every block fits a wavefront
  std::vector<int32_t> loc(static_cast<size_t>(m), 0), blk_id(static_cast<size_t>(m), -1);
  bool dense0 = K > 1 && !sub0;
  if (sub0) {
    P.stages[0].sub = true;
    P.aux_base = aux_base;
    P.groups_whole = group != nullptr;  // a group is never cut, and substitution blocks have no lane layout to respect
  }
  {
    std::vector<int32_t> bsz(static_cast<size_t>(m) + 1, 0);
    for (int v = 0; v < m; ++v)
      if (stage[v] == 0) bsz[blk[v]]++;
    for (int v = 0; v <= m && dense0; ++v) dense0 = bsz[v] <= kDenseBlock;
  }
  BlockOpHost &D0 = P.stages[0].blocks_op;
  if (dense0) {
    P.stages[0].dense = true;
    // blocks in order of their roots; rows of a block in elimination order
    std::vector<int32_t> id_of_root(static_cast<size_t>(m), -1), count;
    for (int v = 0; v < m; ++v)
      if (stage[v] == 0 && blk[v] == v) {
        id_of_root[v] = static_cast<int32_t>(count.size());
        count.push_back(0);
      }
    for (int v = 0; v < m; ++v)
      if (stage[v] == 0) {
        blk_id[v] = id_of_root[blk[v]];
        loc[v] = count[blk_id[v]]++;
      }
    const size_t nblk = count.size();
    D0.row_begin.assign(nblk, 0);
    D0.nrows.assign(nblk, 0);
    D0.w_off.assign(nblk, 0);
    int32_t rb = 0;
    int64_t wo = 0;
    for (size_t b = 0; b < nblk; ++b) {
      D0.row_begin[b] = rb;
      D0.nrows[b] = count[b];
      rb += count[b];
    }
    D0.rows.assign(static_cast<size_t>(rb), 0);
    D0.mask_col.assign(static_cast<size_t>(rb), 0);
    D0.mask_row.assign(static_cast<size_t>(rb), 0);
    D0.off_col.assign(static_cast<size_t>(rb), 0);
    D0.off_row.assign(static_cast<size_t>(rb), 0);
    // structure of W: column j is non-zero on the path from j to the root of its block
    for (int j = 0; j < m; ++j) {
      if (stage[j] != 0) continue;
      const int32_t bj = D0.row_begin[blk_id[j]], lj = loc[j];
      for (int v = j; v >= 0 && stage[v] == 0 && blk[v] == blk[j]; v = parent[v]) {
        D0.mask_col[bj + lj] |= 1ull << loc[v];
        D0.mask_row[bj + loc[v]] |= 1ull << lj;
      }
    }
    wo = 0;
    for (size_t b = 0; b < nblk; ++b) {
      D0.w_off[b] = wo;
      int32_t oc = 0, orw = 0;
      for (int l = 0; l < count[b]; ++l) {
        const size_t at = static_cast<size_t>(D0.row_begin[b]) + l;
        D0.off_col[at] = oc;
        D0.off_row[at] = orw;
        oc += __builtin_popcountll(D0.mask_col[at]);
        orw += __builtin_popcountll(D0.mask_row[at]);
      }
      wo += oc;  // == orw
    }
    D0.w_by_col.assign(static_cast<size_t>(wo), 0.0);
    D0.w_by_row.assign(static_cast<size_t>(wo), 0.0);
    // groups (the rotation rows of a pose) in adjacent lanes of one block: lets the kernel fuse row-unit work
    P.groups_whole = group != nullptr;
    if (group)
      for (int v = 0; v + 1 < m && P.groups_whole; ++v)
        if ((*group)[v] >= 0 && (*group)[v + 1] == (*group)[v] &&
            !(stage[v] == stage[v + 1] && (stage[v] != 0 || (blk_id[v] == blk_id[v + 1] && loc[v + 1] == loc[v] + 1))))
          P.groups_whole = false;
    D0.ext_ptr.assign(static_cast<size_t>(rb) + 1, 0);
    for (int v = 0; v < m; ++v)
      if (stage[v] == 0) {
        const int32_t at = D0.row_begin[blk_id[v]] + loc[v];
        D0.rows[at] = row_of[v];
        for (int32_t q = Lp[v] + 1; q < Lp[v + 1]; ++q)
          if (stage[Li[q]] > 0) D0.ext_ptr[at + 1]++;
      }
    for (int32_t i = 0; i < rb; ++i) D0.ext_ptr[i + 1] += D0.ext_ptr[i];
    D0.ext_col.assign(static_cast<size_t>(D0.ext_ptr[rb]), 0);
    D0.ext_val.assign(static_cast<size_t>(D0.ext_ptr[rb]), 0.0);
    for (int v = 0; v < m; ++v)
      if (stage[v] == 0) {
        int32_t at = D0.ext_ptr[D0.row_begin[blk_id[v]] + loc[v]];
        for (int32_t q = Lp[v] + 1; q < Lp[v + 1]; ++q)
          if (stage[Li[q]] > 0) {
            D0.ext_col[at] = row_of[Li[q]];
            D0.ext_val[at++] = -Lx[q];
          }
      }
  }
  tick("stages");
  // ---- "b" products: explicit inverse of every diagonal block.  Column j of W = L_bb^-1 solves
  // L_bb w = e_j and is non-zero only on the path from j to the root of its block.
  // With substitution blocks only the last stage has one, and nothing it reads changes from here on: it is computed on
  // a thread of its own while the blocks are built (12 ms of the plan at 10^5 poses).
  std::vector<std::vector<int32_t>> wt_row(static_cast<size_t>(K)), wt_col(static_cast<size_t>(K));
  std::vector<std::vector<double>> wt_val(static_cast<size_t>(K));
  std::vector<RowList> bb(static_cast<size_t>(K));
  auto explicit_inverses = [&]() {
    std::vector<double> w(static_cast<size_t>(m), 0.0);
    for (int j = 0; j < m; ++j) {
      const int k = stage[j], b = blk[j];
      if (k == 0 && sub0) continue;
      const bool dense = k == 0 && dense0;
      RowList &B = bb[k];
      if (!dense) B.begin_row(row_of[j]);  // backward "b": x_j = sum_i W_ij t_i  (column j of W)
      w[j] = 1.0;
      for (int v = j; v >= 0 && stage[v] == k && blk[v] == b; v = parent[v]) {
        const double wv = w[v] / Lx[Lp[v]];
        w[v] = 0.0;
        if (dense) {
          const int lj = loc[j], li = loc[v];
          const int64_t base = D0.w_off[blk_id[j]];
          const size_t rb0 = static_cast<size_t>(D0.row_begin[blk_id[j]]);
          D0.w_by_col[base + D0.off_col[rb0 + lj] + __builtin_popcountll(D0.mask_col[rb0 + lj] & ((1ull << li) - 1))] = wv;
          D0.w_by_row[base + D0.off_row[rb0 + li] + __builtin_popcountll(D0.mask_row[rb0 + li] & ((1ull << lj) - 1))] = wv;
          ++P.nnzW;
        } else {
          B.add(row_of[v], wv);
          wt_row[k].push_back(v);
          wt_col[k].push_back(j);
          wt_val[k].push_back(wv);
        }
        for (int32_t q = Lp[v] + 1; q < Lp[v + 1]; ++q) {
          const int i = Li[q];
          if (stage[i] == k && blk[i] == b) w[i] -= Lx[q] * wv;
          else if (sorted) break;  // ancestors outside the block are numbered after all of its rows
        }
      }
      if (!dense) B.end_row();
    }
    if (zero_row >= 0) {  // the pinned row rides along as an empty row of the last stage's backward product
      bb[K - 1].begin_row(zero_row);
      bb[K - 1].end_row();
    }
  };
  std::future<void> inverses_ready;
  if (sub0) inverses_ready = std::async(std::launch::async, explicit_inverses);
  struct JoinInverses {  // (an exception on the way must not leave the thread behind with references to this frame)
    std::future<void> &f;
    ~JoinInverses() { if (f.valid()) f.wait(); }
  } join_inverses{inverses_ready};
  // ---- stage 0 as workgroup blocks solved by substitution (trisolve.h, SubBlockOpHost)
  std::vector<std::vector<int32_t>> aux_of(sub0 ? static_cast<size_t>(m) : 0);  // later-stage variable -> its aux rows
  if (sub0) {
    SubBlockOpHost &SG = P.stages[0].sub_op;
    // blocks in order of their roots, members ascending (= elimination order)
    std::vector<int32_t> id_of_root(static_cast<size_t>(m), -1);
    std::vector<std::vector<int32_t>> members;
    for (int v = 0; v < m; ++v)
      if (stage[v] == 0 && blk[v] == v) {
        id_of_root[v] = static_cast<int32_t>(members.size());
        members.emplace_back();
      }
    for (int v = 0; v < m; ++v)
      if (stage[v] == 0) members[id_of_root[blk[v]]].push_back(v);
    std::vector<int32_t> flev(static_cast<size_t>(m), 0), blev(static_cast<size_t>(m), 0), li_of(static_cast<size_t>(m), -1);
    // lanes per task: the kernel is bound by instruction issue, so as few lanes (wavefronts) as the row lengths
    // allow -- up to kLaneEntries entries per lane
    auto lanes_for = [](int max_len) {
      int g = 1;
      while (g < 64 && (max_len + g - 1) / g > kLaneEntries) g <<= 1;
      return g;
    };
    // Blocks are independent: each one is built into a piece of its own (offsets relative to the piece), several
    // threads at a time, and the pieces are appended in block order afterwards.
    struct Piece {
      SubBlockOpHost S;
      std::vector<int32_t> tgt_var;  // later-stage variable of every target
      bool groups_ok = true;
    };
    std::vector<Piece> pieces(members.size());
    auto build_block = [&](size_t b) {
      SubBlockOpHost &S0 = pieces[b].S;
      std::vector<int32_t> &tgt_var = pieces[b].tgt_var;
      bool &groups_ok = pieces[b].groups_ok;
      const std::vector<int32_t> &mem = members[b];
      const int nb = static_cast<int>(mem.size());
      const int32_t root = mem.back();
      auto inside = [&](int u) { return stage[u] == 0 && blk[u] == root; };
      // ---- supernodes: runs of consecutive variables chained in the elimination tree (a pose: its rotation rows
      // and its translation), at most kSnCap rows.  A supernode is solved in ONE level with the explicit inverse W
      // of its small dense diagonal block folded into its rows:
      //   forward : y_i = sum_{q<=i} W_iq t_q - sum_j (sum_{q<=i} W_iq L_qj) y_j       (j: descendants in the block)
      //   backward: x_i = sum_{q>=i} W_qi t_q - sum_k (sum_{q>=i} W_qi L_kq) x_k       (k: ancestors in the block)
      // so every row is a plain list of (local row, coefficient) pairs over the block's tile -- siblings still hold
      // their right-hand side when a level reads them (read, barrier, write) -- and a block has ~10 levels, not ~32.
      std::vector<int32_t> sn_begin;  // positions in `mem`
      for (int t = 0; t < nb; ++t) {
        const int32_t v = mem[t];
        bool chained = t > 0 && mem[t - 1] == v - 1 && parent[v - 1] == v;
        if (chained) {
          // never cut inside a group (the d rotation rows of a pose share a supernode, hence sit at consecutive tile
          // positions in both sweeps): a group that would not fit starts a supernode of its own
          const int32_t gv = group ? (*group)[v] : -1;
          if (gv >= 0 && (*group)[v - 1] == gv) {
            chained = true;
          } else {
            int glen = 1;
            while (gv >= 0 && t + glen < nb && mem[t + glen] == v + glen && (*group)[v + glen] == gv) ++glen;
            chained = t - sn_begin.back() + glen <= kSnCap;
          }
        }
        if (!chained) sn_begin.push_back(t);
      }
      sn_begin.push_back(nb);
      const int nsn = static_cast<int>(sn_begin.size()) - 1;
      std::vector<int32_t> sn_id(static_cast<size_t>(nb));
      for (int sidx = 0; sidx < nsn; ++sidx)
        for (int t = sn_begin[sidx]; t < sn_begin[sidx + 1]; ++t) sn_id[t] = sidx;
      using Ent = std::pair<int32_t, double>;  // (variable, coefficient)
      // later-stage variables coupled to the block ("targets"), sorted; the backward sweep finds their solution in
      // the rows nb + k of its tile (pseudo-variable m + k in the entry lists below)
      for (int32_t v : mem)
        for (int32_t q = Lp[v] + 1; q < Lp[v + 1]; ++q)
          if (!inside(Li[q])) tgt_var.push_back(Li[q]);
      std::sort(tgt_var.begin(), tgt_var.end());
      tgt_var.erase(std::unique(tgt_var.begin(), tgt_var.end()), tgt_var.end());
      const int ntg = static_cast<int>(tgt_var.size());
      auto tgt_of = [&](int32_t var) { return static_cast<int32_t>(std::lower_bound(tgt_var.begin(), tgt_var.end(), var) - tgt_var.begin()); };
      std::vector<std::vector<Ent>> frow(static_cast<size_t>(nb)), brow(static_cast<size_t>(nb));
      auto add_to = [](std::vector<Ent> &row, int32_t var, double val) {
        for (Ent &e : row)
          if (e.first == var) { e.second += val; return; }
        row.push_back({var, val});
      };
      std::vector<std::vector<double>> Wsn(static_cast<size_t>(nsn));
      int nfl = 0, nbl = 0;
      for (int sidx = 0; sidx < nsn; ++sidx) {
        const int t0 = sn_begin[sidx], sz = sn_begin[sidx + 1] - t0;
        const int32_t v0 = mem[t0];
        // dense diagonal block and its inverse (row-major sz x sz, lower triangular)
        std::vector<double> Ld(static_cast<size_t>(sz) * sz, 0.0), &W = Wsn[sidx];
        for (int i = 0; i < sz; ++i) {
          Ld[i * sz + i] = Lx[Lp[v0 + i]];
          for (int32_t q = rptr[v0 + i]; q < rptr[v0 + i + 1]; ++q)
            if (rcol[q] >= v0) Ld[i * sz + (rcol[q] - v0)] = rval[q];
        }
        W.assign(static_cast<size_t>(sz) * sz, 0.0);
        for (int c = 0; c < sz; ++c)
          for (int i = c; i < sz; ++i) {
            double sacc = i == c ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) sacc -= Ld[i * sz + k] * W[k * sz + c];
            W[i * sz + c] = sacc / Ld[i * sz + i];
          }
        int lev = 0;
        for (int i = 0; i < sz; ++i) {
          std::vector<Ent> &row = frow[t0 + i];
          for (int q = 0; q <= i; ++q) {
            row.push_back({v0 + q, W[i * sz + q]});
            for (int32_t e = rptr[v0 + q]; e < rptr[v0 + q + 1]; ++e)
              if (rcol[e] < v0 && inside(rcol[e])) {
                add_to(row, rcol[e], -W[i * sz + q] * rval[e]);
                lev = std::max(lev, flev[rcol[e]] + 1);
              }
          }
        }
        for (int i = 0; i < sz; ++i) flev[v0 + i] = lev;
        nfl = std::max(nfl, lev + 1);
      }
      for (int sidx = nsn - 1; sidx >= 0; --sidx) {
        const int t0 = sn_begin[sidx], sz = sn_begin[sidx + 1] - t0;
        const int32_t v0 = mem[t0], vend = v0 + sz;
        const std::vector<double> &W = Wsn[sidx];
        int lev = 0;
        for (int i = 0; i < sz; ++i) {
          std::vector<Ent> &row = brow[t0 + i];
          for (int q = i; q < sz; ++q) {
            row.push_back({v0 + q, W[q * sz + i]});
            for (int32_t e = Lp[v0 + q] + 1; e < Lp[v0 + q + 1]; ++e)
              if (Li[e] >= vend && inside(Li[e])) {
                add_to(row, Li[e], -W[q * sz + i] * Lx[e]);
                lev = std::max(lev, blev[Li[e]] + 1);
              } else if (Li[e] >= vend) {  // coupling to the later stage: reads the staged row of the target
                add_to(row, m + tgt_of(Li[e]), -W[q * sz + i] * Lx[e]);
              }
          }
        }
        for (int i = 0; i < sz; ++i) blev[v0 + i] = lev;
        nbl = std::max(nbl, lev + 1);
      }
      // positions of `mem` in forward / backward level order; inside a level the supernodes (kept whole) are sorted by
      // their longest row, so that rows of similar length share a chunk of the level (chunks are padded to one width)
      std::vector<int32_t> ford(static_cast<size_t>(nb)), bord(static_cast<size_t>(nb));
      {
        std::vector<int32_t> fmax(static_cast<size_t>(nsn), 0), bmax(static_cast<size_t>(nsn), 0), so(static_cast<size_t>(nsn));
        for (int t = 0; t < nb; ++t) {
          fmax[sn_id[t]] = std::max<int32_t>(fmax[sn_id[t]], static_cast<int32_t>(frow[t].size()));
          bmax[sn_id[t]] = std::max<int32_t>(bmax[sn_id[t]], static_cast<int32_t>(brow[t].size()));
        }
        auto order_by = [&](const std::vector<int32_t> &lev, const std::vector<int32_t> &mx, std::vector<int32_t> &out) {
          for (int i = 0; i < nsn; ++i) so[i] = i;
          std::stable_sort(so.begin(), so.end(), [&](int32_t x, int32_t y) {
            const int lx = lev[mem[sn_begin[x]]], ly = lev[mem[sn_begin[y]]];
            return lx != ly ? lx < ly : mx[x] > mx[y];
          });
          int at = 0;
          for (int32_t i : so)
            for (int t = sn_begin[i]; t < sn_begin[i + 1]; ++t) out[at++] = t;
        };
        order_by(flev, fmax, ford);
        order_by(blev, bmax, bord);
      }
      std::vector<int32_t> bpos_of(static_cast<size_t>(nb));  // backward position of mem position
      for (int t = 0; t < nb; ++t) {
        li_of[mem[ford[t]]] = t;
        bpos_of[bord[t]] = t;
      }
      std::vector<int32_t> mempos_of_var;  // variable -> position in mem (variables of a block are looked up by search)
      auto mem_pos = [&](int32_t var) { return static_cast<int32_t>(std::lower_bound(mem.begin(), mem.end(), var) - mem.begin()); };
      S0.row_begin.push_back(static_cast<int32_t>(S0.rows.size()));
      S0.nrows.push_back(nb);
      S0.f_ent_begin.push_back(static_cast<int32_t>(S0.f_val.size()));
      S0.b_ent_begin.push_back(static_cast<int32_t>(S0.b_val.size()));
      S0.f_lev_begin.push_back(0);  // set below
      S0.b_lev_begin.push_back(0);
      S0.tgt_begin.push_back(static_cast<int32_t>(S0.tgt_slot.size()));
      // Emits one sweep: rows in `order` (positions of mem), their entries translated to the sweep's local numbering.
      // A level is cut into chunks (levels of their own for the kernel) of rows x lanes per task <= kLevelLanes, never
      // inside a supernode; every row of a chunk is padded with null entries to the chunk's width g * npl (npl <=
      // kLaneEntries per lane), so a lane finds its entries by arithmetic and the kernel's inner loop has no predicates.
      // Header of chunk c: {first row, g, npl, first entry}; one more header closes the list.
      auto emit = [&](const std::vector<int32_t> &order, const std::vector<int32_t> &lev_of_var,
                      std::vector<std::vector<Ent>> &rows_ent, bool backward, std::vector<uint16_t> &idx,
                      std::vector<double> &val, std::vector<int32_t> &hdr, int32_t ent0) {
        for (int k = 0; k < nb; ++k) {
          std::vector<Ent> &row = rows_ent[order[k]];
          std::sort(row.begin(), row.end(), [](const Ent &x, const Ent &y) { return x.first < y.first; });
        }
        int t = 0;
        while (t < nb) {
          const int lev = lev_of_var[mem[order[t]]];
          int t1 = t;
          while (t1 < nb && lev_of_var[mem[order[t1]]] == lev) ++t1;
          if (timing) {  // what a layout with a (g, npl) of its own per WAVEFRONT of a level would read: rows grouped by class
            // rows by length, longest first; a wavefront takes rows while they fit its 64 lanes at the (g, npl) of its first row
            std::vector<int> lens;
            int64_t real = 0;
            for (int q = t; q < t1; ++q) {
              lens.push_back(std::max<int>(1, static_cast<int>(rows_ent[order[q]].size())));
              real += lens.back();
            }
            std::sort(lens.begin(), lens.end(), std::greater<int>());
            int64_t waves = 0, reads = 0;
            for (size_t q = 0; q < lens.size();) {
              const int g = lanes_for(lens[q]), npl = (lens[q] + g - 1) / g;
              q += static_cast<size_t>(64 / g);
              ++waves;
              reads += 64 * npl;
            }
            g_seg_waves += waves; g_seg_reads += reads; g_seg_real += real; g_seg_levels += 1; g_seg_sublevels += (waves + 3) / 4;
          }
          // One barrier level = kSubWaves wavefronts, each with a (lanes per row, entries per lane) pair of its own and
          // a header of its own: a wavefront takes consecutive rows of the (longest-first) order while they fit its 64
          // lanes at the width of its first row -- short rows no longer pay the width of the level's longest, and a level
          // that needs more than kSubWaves wavefronts continues in the next barrier level.
          // The rows of a SUPERNODE read each other's right-hand sides, so all of them sit in ONE barrier level: a wavefront
          // takes whole supernodes; one that is too long for a wavefront (rows of 32 or 64 lanes) spreads over consecutive
          // wavefronts of the same barrier level.
          int nw = 0;
          auto idle_wave = [&](int row) {
            hdr.insert(hdr.end(), {row, 1, static_cast<int32_t>(val.size()) - ent0, static_cast<int32_t>(idx.size())});
            ++nw;
          };
          auto sn_end = [&](int q) {  // one past the last row of the supernode that row q (of the order) belongs to
            int e = q;
            while (e < t1 && sn_id[order[e]] == sn_id[order[q]]) ++e;
            return e;
          };
          for (int c0 = t; c0 < t1;) {
            // lanes per row: from the longest row of the first supernode (supernodes come longest first)
            const int s1 = sn_end(c0);
            int max_len = 1;
            for (int q = c0; q < s1; ++q) max_len = std::max<int>(max_len, static_cast<int>(rows_ent[order[q]].size()));
            const int g = lanes_for(max_len), cap = kWaveLanes / g;
            int c1;
            if (s1 - c0 > cap) {  // the supernode alone needs several wavefronts: all in this barrier level
              const int need = (s1 - c0 + cap - 1) / cap;
              if (need > kSubWaves) throw std::logic_error("cora: a supernode does not fit one barrier level");
              if (nw % kSubWaves + need > kSubWaves)
                while (nw % kSubWaves) idle_wave(c0);
              c1 = c0 + cap;  // (the next turns of the loop take the rest: same g, same barrier level)
            } else if (c0 > t && sn_id[order[c0 - 1]] == sn_id[order[c0]]) {
              c1 = s1;  // the rest of a supernode that spreads over wavefronts: nothing else joins it (its rows may be shorter than the next supernode's)
            } else {
              c1 = s1;
              while (c1 < t1) {  // whole supernodes while they fit
                const int e = sn_end(c1);
                if (e - c0 > cap) break;
                c1 = e;
              }
            }
            int npl = 1;
            for (int q = c0; q < c1; ++q) {
              const int len = std::max<int>(1, static_cast<int>(rows_ent[order[q]].size()));
              if (lanes_for(len) > g) throw std::logic_error("cora: rows of a level are not ordered by length");
              npl = std::max(npl, (len + g - 1) / g);
            }
            // header {first row, g | npl << 8 | rows << 12, first coefficient (block-relative), first index (absolute in the idx array)}
            hdr.insert(hdr.end(), {c0, g | (npl << 8) | ((c1 - c0) << 12), static_cast<int32_t>(val.size()) - ent0, static_cast<int32_t>(idx.size())});
            ++nw;
            if (timing) { g_cur_reads += static_cast<int64_t>(kWaveLanes) * npl; }
            // lane p of row k takes the row's entries p, p + g, ...  Coefficients of the wavefront: slot-major,
            // [u][lane = (k - c0) * g + p] (the kernel streams them, one coalesced load per slot); local row indices:
            // lane-major, [lane][4 or 8] (one load per lane), padded to a multiple of 8 indices
            S0.max_level_lanes = std::max<int32_t>(S0.max_level_lanes, (c1 - c0) * g);
            S0.max_npl = std::max<int32_t>(S0.max_npl, npl);
            const int istride = npl <= 4 ? 4 : 8;
            for (int u = 0; u < npl; ++u)
              for (int k = c0; k < c1; ++k) {
                const std::vector<Ent> &row = rows_ent[order[k]];
                for (int p = 0; p < g; ++p) {
                  const int e = p + u * g;
                  val.push_back(e < static_cast<int>(row.size()) ? row[e].second : 0.0);
                }
              }
            for (int k = c0; k < c1; ++k) {
              const std::vector<Ent> &row = rows_ent[order[k]];
              for (int p = 0; p < g; ++p)
                for (int u = 0; u < istride; ++u) {
                  const int e = p + u * g;
                  const bool real = u < npl && e < static_cast<int>(row.size());
                  idx.push_back(real ? static_cast<uint16_t>(row[e].first >= m ? nb + (row[e].first - m) : backward ? bpos_of[mem_pos(row[e].first)] : li_of[row[e].first]) : uint16_t(0));
                }
            }
            while (idx.size() % 8) idx.push_back(0);
            c0 = c1;
          }
          while (nw % kSubWaves) idle_wave(t1);  // wavefronts without rows in the level's last barrier level
          if (timing) g_cur_sublevels += nw / kSubWaves;
          t = t1;
        }
        for (int w = 0; w < kSubWaves; ++w)  // the closing level: no rows
          hdr.insert(hdr.end(), {nb, 1, static_cast<int32_t>(val.size()) - ent0, static_cast<int32_t>(idx.size())});
      };
      const int32_t fe0 = static_cast<int32_t>(S0.f_val.size()), be0 = static_cast<int32_t>(S0.b_val.size());
      S0.f_lev_begin.back() = static_cast<int32_t>(S0.f_hdr.size() / 4);
      S0.b_lev_begin.back() = static_cast<int32_t>(S0.b_hdr.size() / 4);
      emit(ford, flev, frow, false, S0.f_idx, S0.f_val, S0.f_hdr, fe0);
      emit(bord, blev, brow, true, S0.b_idx, S0.b_val, S0.b_hdr, be0);
      S0.f_nent.push_back(static_cast<int32_t>(S0.f_val.size()) - fe0);
      S0.b_nent.push_back(static_cast<int32_t>(S0.b_val.size()) - be0);
      S0.max_lev = std::max<int32_t>(S0.max_lev, std::max<int32_t>(static_cast<int32_t>(S0.f_hdr.size() / 4) - S0.f_lev_begin.back(),
                                                                    static_cast<int32_t>(S0.b_hdr.size() / 4) - S0.b_lev_begin.back()));
      for (int k = 0; k < nb; ++k) {
        S0.rows.push_back(row_of[mem[ford[k]]]);
        const int32_t v = mem[bord[k]];
        S0.b_rows.push_back(row_of[v]);
      }
      (void)nfl;
      (void)nbl;
      (void)mempos_of_var;
      // forward contributions: one target per later-stage row coupled to the block, entries by local row
      std::vector<std::pair<int32_t, std::pair<int32_t, double>>> trip;  // (target variable, (local row, -L))
      for (int32_t v : mem)
        for (int32_t q = Lp[v] + 1; q < Lp[v + 1]; ++q)
          if (!inside(Li[q])) trip.push_back({Li[q], {li_of[v], -Lx[q]}});
      std::sort(trip.begin(), trip.end(), [](const auto &a, const auto &c) {
        return a.first != c.first ? a.first < c.first : a.second.first < c.second.first;
      });
      for (size_t t = 0; t < trip.size(); ++t) {
        if (t == 0 || trip[t].first != trip[t - 1].first) {
          if (t > 0) S0.c_ptr.push_back(static_cast<int32_t>(S0.c_idx.size()));
          S0.tgt_slot.push_back(S0.n_aux++);
          S0.tgt_row.push_back(row_of[trip[t].first]);
        }
        S0.c_idx.push_back(static_cast<uint16_t>(trip[t].second.first));
        S0.c_val.push_back(trip[t].second.second);
      }
      if (!trip.empty()) S0.c_ptr.push_back(static_cast<int32_t>(S0.c_idx.size()));
      S0.max_rows = std::max(S0.max_rows, nb + ntg);
      S0.max_ent = std::max<int32_t>(S0.max_ent, std::max<int32_t>(static_cast<int32_t>(S0.f_idx.size()) - fe0,
                                                                    static_cast<int32_t>(S0.b_idx.size()) - be0));
    };
    {
      const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
      size_t nth = std::min<size_t>(std::min<size_t>(hw, 48), std::max<size_t>(members.size() / 8, 1));
      if (const char *e = std::getenv("CORA_TRI_THREADS")) nth = std::max(1, std::atoi(e));
      cora::parallel_parts(static_cast<unsigned>(nth), [&](unsigned t) {
        for (size_t b = t; b < members.size(); b += nth) build_block(b);
      });
    }
    tick("blocks (threads)");
    SubBlockOpHost &S0 = SG;
    // The per-block pieces go into one array each.  Where every piece lands follows from the sizes (prefix sums); the
    // 130 MB of entries are then sized by a few threads (one array each: fresh pages) and copied by all of them (one
    // range of pieces each) -- appended one after the other this was a third of the plan's time.
    const size_t np = pieces.size();
    struct Off { size_t rows, brows, tgt, fh, bh, fi, bi, fv, bv, cp, ci; };
    std::vector<Off> off(np + 1);
    off[0] = Off{0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0};  // (c_ptr starts with its leading 0)
    for (size_t b = 0; b < np; ++b) {
      const SubBlockOpHost &Pc = pieces[b].S;
      const Off &o = off[b];
      off[b + 1] = Off{o.rows + Pc.rows.size(), o.brows + Pc.b_rows.size(), o.tgt + Pc.tgt_row.size(), o.fh + Pc.f_hdr.size(),
                       o.bh + Pc.b_hdr.size(), o.fi + Pc.f_idx.size(), o.bi + Pc.b_idx.size(), o.fv + Pc.f_val.size(),
                       o.bv + Pc.b_val.size(), o.cp + Pc.c_ptr.size(), o.ci + Pc.c_idx.size()};
    }
    const Off &tot = off[np];
    {
      // (+ 8 of capacity: the upload pads these arrays for the kernel's read-ahead; without the room that is a copy of
      // 40 MB each)
      std::vector<std::function<void()>> sizing = {
          [&] { S0.f_val.reserve(tot.fv + 8); S0.f_val.resize(tot.fv); },
          [&] { S0.b_val.reserve(tot.bv + 8); S0.b_val.resize(tot.bv); },
          [&] { S0.f_idx.reserve(tot.fi + 8); S0.f_idx.resize(tot.fi); },
          [&] { S0.b_idx.reserve(tot.bi + 8); S0.b_idx.resize(tot.bi); },
          [&] { S0.f_hdr.reserve(tot.fh + 8); S0.f_hdr.resize(tot.fh); S0.b_hdr.reserve(tot.bh + 8); S0.b_hdr.resize(tot.bh); },
          [&] { S0.c_idx.resize(tot.ci); S0.c_val.resize(tot.ci); S0.c_ptr.resize(tot.cp); S0.c_ptr[0] = 0; },
          [&] { S0.rows.resize(tot.rows); S0.b_rows.resize(tot.brows); S0.tgt_row.resize(tot.tgt); S0.tgt_slot.resize(tot.tgt); }};
      cora::parallel_parts(static_cast<unsigned>(sizing.size()), [&](unsigned t) { sizing[t](); });
    }
    // small per-block records and the aux slots, in block order
    S0.row_begin.reserve(np), S0.nrows.reserve(np), S0.f_ent_begin.reserve(np), S0.b_ent_begin.reserve(np);
    S0.f_nent.reserve(np), S0.b_nent.reserve(np), S0.f_lev_begin.reserve(np + 1), S0.b_lev_begin.reserve(np + 1), S0.tgt_begin.reserve(np + 1);
    for (size_t b = 0; b < np; ++b) {
      const SubBlockOpHost &Pc = pieces[b].S;
      const Off &o = off[b];
      if (!pieces[b].groups_ok) P.groups_whole = false;
      S0.row_begin.push_back(static_cast<int32_t>(o.rows));
      S0.nrows.push_back(Pc.nrows[0]);
      S0.f_ent_begin.push_back(static_cast<int32_t>(o.fv));
      S0.b_ent_begin.push_back(static_cast<int32_t>(o.bv));
      S0.f_nent.push_back(Pc.f_nent[0]);
      S0.b_nent.push_back(Pc.b_nent[0]);
      S0.f_lev_begin.push_back(static_cast<int32_t>(o.fh / 4));
      S0.b_lev_begin.push_back(static_cast<int32_t>(o.bh / 4));
      S0.tgt_begin.push_back(static_cast<int32_t>(o.tgt));
      for (size_t t = 0; t < Pc.tgt_slot.size(); ++t) {
        aux_of[pieces[b].tgt_var[t]].push_back(S0.n_aux);
        S0.tgt_slot[o.tgt + t] = S0.n_aux++;
      }
      S0.max_rows = std::max(S0.max_rows, Pc.max_rows);
      S0.max_ent = std::max(S0.max_ent, Pc.max_ent);
      S0.max_lev = std::max(S0.max_lev, Pc.max_lev);
      S0.max_level_lanes = std::max(S0.max_level_lanes, Pc.max_level_lanes);
      S0.max_npl = std::max(S0.max_npl, Pc.max_npl);
    }
    {
      const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
      const size_t nth = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw, 32), np / 8));
      cora::parallel_parts(static_cast<unsigned>(nth), [&](unsigned t) {
          auto put = [](auto &dst, size_t at, const auto &src) { std::copy(src.begin(), src.end(), dst.begin() + static_cast<std::ptrdiff_t>(at)); };
          for (size_t b = np * t / nth; b < np * (t + 1) / nth; ++b) {
            SubBlockOpHost &Pc = pieces[b].S;
            const Off &o = off[b];
            put(S0.rows, o.rows, Pc.rows);
            put(S0.b_rows, o.brows, Pc.b_rows);
            put(S0.tgt_row, o.tgt, Pc.tgt_row);
            for (size_t q = 3; q < Pc.f_hdr.size(); q += 4) Pc.f_hdr[q] += static_cast<int32_t>(o.fi);  // indices: absolute
            for (size_t q = 3; q < Pc.b_hdr.size(); q += 4) Pc.b_hdr[q] += static_cast<int32_t>(o.bi);
            put(S0.f_hdr, o.fh, Pc.f_hdr);
            put(S0.b_hdr, o.bh, Pc.b_hdr);
            put(S0.f_idx, o.fi, Pc.f_idx);
            put(S0.b_idx, o.bi, Pc.b_idx);
            put(S0.f_val, o.fv, Pc.f_val);
            put(S0.b_val, o.bv, Pc.b_val);
            for (size_t q = 0; q < Pc.c_ptr.size(); ++q) S0.c_ptr[o.cp + q] = static_cast<int32_t>(o.ci) + Pc.c_ptr[q];
            put(S0.c_idx, o.ci, Pc.c_idx);
            put(S0.c_val, o.ci, Pc.c_val);
            pieces[b] = Piece();
          }
        });
    }
    S0.tgt_begin.push_back(static_cast<int32_t>(S0.tgt_slot.size()));
    S0.f_lev_begin.push_back(static_cast<int32_t>(S0.f_hdr.size() / 4));
    S0.b_lev_begin.push_back(static_cast<int32_t>(S0.b_hdr.size() / 4));
    for (int v = 0; v < m; ++v)
      if (stage[v] == 1) P.top_rows.push_back(row_of[v]);
    if (zero_row >= 0) P.top_rows.push_back(zero_row);
  }
  tick("merge / dense blocks");
  if (timing && sub0) {
    std::fprintf(stderr, "  [tri plan] tile reads per lane-slot, both sweeps: now %lld in %lld barrier levels; a (g, npl) per wavefront: %lld in %lld (tree levels %lld); real entries %lld\n",
                 static_cast<long long>(g_cur_reads.load()), static_cast<long long>(g_cur_sublevels.load()), static_cast<long long>(g_seg_reads.load()),
                 static_cast<long long>(g_seg_sublevels.load()), static_cast<long long>(g_seg_levels.load()), static_cast<long long>(g_seg_real.load()));
    g_seg_waves = g_seg_reads = g_seg_real = g_seg_levels = g_seg_sublevels = g_cur_reads = g_cur_sublevels = 0;  // shape of the substitution blocks: levels (= dependent steps of a sweep) and how full they are
    const SubBlockOpHost &S0 = P.stages[0].sub_op;
    const size_t nbk = S0.nrows.size();
    int64_t fl = 0, bl = 0, lanes_f = 0, lanes_b = 0, rows = 0, ent_f = 0, ent_b = 0, slots_f = 0, slots_b = 0;
    int fmax = 0, bmax = 0;
    for (size_t b = 0; b < nbk; ++b) {
      const int nf = S0.f_lev_begin[b + 1] - S0.f_lev_begin[b] - kSubWaves, nbw = S0.b_lev_begin[b + 1] - S0.b_lev_begin[b] - kSubWaves;  // headers
      fl += nf / kSubWaves; bl += nbw / kSubWaves; fmax = std::max(fmax, nf / kSubWaves); bmax = std::max(bmax, nbw / kSubWaves);
      rows += S0.nrows[b];
      for (int l = 0; l < nf; ++l) {
        const int32_t *h = &S0.f_hdr[4 * (static_cast<size_t>(S0.f_lev_begin[b]) + l)];
        const int g = h[1] & 0xff, npl = (h[1] >> 8) & 0xf, nr = h[1] >> 12;
        lanes_f += static_cast<int64_t>(nr) * g; slots_f += static_cast<int64_t>(nr) * g * npl;
      }
      for (int l = 0; l < nbw; ++l) {
        const int32_t *h = &S0.b_hdr[4 * (static_cast<size_t>(S0.b_lev_begin[b]) + l)];
        const int g = h[1] & 0xff, npl = (h[1] >> 8) & 0xf, nr = h[1] >> 12;
        lanes_b += static_cast<int64_t>(nr) * g; slots_b += static_cast<int64_t>(nr) * g * npl;
      }
      ent_f += S0.f_nent[b]; ent_b += S0.b_nent[b];
    }
    std::fprintf(stderr, "  [tri plan] %zu substitution blocks, %.0f rows each: forward %.1f levels per block (max %d), %.0f lanes and %.0f entry slots per level; "
                 "backward %.1f levels (max %d), %.0f lanes and %.0f slots per level; stored entries %lld + %lld\n",
                 nbk, double(rows) / nbk, double(fl) / nbk, fmax, double(lanes_f) / std::max<int64_t>(fl, 1), double(slots_f) / std::max<int64_t>(fl, 1),
                 double(bl) / nbk, bmax, double(lanes_b) / std::max<int64_t>(bl, 1), double(slots_b) / std::max<int64_t>(bl, 1),
                 static_cast<long long>(ent_f), static_cast<long long>(ent_b));
  }
  // ---- "a" products: the couplings between stages
  for (int k = 0; k < K && !sub0; ++k) {
    RowList fa, ba;
    for (int i = 0; i < m; ++i) {
      if (stage[i] != k) continue;
      if (k > 0) {  // forward: row i of L restricted to earlier stages
        fa.begin_row(row_of[i]);
        for (int32_t q = rptr[i]; q < rptr[i + 1]; ++q) {
          if (stage[rcol[q]] > k) throw std::logic_error("cora: stage order violates the elimination tree");
          if (stage[rcol[q]] < k) fa.add(row_of[rcol[q]], -rval[q]);
        }
        fa.end_row();
      }
      if (k < K - 1 && !(k == 0 && dense0)) {  // backward: column i of L restricted to later stages
        ba.begin_row(row_of[i]);
        for (int32_t q = Lp[i] + 1; q < Lp[i + 1]; ++q)
          if (stage[Li[q]] > k) ba.add(row_of[Li[q]], -Lx[q]);
        ba.end_row();
      }
    }
    if (k > 0) finalize(fa, P.stages[k].fwd_a);
    if (k < K - 1 && !(k == 0 && dense0)) finalize(ba, P.stages[k].bwd_a);
  }
  tick("a products");
  if (inverses_ready.valid()) inverses_ready.get();
  else explicit_inverses();
  for (int k = 0; k < K; ++k) {
    if (k == 0 && (dense0 || sub0)) continue;
    finalize(bb[k], P.stages[k].bwd_b);
    bb[k] = RowList();
    // forward "b": y_i = sum_j W_ij t_j  (row i of W): bucket the triplets by row
    const size_t nz = wt_row[k].size();
    P.nnzW += static_cast<int64_t>(nz);
    std::vector<int32_t> cnt(static_cast<size_t>(m) + 1, 0);
    for (size_t t = 0; t < nz; ++t) cnt[wt_row[k][t] + 1]++;
    for (int i = 0; i < m; ++i) cnt[i + 1] += cnt[i];
    std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1), cc(nz);
    std::vector<double> vv(nz);
    for (size_t t = 0; t < nz; ++t) {
      const int32_t at = pos[wt_row[k][t]]++;
      cc[at] = wt_col[k][t];
      vv[at] = wt_val[k][t];
    }
    // two-stage form: t_j = rhs_j + the aux rows the stage-0 blocks next to j wrote (row order: fixed).  Either the sum is
    // folded into the product (every entry W_ij repeated once per aux row of j: one launch less, what small top stages
    // want) or it is its own product in the slot of the unused "a" product (t_1 += sum of its aux rows, in place) --
    // at 10^6 poses the folded product had 3.3 entries per entry of W and took 203 us against 33 us for W^T.
    bool fold = true;
    if (sub0) {
      int64_t extra = 0;
      for (size_t t = 0; t < nz; ++t) extra += static_cast<int64_t>(aux_of[wt_col[k][t]].size());
      int64_t unfold_min = 2000000;  // extra entries that outweigh a launch AND the sum's own chain of latencies (measured: folded wins at 10^4 and 10^5 poses -- 7.7 vs 10.8 us, 13.6 vs 19.7 us --, loses at 10^6: 203 vs 99 us)
      if (const char *e = std::getenv("CORA_TRI_UNFOLD_MIN")) unfold_min = std::atoll(e);
      fold = extra < unfold_min;
      if (timing) std::fprintf(stderr, "  [tri plan] top stage: %lld entries, %lld more with the aux sums folded in: %s\n",
                               static_cast<long long>(nz), static_cast<long long>(extra), fold ? "folded" : "separate sum");
    }
    RowList F;
    for (int i = 0; i < m; ++i) {
      if (stage[i] != k) continue;
      F.begin_row(row_of[i]);
      for (int32_t q = cnt[i]; q < cnt[i + 1]; ++q) {
        F.add(row_of[cc[q]], vv[q]);
#ifdef CORA_LAB_BUILD
        // LAB (wrong results): upper bounds of two forms of the folded product that were priced before building --
        //   CORA_LAB_DROP_PAIR_AUX : columns with at most two aux rows (the separators) gather none of them: what a grouped gather
        //                            of [b | aux | aux] could gain at best (round 6, step 8: 13.0 -> 9.7 us);
        //   CORA_LAB_AUX_CAP=n     : columns with more than n aux rows (the landmarks: one per solve block) gather n of them: what a
        //                            hierarchical reduction of the landmark slots inside the forward sweep (the round-5 review's
        //                            item 1a: "the last stage sees 8 partials per row") could gain at best.
        static const bool lab_drop = std::getenv("CORA_LAB_DROP_PAIR_AUX") != nullptr;
        static const int lab_cap = std::getenv("CORA_LAB_AUX_CAP") ? std::atoi(std::getenv("CORA_LAB_AUX_CAP")) : 0;
        if (lab_drop && aux_of[cc[q]].size() <= 2) continue;
        if (lab_cap > 0 && sub0 && fold && static_cast<int>(aux_of[cc[q]].size()) > lab_cap) {
          for (int a = 0; a < lab_cap; ++a) F.add(aux_base + aux_of[cc[q]][a], vv[q]);
          continue;
        }
#endif
        if (sub0 && fold)
          for (int32_t a : aux_of[cc[q]]) F.add(aux_base + a, vv[q]);
      }
      F.end_row();
    }
    finalize(F, P.stages[k].fwd_b);
    if (sub0 && !fold) {
      RowList A;
      for (int i = 0; i < m; ++i) {
        if (stage[i] != k || aux_of[i].empty()) continue;
        A.begin_row(row_of[i]);
        for (int32_t a : aux_of[i]) A.add(aux_base + a, 1.0);
        A.end_row();
      }
      finalize(A, P.stages[k].fwd_a);
    }
    wt_row[k] = std::vector<int32_t>();
    wt_col[k] = std::vector<int32_t>();
    wt_val[k] = std::vector<double>();
  }
  tick("b products");
  if (timing)  // shape of every product: rows per length class and the longest row of each
    for (int k = 0; k < K; ++k)
      for (const auto &pr : {std::make_pair("fwd_a", &P.stages[k].fwd_a), std::make_pair("fwd_b", &P.stages[k].fwd_b),
                             std::make_pair("bwd_a", &P.stages[k].bwd_a), std::make_pair("bwd_b", &P.stages[k].bwd_b)}) {
        const RowOpHost &op = *pr.second;
        if (op.empty()) continue;
        int64_t e8 = 0, e64 = 0, m8 = 0, m64 = 0;
        for (int r = 0; r < op.n8 + op.n64; ++r) {
          const int64_t len = op.end[r] - op.begin[r];
          (r < op.n8 ? e8 : e64) += len;
          (r < op.n8 ? m8 : m64) = std::max(r < op.n8 ? m8 : m64, len);
        }
        std::fprintf(stderr, "  [tri plan] stage %d %s: %d rows of 8 lanes (%lld entries, longest %lld), %d wavefront rows (%lld, longest %lld), %zu long rows in %zu chunks\n",
                     k, pr.first, op.n8, static_cast<long long>(e8), static_cast<long long>(m8), op.n64, static_cast<long long>(e64),
                     static_cast<long long>(m64), op.long_out.size(), op.chunk_begin.size());
        if (!op.long_out.empty()) {  // chunks per long row
          std::fprintf(stderr, "  [tri plan]   chunks per long row:");
          for (size_t r = 0; r + 1 < op.long_chunk_ptr.size(); ++r) std::fprintf(stderr, " %d", op.long_chunk_ptr[r + 1] - op.long_chunk_ptr[r]);
          std::fprintf(stderr, "\n");
        }
      }
}


// ---- test hook: the staged products executed on the host, in the order factor_solve launches them
namespace {
void apply_rowop(const RowOpHost &op, const double *src0, const double *src, double *dst) {
  const int n = op.n8 + op.n64;
  for (int r = 0; r < n; ++r) {
    double s = src0 ? src0[op.out_row[r]] : 0.0;
    for (int32_t k = op.begin[r]; k < op.end[r]; ++k) s += op.val[k] * src[op.col[k]];
    dst[op.out_row[r]] = s;
  }
  for (size_t r = 0; r < op.long_out.size(); ++r) {
    double s = src0 ? src0[op.long_out[r]] : 0.0;
    for (int32_t ch = op.long_chunk_ptr[r]; ch < op.long_chunk_ptr[r + 1]; ++ch)
      for (int32_t k = op.chunk_begin[ch]; k < op.chunk_end[ch]; ++k) s += op.val[k] * src[op.col[k]];
    dst[op.long_out[r]] = s;
  }
}
void apply_blocks(const BlockOpHost &B, bool bwd, const double *src, double *dst) {
  std::vector<double> t, acc;
  for (size_t b = 0; b < B.nrows.size(); ++b) {
    const int nb = B.nrows[b], rb = B.row_begin[b];
    t.assign(nb, 0.0);
    acc.assign(nb, 0.0);
    for (int l = 0; l < nb; ++l) {
      t[l] = src[B.rows[rb + l]];
      if (bwd)
        for (int32_t k = B.ext_ptr[rb + l]; k < B.ext_ptr[rb + l + 1]; ++k) t[l] += B.ext_val[k] * src[B.ext_col[k]];
    }
    const double *W = (bwd ? B.w_by_row.data() : B.w_by_col.data()) + B.w_off[b];
    for (int q = 0; q < nb; ++q) {  // the kernel's loop: lane l takes entry off + popcount(mask below l)
      const uint64_t mask = bwd ? B.mask_row[rb + q] : B.mask_col[rb + q];
      const int32_t off = bwd ? B.off_row[rb + q] : B.off_col[rb + q];
      for (int l = 0; l < nb; ++l)
        if (mask >> l & 1) acc[l] += W[off + __builtin_popcountll(mask & ((1ull << l) - 1))] * t[q];
    }
    for (int l = 0; l < nb; ++l) dst[B.rows[rb + l]] = acc[l];
  }
}

// the substitution blocks, in the kernel's order of operations (levels, tasks, partial sums of the lanes of a task
// added pairwise: neighbours first)
double lane_tree_sum(double *part, int g) {
  for (int off = 1; off < g; off <<= 1)
    for (int p = 0; p + off < g; p += 2 * off) part[p] += part[p + off];
  return part[0];
}
void sub_levels(const int32_t *hdr, int nhdr, const uint16_t *idx, const double *val, std::vector<double> &T) {
  // nhdr headers = barrier levels of kSubWaves wavefronts each (the closing level not counted): every row of a barrier
  // level reads the tile before any of them writes it
  std::vector<std::pair<int, double>> res;
  for (int l0 = 0; l0 + kSubWaves <= nhdr; l0 += kSubWaves) {
    res.clear();
    for (int l = l0; l < l0 + kSubWaves; ++l) {
      const int r0 = hdr[4 * l], g = hdr[4 * l + 1] & 0xff, npl = (hdr[4 * l + 1] >> 8) & 0xf, e0 = hdr[4 * l + 2], i0 = hdr[4 * l + 3],
                r1 = r0 + (hdr[4 * l + 1] >> 12);
      const int nlane = (r1 - r0) * g, istride = npl <= 4 ? 4 : 8;
      for (int r = r0; r < r1; ++r) {
        double part[64];
        for (int p = 0; p < g; ++p) {
          const int lane = (r - r0) * g + p;
          part[p] = 0.0;
          for (int u = 0; u < npl; ++u) part[p] += val[e0 + u * nlane + lane] * T[idx[i0 + lane * istride + u]];
        }
        res.push_back({r, lane_tree_sum(part, g)});
      }
    }
    for (const auto &rv : res) T[rv.first] = rv.second;
  }
}
void apply_sub_forward(const SubBlockOpHost &S, const double *rhs, double *y, double *aux) {
  std::vector<double> T;
  for (size_t b = 0; b < S.nrows.size(); ++b) {
    const int nb = S.nrows[b], rb = S.row_begin[b];
    T.assign(nb, 0.0);
    for (int l = 0; l < nb; ++l) T[l] = rhs[S.rows[rb + l]];
    sub_levels(&S.f_hdr[4 * S.f_lev_begin[b]], S.f_lev_begin[b + 1] - S.f_lev_begin[b] - kSubWaves, S.f_idx.data(),
               &S.f_val[S.f_ent_begin[b]], T);
    for (int l = 0; l < nb; ++l) y[S.rows[rb + l]] = T[l];
    for (int32_t t = S.tgt_begin[b]; t < S.tgt_begin[b + 1]; ++t) {
      double part[16];
      for (int p = 0; p < 16; ++p) {
        part[p] = 0.0;
        for (int32_t k = S.c_ptr[t] + p; k < S.c_ptr[t + 1]; k += 16) part[p] += S.c_val[k] * T[S.c_idx[k]];
      }
      aux[S.tgt_slot[t]] = lane_tree_sum(part, 16);
    }
  }
}
void apply_sub_backward(const SubBlockOpHost &S, const double *y, const double *xlater, double *x) {
  std::vector<double> T;
  for (size_t b = 0; b < S.nrows.size(); ++b) {
    const int nb = S.nrows[b], rb = S.row_begin[b], ntg = S.tgt_begin[b + 1] - S.tgt_begin[b];
    T.assign(nb + ntg, 0.0);
    for (int l = 0; l < nb; ++l) T[l] = y[S.b_rows[rb + l]];
    for (int k = 0; k < ntg; ++k) T[nb + k] = xlater[S.tgt_row[S.tgt_begin[b] + k]];  // the later stage's solution
    sub_levels(&S.b_hdr[4 * S.b_lev_begin[b]], S.b_lev_begin[b + 1] - S.b_lev_begin[b] - kSubWaves, S.b_idx.data(),
               &S.b_val[S.b_ent_begin[b]], T);
    for (int l = 0; l < nb; ++l) x[S.b_rows[rb + l]] = T[l];
  }
}
}  // namespace

void tri_plan_solve_host(const TriPlan &P, int64_t rows, const double *rhs, double *out) {
  const int K = static_cast<int>(P.stages.size());
  if (K == 2 && P.stages[0].sub) {  // factor_solve_sub's sequence
    const SubBlockOpHost &S = P.stages[0].sub_op;
    std::vector<double> work(static_cast<size_t>(P.aux_base) + S.n_aux, 0.0), t2(static_cast<size_t>(rows), 0.0);
    apply_sub_forward(S, rhs, out, work.data() + P.aux_base);
    for (int32_t r : P.top_rows) work[r] = rhs[r];
    if (!P.stages[1].fwd_a.empty()) apply_rowop(P.stages[1].fwd_a, work.data(), work.data(), work.data());  // aux sums, in place
    apply_rowop(P.stages[1].fwd_b, nullptr, work.data(), t2.data());
    apply_rowop(P.stages[1].bwd_b, nullptr, t2.data(), work.data());
    apply_sub_backward(S, out, work.data(), out);
    for (int32_t r : P.top_rows) out[r] = work[r];
    return;
  }
  std::vector<double> t(static_cast<size_t>(rows), 0.0), t2(static_cast<size_t>(rows), 0.0);
  for (int k = 0; k < K; ++k) {
    const TriStage &S = P.stages[k];
    if (S.dense) {
      apply_blocks(S.blocks_op, false, rhs, out);
      continue;
    }
    const double *tk = rhs;
    if (k > 0) {
      apply_rowop(S.fwd_a, rhs, out, t.data());
      tk = t.data();
    }
    apply_rowop(S.fwd_b, nullptr, tk, k == K - 1 ? t2.data() : out);
  }
  for (int k = K - 1; k >= 0; --k) {
    const TriStage &S = P.stages[k];
    if (S.dense) {
      apply_blocks(S.blocks_op, true, out, out);
      continue;
    }
    const double *tk = t2.data();
    if (k + 1 < K) {
      apply_rowop(S.bwd_a, out, out, t.data());
      tk = t.data();
    }
    apply_rowop(S.bwd_b, nullptr, tk, out);
  }
}

}  // namespace cora
