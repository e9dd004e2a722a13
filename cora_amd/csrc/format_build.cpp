// Host-side construction of the device format of the data matrix Q:
//   (1) row partition across ranks (pose-aligned, nnz-balanced),
//   (2) internal row order (rank-major; per rank: rotations | ranges | translations),
//   (3) sliced-ELL storage, one 64-row slice per wavefront, slot-major so every
//       wavefront load is coalesced, plus a separate path for the few very long
//       (landmark) rows.
// Input is the reference's `Problem::data_matrix_` (Eigen row-major CSR,
// src/CORA_problem.cpp:625-712) with the variable layout of
// include/CORA/CORA_problem.h:151-157.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <thread>

#include "cora_internal.h"
#include "parallel.h"

namespace cora {

int g_sigma = kSigma;
int g_pad_even = 0;
int g_long_chunk = kLongChunk;
int g_interleave = 0;  // measured: no gain on MI355X (kept for the lab)
// pose slices in the chain layout (CORA_CHAIN_SLICES=0: the plain layout with every column explicit, measurement switch)
int g_chain_slices = [] { const char *e = std::getenv("CORA_CHAIN_SLICES"); return (e && e[0] == '0') ? 0 : 1; }();

namespace {

struct RowRef {
  int32_t api_row;
  int32_t int_row;
  int32_t len;
};

void emit_slice(HostFormat &F, const std::vector<RowRef> &rows, size_t begin,
                size_t end, int32_t type, int32_t row0, int32_t aux0,
                const int32_t *rowptr, const int32_t *col, const double *val) {
  SliceDesc s{};
  s.row0 = row0;
  s.nrows = static_cast<int32_t>(end - begin);
  s.type = type;
  s.aux0 = aux0;
  int width = 0;
  for (size_t i = begin; i < end; ++i) width = std::max(width, rows[i].len);
  s.width = width;
  s.off = static_cast<int64_t>(F.sval.size());
  s.coff = static_cast<int32_t>(F.scol.size());
  F.max_width = std::max(F.max_width, width);
  const size_t base = F.sval.size(), cbase = F.scol.size();
  F.sval.resize(base + static_cast<size_t>(width) * kWave, 0.0);
  F.scol.resize(cbase + static_cast<size_t>(width) * kWave, 0);
  for (int lane = 0; lane < kWave; ++lane) {
    // padding lanes replicate the last active row's columns with zero values
    const size_t src = begin + std::min<size_t>(lane, end - begin - 1);
    const RowRef &rr = rows[src];
    const bool active = lane < s.nrows;
    const int32_t p0 = rowptr[rr.api_row];
    int32_t fill = rr.len > 0 ? F.api2int[col[p0]] : rr.int_row;
    for (int k = 0; k < width; ++k) {
      const size_t dst = base + static_cast<size_t>(k) * kWave + lane;
      const size_t cdst = cbase + static_cast<size_t>(k) * kWave + lane;
      if (k < rr.len) {
        F.scol[cdst] = F.api2int[col[p0 + k]];
        F.sval[dst] = active ? val[p0 + k] : 0.0;
        fill = F.scol[cdst];
      } else {
        F.scol[cdst] = fill;  // padded slot: re-reads a row already in cache
        F.sval[dst] = 0.0;
      }
    }
  }
  F.padded_nnz += static_cast<int64_t>(width) * kWave;
  F.slices.push_back(s);
}

}  // namespace

void build_format(int d, int n, int r, int nt, const int32_t *rowptr,
                  const int32_t *col, const double *val, int rank, int world,
                  HostFormat &F, bool distribute_long_rows) {
  const bool dist_long = distribute_long_rows && world > 1;
  const bool timing = std::getenv("CORA_FORMAT_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "  [format] %-28s %.4f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  if (d != 2 && d != 3) throw std::runtime_error("cora: dimension d must be 2 or 3");
  if (n < 0 || r < 0 || nt < n) throw std::runtime_error("cora: invalid problem sizes");
  if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("cora: invalid rank/world");
  const int64_t dn = static_cast<int64_t>(d) * n;
  const int64_t N = dn + r + nt;
  if (N <= 0) throw std::runtime_error("cora: empty problem");
  if (N > 2000000000LL) throw std::runtime_error("cora: problem too large for int32 rows");
  if (rowptr[0] != 0) throw std::runtime_error("cora: rowptr[0] must be 0 (call makeCompressed())");
  const int l = nt - n;
  Layout &L = F.L;
  L.d = d; L.n = n; L.r = r; L.nt = nt; L.N = N; L.rank = rank; L.world = world;
  F.nnz_global = rowptr[N];
  for (int64_t i = 0; i < N; ++i) {
    if (rowptr[i + 1] < rowptr[i]) throw std::runtime_error("cora: rowptr not monotone");
    for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q)
      if (col[q] < 0 || col[q] >= N) throw std::runtime_error("cora: column index out of range");
  }
  auto rowlen = [&](int64_t i) { return rowptr[i + 1] - rowptr[i]; };

  tick("checks");
  // ---- 1. owner of every pose / range row / landmark -----------------------
  std::vector<int> pose_owner(n, 0), range_pose(r, -1), range_lm(r, -1), range_owner(r, 0), lm_owner(l, 0);
  const int64_t tb = dn + r;
  // a range row belongs with the first pose translation it touches (Q23 holds
  // its two endpoints, src/CORA_problem.cpp:660-663)
  for (int k = 0; k < r; ++k) {
    const int64_t row = dn + k;
    for (int32_t q = rowptr[row]; q < rowptr[row + 1]; ++q) {
      const int64_t c = col[q];
      if (c >= tb) {
        const int64_t t = c - tb;
        if (t < n) { if (range_pose[k] < 0) range_pose[k] = static_cast<int>(t); }
        else if (range_lm[k] < 0) range_lm[k] = static_cast<int>(t - n);
      }
    }
  }
  if (world > 1) {
    for (int j = 0; j < l; ++j) lm_owner[j] = j % world;
    for (int k = 0; k < r; ++k)
      if (range_pose[k] < 0) range_owner[k] = range_lm[k] >= 0 ? lm_owner[range_lm[k]] : 0;
    std::vector<int64_t> w(n, 0), lw(world, 0);
    for (int i = 0; i < n; ++i) {
      for (int a = 0; a < d; ++a) w[i] += rowlen(static_cast<int64_t>(i) * d + a);
      w[i] += rowlen(tb + i);
    }
    for (int k = 0; k < r; ++k) {
      if (range_pose[k] >= 0) w[range_pose[k]] += rowlen(dn + k);
      else lw[range_owner[k]] += rowlen(dn + k);
    }
    for (int j = 0; j < l; ++j) {
      const int64_t len = rowlen(tb + n + j);
      if (len > kLongRow && distribute_long_rows) {  // a distributed long row (below): every rank works on the columns it owns
        for (int gg = 0; gg < world; ++gg) lw[gg] += len / world;
      } else {
        lw[lm_owner[j]] += len;
      }
    }
    const int64_t total = std::accumulate(w.begin(), w.end(), int64_t{0}) +
                          std::accumulate(lw.begin(), lw.end(), int64_t{0});
    int g = 0;
    int64_t acc = lw[0];
    const double target = static_cast<double>(total) / world;
    for (int i = 0; i < n; ++i) {
      // move on when this rank is full, keeping enough poses for the ranks left
      if (g < world - 1 && acc + w[i] / 2 > target) {
        ++g;
        acc = lw[g];
      }
      pose_owner[i] = g;
      acc += w[i];
    }
    for (int k = 0; k < r; ++k)
      if (range_pose[k] >= 0) range_owner[k] = pose_owner[range_pose[k]];
  }

  // ---- 2. internal numbering ------------------------------------------------
  std::vector<int64_t> np(world, 0), nr(world, 0), nlm(world, 0);
  for (int i = 0; i < n; ++i) np[pose_owner[i]]++;
  for (int k = 0; k < r; ++k) nr[range_owner[k]]++;
  for (int j = 0; j < l; ++j) nlm[lm_owner[j]]++;
  int64_t shard = 0;
  for (int g = 0; g < world; ++g)
    shard = std::max(shard, d * np[g] + nr[g] + np[g] + nlm[g]);
  if (world > 1) shard = (shard + 7) & ~int64_t{7};
  L.shard_rows = shard;
  L.rows = shard * world;
  if (L.rows > 2000000000LL) throw std::runtime_error("cora: too many internal rows");
  L.base = shard * rank;
  L.nl_poses = static_cast<int>(np[rank]);
  L.nl_ranges = static_cast<int>(nr[rank]);
  L.nl_trans = static_cast<int>(np[rank] + nlm[rank]);
  L.rot_base = L.base;
  L.rng_base = L.rot_base + static_cast<int64_t>(d) * L.nl_poses;
  L.trn_base = L.rng_base + L.nl_ranges;
  L.local_rows = static_cast<int64_t>(d) * L.nl_poses + L.nl_ranges + L.nl_trans;

  F.api2int.assign(N, -1);
  F.int2api.assign(L.rows, -1);
  {
    std::vector<int64_t> cp(world, 0), cr(world, 0), ct(world, 0);
    for (int i = 0; i < n; ++i) {
      const int g = pose_owner[i];
      const int64_t b = shard * g;
      for (int a = 0; a < d; ++a)
        F.api2int[static_cast<int64_t>(i) * d + a] = static_cast<int32_t>(b + d * cp[g] + a);
      // pose translation
      F.api2int[dn + r + i] = static_cast<int32_t>(b + d * np[g] + nr[g] + cp[g]);
      cp[g]++;
    }
    // range rows follow the order of the pose they hang off, so that the Q23 /
    // Q32 blocks are banded in the internal order whatever the measurement order
    std::vector<int32_t> rorder(r);
    std::iota(rorder.begin(), rorder.end(), 0);
    std::stable_sort(rorder.begin(), rorder.end(), [&](int32_t a, int32_t b) {
      const int pa = range_pose[a] < 0 ? n : range_pose[a], pb = range_pose[b] < 0 ? n : range_pose[b];
      return pa < pb;
    });
    for (int32_t k : rorder) {
      const int g = range_owner[k];
      F.api2int[dn + k] = static_cast<int32_t>(shard * g + d * np[g] + cr[g]++);
    }
    for (int j = 0; j < l; ++j) {
      const int g = lm_owner[j];
      F.api2int[dn + r + n + j] =
          static_cast<int32_t>(shard * g + d * np[g] + nr[g] + np[g] + ct[g]++);
    }
    for (int64_t i = 0; i < N; ++i) F.int2api[F.api2int[i]] = static_cast<int32_t>(i);
  }

  // local pose index each local range row hangs off (work ordering key)
  std::vector<double> local_range_pose(static_cast<size_t>(std::max(L.nl_ranges, 1)), 0.0);
  for (int k = 0; k < r; ++k) {
    if (range_owner[k] != rank) continue;
    const int64_t li = F.api2int[dn + k] - L.rng_base;
    double key = L.nl_poses;  // ranges between landmarks go last
    if (range_pose[k] >= 0 && pose_owner[range_pose[k]] == rank)
      key = static_cast<double>((F.api2int[static_cast<int64_t>(range_pose[k]) * d] - L.rot_base) / d);
    local_range_pose[li] = key;
  }

  tick("owners + numbering");
  // ---- 3. local rows -> slices ---------------------------------------------
  std::vector<double> slice_key;  // position of each slice along the pose chain (work ordering)
  F.slices.clear(); F.sval.clear(); F.scol.clear(); F.perm.clear(); F.head_val.clear();
  F.chunks.clear(); F.lval.clear(); F.lcol.clear();
  F.padded_nnz = F.long_nnz = F.nnz_local = 0; F.max_width = 0; F.n_long_rows = 0;
  F.diag.assign(static_cast<size_t>(std::max<int64_t>(L.local_rows, 1)), 0.0);

  auto local_row = [&](int64_t int_row, int64_t *nnz_acc = nullptr) {  // (nnz_acc: a thread's own count, added up later)
    RowRef rr;
    rr.int_row = static_cast<int32_t>(int_row);
    rr.api_row = F.int2api[int_row];
    rr.len = rowlen(rr.api_row);
    *(nnz_acc ? nnz_acc : &F.nnz_local) += rr.len;
    for (int32_t q = rowptr[rr.api_row]; q < rowptr[rr.api_row + 1]; ++q)
      if (col[q] == rr.api_row) F.diag[int_row - L.base] += val[q];
    return rr;
  };

  // Pose slices: lane = pose.  The d rotation rows of a pose share (almost)
  // the same column pattern -- Q11 is made of dense d x d blocks and Q13 of
  // d x 1 columns (src/CORA_problem.cpp:297-377, 639-652) -- so the union
  // pattern is stored once with d values per column: one 4-byte index and one
  // X-row gather serve d nonzeros.
  //
  // Chain layout (kSliceChainFlag, cora_internal.h).  Along a pose chain every pose has the same columns -- its own
  // rotation block, the next and the previous pose's, the translations t_P and t_{P+1} -- so they carry no index; Q is
  // symmetric, so the previous pose's block, the rotation part of the pose's translation row (Q31 = Q13^T) and the
  // sub-diagonal of Q33 are what a neighbouring slot already holds; and the lane takes the pose's translation row with
  // it: what is left of that row (its range measurements) is a short compact tail.  26 values + a 4-byte tail
  // descriptor per pose at d = 3 instead of 33 values + 11 indices in the pose slice and ~11 values + 11 gathered rows
  // of X in a translation-row slice of its own.  Every identity the layout relies on is checked bit for bit here; a
  // slice that fails one keeps the plain layout (all columns explicit) and its translation rows go to the row slices.
  std::vector<char> trn_owned(static_cast<size_t>(std::max(L.nl_trans, 1)), 0);
  {
    struct PoseCols { std::vector<int32_t> c; std::vector<double> v; };  // v[k*d + a]
    struct RowEnt { std::vector<int32_t> c; std::vector<double> v; };
    struct Ent { int32_t col, a, seq; double v; };  // one nonzero: internal column, row of the pose, position in the CSR
    struct ChainLane {
      double s0[4], s1[4], nxt[9], own[9], hq[3], ht, prev[9];
      int nlocal = 0;  // pairs of the tail whose columns are rows of this shard (they come first)
      std::vector<int32_t> gc, tc;
      std::vector<double> gv, tv;
    };
    // a thread's scratch: kept from slice to slice so that the loop allocates nothing once it is warm
    struct Scratch {
      std::vector<PoseCols> pc;
      std::vector<RowEnt> tr;
      std::vector<ChainLane> cl;
      std::vector<Ent> ent;
      std::vector<int32_t> lc, rc;
      std::vector<double> lv, rv;
    };
    const int lanes = std::min(kWave, std::max(L.nl_poses, 1));
    const int FV = kChainFixed(d), HV = kChainHead(d);
    // The slices are independent of each other: a few threads build them, each slice into buffers of its own, and
    // they are put together in order afterwards (the same format whatever the thread count).
    struct SliceOut {
      SliceDesc sd{};
      std::vector<double> v;
      std::vector<int32_t> c;
      int64_t padded = 0, nnz = 0;
      int maxw = 0;
    };
    const int n_pose_slices = (L.nl_poses + kWave - 1) / kWave;
    std::vector<SliceOut> outs(static_cast<size_t>(n_pose_slices));
    F.head_val.assign(static_cast<size_t>(n_pose_slices) * HV, 0.0);
    auto build_slice = [&](int p0, SliceOut &O, Scratch &W) {
      std::vector<PoseCols> &pc = W.pc;
      std::vector<RowEnt> &tr = W.tr;
      std::vector<Ent> &ent = W.ent;
      const int cnt = std::min(kWave, L.nl_poses - p0);
      int width = 0;
      for (int q = 0; q < cnt; ++q) {
        PoseCols &P = pc[q];
        P.c.clear(); P.v.clear();
        ent.clear();
        for (int a = 0; a < d; ++a) {
          const RowRef rr = local_row(L.rot_base + static_cast<int64_t>(p0 + q) * d + a, &O.nnz);
          for (int32_t t = rowptr[rr.api_row]; t < rowptr[rr.api_row + 1]; ++t)
            ent.push_back({F.api2int[col[t]], a, static_cast<int32_t>(ent.size()), val[t]});
        }
        // by column, equal columns in the order the CSR holds them (they are summed in that order)
        if (!g_interleave) {
          std::sort(ent.begin(), ent.end(),
                    [](const Ent &x, const Ent &y) { return x.col != y.col ? x.col < y.col : x.seq < y.seq; });
        } else {
          // columns visited ordered by (col mod d, col / d): neighbouring poses then read one X row in
          // consecutive slots, so it is still in L1 when re-referenced
          std::sort(ent.begin(), ent.end(), [d](const Ent &x, const Ent &y) {
            const int xa = x.col % d, ya = y.col % d;
            return xa != ya ? xa < ya : x.col != y.col ? x.col < y.col : x.seq < y.seq;
          });
        }
        for (const Ent &e : ent) {
          if (P.c.empty() || P.c.back() != e.col) {
            P.c.push_back(e.col);
            P.v.resize(P.v.size() + d, 0.0);
          }
          P.v[(P.c.size() - 1) * d + e.a] += e.v;
        }
        width = std::max(width, static_cast<int>(P.c.size()));
      }
      // ---- chain layout: gather what every lane stores and check what it does not store ----
      bool chain = g_chain_slices != 0;
      std::vector<ChainLane> &cl = W.cl;
      if (chain && cl.size() < static_cast<size_t>(cnt)) cl.resize(static_cast<size_t>(cnt));
      if (chain) {
        for (int q = 0; q < cnt; ++q) {  // the pose's translation row, internal columns in increasing order
          RowEnt &T = tr[q];
          T.c.clear(); T.v.clear();
          const int32_t api = F.int2api[L.trn_base + p0 + q];
          if (rowptr[api + 1] - rowptr[api] > kLongRow) chain = false;  // a long row stays on the chunked path
          ent.clear();
          for (int32_t t = rowptr[api]; t < rowptr[api + 1]; ++t)
            ent.push_back({F.api2int[col[t]], 0, static_cast<int32_t>(ent.size()), val[t]});
          std::sort(ent.begin(), ent.end(),
                    [](const Ent &x, const Ent &y) { return x.col != y.col ? x.col < y.col : x.seq < y.seq; });
          for (const Ent &e : ent) {
            if (!T.c.empty() && T.c.back() == e.col) T.v.back() += e.v;
            else { T.c.push_back(e.col); T.v.push_back(e.v); }
          }
        }
        for (int q = 0; q < cnt; ++q) {
          ChainLane &C = cl[q];
          std::memset(C.s0, 0, sizeof C.s0); std::memset(C.s1, 0, sizeof C.s1);
          std::memset(C.nxt, 0, sizeof C.nxt); std::memset(C.own, 0, sizeof C.own);
          std::memset(C.hq, 0, sizeof C.hq); std::memset(C.prev, 0, sizeof C.prev);
          C.ht = 0.0;
          C.nlocal = 0;
          C.gc.clear(); C.gv.clear(); C.tc.clear(); C.tv.clear();
          const int P = p0 + q;
          const int64_t me = L.rot_base + static_cast<int64_t>(P) * d, tme = L.trn_base + P;
          const bool has_next = P + 1 < L.nl_poses, has_prev = P > 0;
          const PoseCols &R = pc[q];
          for (size_t k = 0; k < R.c.size(); ++k) {
            const int64_t c = R.c[k];
            const double *v = &R.v[k * d];
            if (c == tme) { for (int a = 0; a < d; ++a) C.s0[a] = v[a]; }
            else if (has_next && c == tme + 1) { for (int a = 0; a < d; ++a) C.s1[a] = v[a]; }
            else if (has_next && c >= me + d && c < me + 2 * d) { for (int a = 0; a < d; ++a) C.nxt[(c - me - d) * d + a] = v[a]; }
            else if (c >= me && c < me + d) { for (int a = 0; a < d; ++a) C.own[(c - me) * d + a] = v[a]; }
            else if (has_prev && c >= me - d && c < me) { for (int a = 0; a < d; ++a) C.prev[(c - me + d) * d + a] = v[a]; }
            else { C.gc.push_back(static_cast<int32_t>(c)); for (int a = 0; a < d; ++a) C.gv.push_back(v[a]); }
          }
          const RowEnt &T = tr[q];
          double trot[3] = {0.0, 0.0, 0.0};  // Q(t_P, rot(P)_c)
          for (size_t k = 0; k < T.c.size(); ++k) {
            const int64_t c = T.c[k];
            const double v = T.v[k];
            if (c == tme) C.s0[d] = v;
            else if (has_next && c == tme + 1) C.s1[d] = v;
            else if (c >= me && c < me + d) trot[c - me] = v;
            else if (has_prev && c >= me - d && c < me) C.hq[c - me + d] = v;
            else if (has_prev && c == tme - 1) C.ht = v;
            else { C.tc.push_back(static_cast<int32_t>(c)); C.tv.push_back(v); }
          }
          for (int c = 0; c < d; ++c)
            if (trot[c] != C.s0[c]) chain = false;  // Q31 = Q13^T on the pose's own block
        }
        for (int q = 1; q < cnt && chain; ++q) {  // what lane q takes from lane q - 1
          const ChainLane &C = cl[q], &B = cl[q - 1];
          for (int a = 0; a < d; ++a)
            for (int c = 0; c < d; ++c)
              if (C.prev[c * d + a] != B.nxt[a * d + c]) chain = false;  // Q(rot(P)_a, rot(P-1)_c) = Q(rot(P-1)_c, rot(P)_a)
          for (int c = 0; c < d; ++c)
            if (C.hq[c] != B.s1[c]) chain = false;                        // Q(t_P, rot(P-1)_c) = Q(rot(P-1)_c, t_P)
          if (C.ht != B.s1[d]) chain = false;                             // Q(t_P, t_{P-1}) = Q(t_{P-1}, t_P)
        }
        size_t T = 0;  // the tail is stored as PAIRS of entries (cora_internal.h): an odd count is padded with a zero
        for (int q = 0; q < cnt && chain; ++q) {
          ChainLane &C = cl[q];
          // columns of this shard first, rows of other ranks after them (each group padded to whole pairs): a
          // partitioned handle can run the local part before the exchange of the operand has landed
          std::vector<int32_t> &lc = W.lc, &rc = W.rc;
          std::vector<double> &lv = W.lv, &rv = W.rv;
          lc.clear(); rc.clear(); lv.clear(); rv.clear();
          for (size_t k = 0; k < C.tc.size(); ++k) {
            const bool local = C.tc[k] >= L.base && C.tc[k] < L.base + L.shard_rows;
            (local ? lc : rc).push_back(C.tc[k]);
            (local ? lv : rv).push_back(C.tv[k]);
          }
          if (lc.size() & 1) { lc.push_back(lc.back()); lv.push_back(0.0); }
          if (rc.size() & 1) { rc.push_back(rc.back()); rv.push_back(0.0); }
          C.nlocal = static_cast<int>(lc.size() / 2);
          C.tc.assign(lc.begin(), lc.end()); C.tc.insert(C.tc.end(), rc.begin(), rc.end());
          C.tv.assign(lv.begin(), lv.end()); C.tv.insert(C.tv.end(), rv.begin(), rv.end());
          if (C.tc.size() / 2 > static_cast<size_t>(kSliceTailMaxMask)) chain = false;
          T += C.tc.size() / 2;
        }
        if (T > 0xffffu) chain = false;
      }
      SliceDesc sd{};
      sd.row0 = static_cast<int32_t>(L.rot_base + static_cast<int64_t>(p0) * d);
      sd.nrows = cnt;
      sd.off = 0;  // (placed when the slices are put together, below)
      sd.coff = 0;
      sd.aux0 = p0;
      if (chain) {
        int gw = 0;
        size_t T = 0, mc = 0;
        for (int q = 0; q < cnt; ++q) {
          gw = std::max(gw, static_cast<int>(cl[q].gc.size()));
          mc = std::max(mc, cl[q].tc.size() / 2);
          T += cl[q].tc.size() / 2;
        }
        double *hv = &F.head_val[static_cast<size_t>(p0 / kWave) * HV];
        for (int a = 0; a < d; ++a)
          for (int c = 0; c < d; ++c) hv[a * d + c] = cl[0].prev[c * d + a];
        for (int c = 0; c < d; ++c) hv[d * d + c] = cl[0].hq[c];
        hv[d * d + d] = cl[0].ht;
        sd.width = gw;
        sd.type = kSliceStiefel | kSliceChainFlag | static_cast<int32_t>(mc << kSliceTailMaxShift) |
                  static_cast<int32_t>(static_cast<uint32_t>(T) << kSliceTailShift);
        O.maxw = gw + 2 + 2 * d;
        O.v.assign((static_cast<size_t>(FV) + static_cast<size_t>(gw) * d) * kWave + 2 * T, 0.0);
        O.c.assign((1 + static_cast<size_t>(gw)) * kWave + 2 * T, 0);
        double *tv = O.v.data() + (static_cast<size_t>(FV) + static_cast<size_t>(gw) * d) * kWave;
        int32_t *tc = O.c.data() + (1 + static_cast<size_t>(gw)) * kWave;
        size_t te = 0;
        for (int lane = 0; lane < kWave; ++lane) {
          const bool active = lane < cnt;
          const ChainLane &C = cl[std::min(lane, cnt - 1)];
          auto put = [&](int slot, double v) { O.v[static_cast<size_t>(slot) * kWave + lane] = active ? v : 0.0; };
          for (int a = 0; a <= d; ++a) { put(a, C.s0[a]); put(d + 1 + a, C.s1[a]); }
          for (int k = 0; k < d * d; ++k) { put(2 * (d + 1) + k, C.nxt[k]); put(2 * (d + 1) + d * d + k, C.own[k]); }
          // general slots; padded slots repeat a column of the lane (or the lane's own first row) with zero values
          int32_t fill = static_cast<int32_t>(L.rot_base + static_cast<int64_t>(std::min(p0 + lane, L.nl_poses - 1)) * d);
          for (int k = 0; k < gw; ++k) {
            const bool have = k < static_cast<int>(C.gc.size());
            if (have) fill = C.gc[k];
            O.c[(1 + static_cast<size_t>(k)) * kWave + lane] = fill;
            for (int a = 0; a < d; ++a) put(FV + k * d + a, have ? C.gv[static_cast<size_t>(k) * d + a] : 0.0);
          }
          uint32_t info = static_cast<uint32_t>(te);
          if (active) {
            info |= static_cast<uint32_t>(C.tc.size() / 2) << 16 | static_cast<uint32_t>(C.nlocal) << 24;
            for (size_t k = 0; k + 1 < C.tc.size(); k += 2) {
              tc[2 * te] = C.tc[k]; tv[2 * te] = C.tv[k];
              tc[2 * te + 1] = C.tc[k + 1]; tv[2 * te + 1] = C.tv[k + 1];
              ++te;
            }
            trn_owned[static_cast<size_t>(p0 + lane)] = 1;
          }
          O.c[lane] = static_cast<int32_t>(info);
        }
        O.padded += (static_cast<int64_t>(FV) + static_cast<int64_t>(gw) * d) * kWave + 2 * static_cast<int64_t>(T);
      } else {
        sd.width = width;
        sd.type = kSliceStiefel;
        O.maxw = width;
        O.v.assign(static_cast<size_t>(width) * d * kWave, 0.0);
        O.c.assign(static_cast<size_t>(width) * kWave, 0);
        for (int lane = 0; lane < kWave; ++lane) {
          const PoseCols &P = pc[std::min(lane, cnt - 1)];
          const bool active = lane < cnt;
          int32_t fill = P.c.empty() ? sd.row0 : P.c[0];
          for (int k = 0; k < width; ++k) {
            const bool have = k < static_cast<int>(P.c.size());
            if (have) fill = P.c[k];
            O.c[static_cast<size_t>(k) * kWave + lane] = fill;
            for (int a = 0; a < d; ++a)
              O.v[(static_cast<size_t>(k) * d + a) * kWave + lane] =
                  (have && active) ? P.v[static_cast<size_t>(k) * d + a] : 0.0;
          }
        }
        O.padded += static_cast<int64_t>(width) * d * kWave;
      }
      O.sd = sd;
    };
    {
      unsigned nth = n_pose_slices < 64 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
      if (const char *e = std::getenv("CORA_FORMAT_THREADS")) nth = static_cast<unsigned>(std::max(1, std::atoi(e)));
      parallel_parts(nth, [&](unsigned t) {
        Scratch W;
        W.pc.resize(static_cast<size_t>(lanes));
        W.tr.resize(static_cast<size_t>(lanes));
        const int s_begin = static_cast<int>(static_cast<int64_t>(n_pose_slices) * t / nth);
        const int s_end = static_cast<int>(static_cast<int64_t>(n_pose_slices) * (t + 1) / nth);
        for (int sl = s_begin; sl < s_end; ++sl) build_slice(sl * kWave, outs[static_cast<size_t>(sl)], W);
      });
    }
    tick("pose slices: built");
    size_t nv = F.sval.size(), nc = F.scol.size();
    for (const SliceOut &O : outs) { nv += O.v.size(); nc += O.c.size(); }
    F.sval.reserve(nv);
    F.scol.reserve(nc);
    for (SliceOut &O : outs) {
      O.sd.off = static_cast<int64_t>(F.sval.size());
      O.sd.coff = static_cast<int32_t>(F.scol.size());
      F.sval.insert(F.sval.end(), O.v.begin(), O.v.end());
      F.scol.insert(F.scol.end(), O.c.begin(), O.c.end());
      F.padded_nnz += O.padded;
      F.nnz_local += O.nnz;
      F.max_width = std::max(F.max_width, O.maxw);
      slice_key.push_back(O.sd.aux0);
      F.slices.push_back(O.sd);
      std::vector<double>().swap(O.v);
      std::vector<int32_t>().swap(O.c);
    }
  }
  tick("pose slices");
  // Oblique slices
  {
    std::vector<RowRef> rows;
    rows.reserve(L.nl_ranges);
    for (int64_t i = 0; i < L.nl_ranges; ++i) rows.push_back(local_row(L.rng_base + i));
    // LAB BUILDS ONLY (-DCORA_LAB_BUILD): a launch without the range slices -- WRONG range rows -- measured what folding them
    // into the pose slices could gain at best (round 5: nothing).  Not in the product library: a user who set the variable got
    // silent garbage.
#ifdef CORA_LAB_BUILD
    const bool lab_skip = std::getenv("CORA_LAB_SKIP_RANGE_SLICES") != nullptr;
#else
    constexpr bool lab_skip = false;
#endif
    for (int64_t k0 = 0; k0 < L.nl_ranges && !lab_skip; k0 += kWave) {
      const int64_t cnt = std::min<int64_t>(kWave, L.nl_ranges - k0);
      emit_slice(F, rows, k0, k0 + cnt, kSliceOblique,
                 static_cast<int32_t>(L.rng_base + k0), static_cast<int32_t>(k0),
                 rowptr, col, val);
      slice_key.push_back(local_range_pose[k0] + 0.25);
    }
  }
  tick("range slices");
  // Translation rows: long rows -> chunked path; the rest sorted by length
  // inside windows of kSigma rows (keeps column locality) and sliced.
  {
    std::vector<RowRef> rows;
    rows.reserve(L.nl_trans);
    // Partitioned handles: DISTRIBUTED long rows.  The columns of a landmark row span every rank (the translation
    // and range rows of all the poses that saw the landmark), so its owner used to ask for ~10^4 remote rows of X per
    // landmark (18 % of the vector at 8 ranks on the 10^5-pose graph).  Instead every rank multiplies the part of every
    // long row that falls on the columns it owns -- the same list of long rows, in API order, on every rank -- into
    // slot j of a small buffer (SpmmArgs::long_out), the buffers are summed over the ranks after the product and the
    // owner copies its rows out (capi.hip, finish_long_rows): the exchange shrinks to the chain halo plus the
    // landmarks' own rows of X.
    F.long_rows.clear();
    F.long_owner.clear();
    if (dist_long) {
      for (int64_t api = tb; api < N; ++api) {
        const int len = rowlen(api);
        if (len <= kLongRow) continue;
        const int32_t int_row = F.api2int[api];
        const int owner = static_cast<int>(int_row / shard);
        const int32_t p0 = rowptr[api];
        const int32_t k_begin = static_cast<int32_t>(F.lval.size());
        for (int k = 0; k < len; ++k) {
          const int32_t ic = F.api2int[col[p0 + k]];
          if (ic / shard != rank) continue;
          F.lval.push_back(val[p0 + k]);
          F.lcol.push_back(ic);
        }
        const int mylen = static_cast<int>(F.lval.size()) - k_begin;
        const int nch = (mylen + g_long_chunk - 1) / g_long_chunk;  // 0: nothing of this row on this rank
        const int32_t first = static_cast<int32_t>(F.chunks.size());
        for (int c = 0; c < nch; ++c) {
          LongChunk ch{};
          ch.row = int_row;
          ch.k0 = k_begin + c * g_long_chunk;
          ch.k1 = k_begin + std::min(mylen, (c + 1) * g_long_chunk);
          ch.nchunks = nch;
          ch.first = first;
          ch.slot = F.n_long_rows;
          F.chunks.push_back(ch);
        }
        F.long_rows.push_back(int_row);
        F.long_owner.push_back(owner);
        F.n_long_rows++;
        F.long_nnz += mylen;
        F.nnz_local += mylen;  // (the owner's local_row() below counts the whole row: taken back there)
      }
    }
    for (int64_t i = 0; i < L.nl_trans; ++i) {
      RowRef rr = local_row(L.trn_base + i);
      if (rr.len > kLongRow && dist_long) {  // distributed above (local_row() has taken its diagonal)
        F.nnz_local -= rr.len;
        continue;
      }
      if (i < L.nl_poses && trn_owned[static_cast<size_t>(i)]) continue;  // the pose's chain slice has it
      if (rr.len > kLongRow) {
        const int32_t p0 = rowptr[rr.api_row];
        const int32_t k_begin = static_cast<int32_t>(F.lval.size());
        for (int k = 0; k < rr.len; ++k) {
          F.lval.push_back(val[p0 + k]);
          F.lcol.push_back(F.api2int[col[p0 + k]]);
        }
        const int nch = (rr.len + g_long_chunk - 1) / g_long_chunk;
        const int32_t first = static_cast<int32_t>(F.chunks.size());
        for (int c = 0; c < nch; ++c) {
          LongChunk ch{};
          ch.row = rr.int_row;
          ch.k0 = k_begin + c * g_long_chunk;
          ch.k1 = k_begin + std::min(rr.len, (c + 1) * g_long_chunk);
          ch.nchunks = nch;
          ch.first = first;
          ch.slot = F.n_long_rows;
          F.chunks.push_back(ch);
        }
        F.n_long_rows++;
        F.long_nnz += rr.len;
      } else {
        rows.push_back(rr);
      }
    }
    const size_t sigma = static_cast<size_t>(std::max(g_sigma, kWave) / kWave) * kWave;
    std::vector<double> window_key;
    for (size_t w0 = 0; w0 < rows.size(); w0 += sigma) {
      const size_t w1 = std::min(rows.size(), w0 + sigma);
      window_key.push_back(static_cast<double>(rows[w0].int_row - L.trn_base));
      std::stable_sort(rows.begin() + w0, rows.begin() + w1,
                       [](const RowRef &a, const RowRef &b) { return a.len > b.len; });
    }
    for (size_t k0 = 0; k0 < rows.size(); k0 += kWave) {
      const size_t cnt = std::min<size_t>(kWave, rows.size() - k0);
      slice_key.push_back(window_key[k0 / sigma] + 0.5 + 1e-3 * static_cast<double>((k0 % sigma) / kWave));
      const int32_t poff = static_cast<int32_t>(F.perm.size());
      for (int lane = 0; lane < kWave; ++lane)
        F.perm.push_back(rows[k0 + std::min<size_t>(lane, cnt - 1)].int_row);
      emit_slice(F, rows, k0, k0 + cnt, kSliceEuclidPerm, poff, 0, rowptr, col, val);
    }
  }

  tick("translation + long rows");
  // long-row chunks: launch order by the region of X they read
  F.chunk_order.resize(F.chunks.size());
  std::iota(F.chunk_order.begin(), F.chunk_order.end(), 0);
  std::stable_sort(F.chunk_order.begin(), F.chunk_order.end(), [&](int32_t a, int32_t b) {
    const int32_t ca = F.chunks[a].k1 > F.chunks[a].k0 ? F.lcol[F.chunks[a].k0] : 0;
    const int32_t cb = F.chunks[b].k1 > F.chunks[b].k0 ? F.lcol[F.chunks[b].k0] : 0;
    return ca < cb;
  });

  // ---- 4. work order: walk the pose chain, so that the pose / range /
  // translation slices that gather the same rows of X run close together in
  // time (and, with the kernel's per-XCD block chunking, on the same L2).
  {
    std::vector<size_t> order(F.slices.size());
    std::iota(order.begin(), order.end(), size_t{0});
    std::stable_sort(order.begin(), order.end(),
                     [&](size_t a, size_t b) { return slice_key[a] < slice_key[b]; });
    std::vector<SliceDesc> sorted;
    sorted.reserve(F.slices.size());
    for (size_t i : order) sorted.push_back(F.slices[i]);
    F.slices.swap(sorted);
    // Second order of the same list for small row strides: inside each XCD's contiguous eighth (k_spmm: per_xcd =
    // ceil(n_slices / 8)) the pose slices go first -- their wavefronts live longest (X window + d x LD accumulators
    // + the Hvp epilogue), and launched last they were the tail of the kernel.  The eighth still covers the same
    // poses, so its rows of X stay in that XCD's L2.  Measured at 10^5 poses (profiles/r02_rank_sweep.md): Hvp
    // 2-4 % faster up to a row stride of 6, 3-9 % SLOWER from 10 on, hence kPoseFirstMaxLD.  CORA_SLICE_LJF=0
    // switches it off (measurement switch).
    F.slices_pose_first.clear();
    const char *ljf = std::getenv("CORA_SLICE_LJF");
    if (!(ljf && ljf[0] == '0')) {
      F.slices_pose_first = F.slices;
      const size_t per = (F.slices.size() + 7) / 8;
      for (size_t x = 0; x < 8; ++x) {
        const size_t b = std::min(x * per, F.slices.size()), e = std::min(b + per, F.slices.size());
        std::stable_partition(F.slices_pose_first.begin() + b, F.slices_pose_first.begin() + e,
                              [](const SliceDesc &sd) { return (sd.type & kSliceTypeMask) == kSliceStiefel; });
      }
    }
  }
  tick("work order");
}

void slice_columns(const HostFormat &F, const SliceDesc &sd, std::vector<int32_t> &out) {
  const int32_t *cb = F.scol.data() + sd.coff;
  if (sd.type & kSliceChainFlag) {
    const size_t T = static_cast<uint32_t>(sd.type) >> kSliceTailShift;
    out.insert(out.end(), cb + kWave, cb + (1 + static_cast<size_t>(sd.width)) * kWave + 2 * T);
  } else {
    out.insert(out.end(), cb, cb + static_cast<size_t>(sd.width) * kWave);
  }
}

void format_spmm_host(const HostFormat &F, const double *X, int ld, double *out) {
  const int d = F.L.d;
  const Layout &L = F.L;
  std::vector<double> acc(static_cast<size_t>(ld) * 4);
  auto axpy = [&](double *y, double v, int64_t row) {
    const double *xr = X + static_cast<size_t>(row) * ld;
    for (int c = 0; c < ld; ++c) y[c] += v * xr[c];
  };
  for (const SliceDesc &s : F.slices) {
    for (int lane = 0; lane < s.nrows; ++lane) {
      std::fill(acc.begin(), acc.end(), 0.0);
      if ((s.type & kSliceTypeMask) == kSliceStiefel && (s.type & kSliceChainFlag)) {
        // the chain layout exactly as the kernel reads it (cora_internal.h): implied columns, the lane before, the tail
        const int FV = kChainFixed(d), HV = kChainHead(d);
        const double *vb = F.sval.data() + s.off;
        const int32_t *cb = F.scol.data() + s.coff;
        const double *tv = vb + (static_cast<size_t>(FV) + static_cast<size_t>(s.width) * d) * kWave;
        const int32_t *tc = cb + (1 + static_cast<size_t>(s.width)) * kWave;
        auto V = [&](int slot, int ln) { return vb[static_cast<size_t>(slot) * kWave + ln]; };
        const int P = s.aux0 + lane, np = L.nl_poses;
        const int64_t own_row = L.rot_base + static_cast<int64_t>(P) * d;
        const int64_t nxt_row = L.rot_base + static_cast<int64_t>(std::min(P + 1, np - 1)) * d;
        const int64_t prv_row = L.rot_base + static_cast<int64_t>(std::max(P - 1, 0)) * d;
        const int64_t t_own = L.trn_base + P, t_nxt = L.trn_base + std::min(P + 1, np - 1), t_prv = L.trn_base + std::max(P - 1, 0);
        double *acct = &acc[static_cast<size_t>(d) * ld];
        for (int a = 0; a <= d; ++a) {
          axpy(&acc[static_cast<size_t>(a) * ld], V(a, lane), t_own);
          axpy(&acc[static_cast<size_t>(a) * ld], V(d + 1 + a, lane), t_nxt);
        }
        if (P > 0) {
          const double *hv = &F.head_val[static_cast<size_t>(s.aux0 / kWave) * HV];
          axpy(acct, lane > 0 ? V(d + 1 + d, lane - 1) : hv[d * d + d], t_prv);
          for (int c = 0; c < d; ++c) {
            for (int a = 0; a < d; ++a)
              axpy(&acc[static_cast<size_t>(a) * ld], lane > 0 ? V(2 * (d + 1) + a * d + c, lane - 1) : hv[a * d + c], prv_row + c);
            axpy(acct, lane > 0 ? V(d + 1 + c, lane - 1) : hv[d * d + c], prv_row + c);
          }
        }
        for (int c = 0; c < d; ++c) {
          for (int a = 0; a < d; ++a) {
            axpy(&acc[static_cast<size_t>(a) * ld], V(2 * (d + 1) + c * d + a, lane), nxt_row + c);
            axpy(&acc[static_cast<size_t>(a) * ld], V(2 * (d + 1) + d * d + c * d + a, lane), own_row + c);
          }
          axpy(acct, V(c, lane), own_row + c);
        }
        for (int k = 0; k < s.width; ++k)
          for (int a = 0; a < d; ++a)
            axpy(&acc[static_cast<size_t>(a) * ld], V(FV + k * d + a, lane), cb[(1 + static_cast<size_t>(k)) * kWave + lane]);
        const uint32_t info = static_cast<uint32_t>(cb[lane]);
        for (uint32_t e = info & 0xffffu; e < (info & 0xffffu) + ((info >> 16) & 0x7fu); ++e) {  // pairs of entries
          axpy(acct, tv[2 * e], tc[2 * e]);
          axpy(acct, tv[2 * e + 1], tc[2 * e + 1]);
        }
        for (int a = 0; a < d; ++a)
          for (int c = 0; c < ld; ++c) out[static_cast<size_t>(own_row + a) * ld + c] = acc[a * ld + c];
        for (int c = 0; c < ld; ++c) out[static_cast<size_t>(t_own) * ld + c] = acct[c];
        continue;
      }
      if ((s.type & kSliceTypeMask) == kSliceStiefel) {
        for (int k = 0; k < s.width; ++k) {
          const double *xr = X + static_cast<size_t>(F.scol[s.coff + static_cast<size_t>(k) * kWave + lane]) * ld;
          for (int a = 0; a < d; ++a) {
            const double v = F.sval[s.off + (static_cast<size_t>(k) * d + a) * kWave + lane];
            for (int c = 0; c < ld; ++c) acc[a * ld + c] += v * xr[c];
          }
        }
        for (int a = 0; a < d; ++a)
          for (int c = 0; c < ld; ++c)
            out[(static_cast<size_t>(s.row0) + static_cast<size_t>(lane) * d + a) * ld + c] = acc[a * ld + c];
        continue;
      }
      for (int k = 0; k < s.width; ++k) {
        const double v = F.sval[s.off + static_cast<size_t>(k) * kWave + lane];
        const double *xr = X + static_cast<size_t>(F.scol[s.coff + static_cast<size_t>(k) * kWave + lane]) * ld;
        for (int c = 0; c < ld; ++c) acc[c] += v * xr[c];
      }
      const int64_t row = (s.type == kSliceEuclidPerm) ? F.perm[s.row0 + lane] : s.row0 + lane;
      for (int c = 0; c < ld; ++c) out[static_cast<size_t>(row) * ld + c] = acc[c];
    }
  }
  size_t ci = 0;
  while (ci < F.chunks.size()) {
    const LongChunk &c0 = F.chunks[ci];
    std::fill(acc.begin(), acc.end(), 0.0);
    for (int c = 0; c < c0.nchunks; ++c) {
      const LongChunk &ch = F.chunks[ci + c];
      for (int32_t k = ch.k0; k < ch.k1; ++k) {
        const double *xr = X + static_cast<size_t>(F.lcol[k]) * ld;
        for (int j = 0; j < ld; ++j) acc[j] += F.lval[k] * xr[j];
      }
    }
    for (int j = 0; j < ld; ++j) out[static_cast<size_t>(c0.row) * ld + j] = acc[j];
    ci += c0.nchunks;
  }
}

}  // namespace cora
