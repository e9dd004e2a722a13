#include "trisolve.h"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>

namespace cora {

namespace {
constexpr int kBorderRowNnz = 4096;  // rows of L longer than this (landmarks) always belong to the last stage
constexpr int kFirstCap = 64;        // rows of a stage-0 subtree: what one wavefront holds (a pair of
                                     // nested-dissection leaves, their separator and range rows); 32 / 40 / 48 /
                                     // 64 measured 154 / 155 / 162 / 151 us per apply at 10^5 poses
constexpr int kCapGrowth = 16;       // stage k subtrees hold up to kFirstCap * kCapGrowth^k rows
constexpr int kTopCap = 1536;        // stop cutting once this few rows are left: they form the last stage
int64_t kTopInverseNnz = 2500000;  // ... or once the inverse of what is left has this few entries:
                                             // a launch floor (~5 us) is worth ~25 MB of traffic, so small
                                             // factors are applied as ONE explicit inverse (2 products)
constexpr int kShortRow = 64;        // entries: <= this -> 8 lanes per row
constexpr int kWaveRow = 1024;       // entries: <= this -> one wavefront per row, else chunked
constexpr int kChunk = 512;
constexpr int kDenseBlock = 64;      // stage-0 blocks up to this many rows use the dense wavefront kernel
constexpr int kMinBlock = 8;         // smaller subtrees are left to the next stage (a wavefront per block would idle)

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
                    const std::vector<int32_t> *group) {
  if (const char *e = std::getenv("CORA_TRI_TOP_INV")) kTopInverseNnz = std::atoll(e);
  P = TriPlan();
  P.m = m;
  P.zero_row = zero_row;
  if (m <= 0) return;
  P.nnzL = Lp[m];
  // ---- elimination tree and row lengths
  std::vector<int32_t> parent(m, -1), rcount(m, 0);
  bool sorted = true;  // columns with ascending row indices: the rows of a column's own block come first
  for (int j = 0; j < m; ++j) {
    if (Li[Lp[j]] != j) throw std::runtime_error("cora: Cholesky factor must store the diagonal first in each column");
    for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) {
      if (Li[q] <= j || Li[q] >= m) throw std::runtime_error("cora: Cholesky factor has an entry above the diagonal");
      if (parent[j] < 0 || Li[q] < parent[j]) parent[j] = Li[q];
      if (q > Lp[j] + 1 && Li[q] < Li[q - 1]) sorted = false;
      rcount[Li[q]]++;
    }
  }
  for (int i = 0; i < m; ++i)
    if (row_of[i] < 0) throw std::runtime_error("cora: factor row outside the handle");
  // ---- CSR of the strictly lower part (rows of L, columns ascending)
  std::vector<int32_t> rptr(static_cast<size_t>(m) + 1, 0);
  for (int i = 0; i < m; ++i) rptr[i + 1] = rptr[i] + rcount[i];
  std::vector<int32_t> rcol(static_cast<size_t>(rptr[m]));
  std::vector<double> rval(static_cast<size_t>(rptr[m]));
  {
    std::vector<int32_t> fill(rptr.begin(), rptr.end() - 1);
    for (int j = 0; j < m; ++j)
      for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) {
        rcol[fill[Li[q]]] = j;
        rval[fill[Li[q]]++] = Lx[q];
      }
  }
  // ---- stages: repeatedly peel the maximal subtrees of the remaining forest that fit the cap
  int first_border = m;  // trailing run of long rows: forced into the last stage
  while (first_border > 0 && rcount[first_border - 1] > kBorderRowNnz) --first_border;
  std::vector<int32_t> stage(m, -1), blk(m, -1), sz(m, 0);
  int nstage = 0;
  int64_t remaining = first_border, cap = kFirstCap;
  std::vector<int32_t> depth(m, 0);
  while (remaining + (m - first_border) > kTopCap && cap < 4LL * m) {
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
  bool dense0 = K > 1;
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
  // ---- "a" products: the couplings between stages
  for (int k = 0; k < K; ++k) {
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
  // ---- "b" products: explicit inverse of every diagonal block.  Column j of W = L_bb^-1 solves
  // L_bb w = e_j and is non-zero only on the path from j to the root of its block.
  std::vector<double> w(static_cast<size_t>(m), 0.0);
  std::vector<std::vector<int32_t>> wt_row(static_cast<size_t>(K)), wt_col(static_cast<size_t>(K));
  std::vector<std::vector<double>> wt_val(static_cast<size_t>(K));
  std::vector<RowList> bb(static_cast<size_t>(K));
  for (int j = 0; j < m; ++j) {
    const int k = stage[j], b = blk[j];
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
  for (int k = 0; k < K; ++k) {
    if (k == 0 && dense0) continue;
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
    RowList F;
    for (int i = 0; i < m; ++i) {
      if (stage[i] != k) continue;
      F.begin_row(row_of[i]);
      for (int32_t q = cnt[i]; q < cnt[i + 1]; ++q) F.add(row_of[cc[q]], vv[q]);
      F.end_row();
    }
    finalize(F, P.stages[k].fwd_b);
    wt_row[k] = std::vector<int32_t>();
    wt_col[k] = std::vector<int32_t>();
    wt_val[k] = std::vector<double>();
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
}  // namespace

void tri_plan_solve_host(const TriPlan &P, int64_t rows, const double *rhs, double *out) {
  const int K = static_cast<int>(P.stages.size());
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
