#include "trisolve.h"

#include <algorithm>
#include <stdexcept>

namespace cora {

namespace {
constexpr int kBorderRowNnz = 4096;   // rows of L longer than this form the dense border
constexpr int kBorderChunk = 2048;

int lanes_for(int max_len) { return max_len <= 12 ? 1 : (max_len <= 192 ? 8 : 64); }
}  // namespace

void build_tri_plan(int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                    const std::vector<int32_t> &row_of, TriPlan &P) {
  P = TriPlan();
  P.m = m;
  P.nnzL = Lp[m];
  // ---- row counts of L (strictly lower part)
  std::vector<int32_t> rcount(m, 0);
  for (int j = 0; j < m; ++j) {
    if (Li[Lp[j]] != j) throw std::runtime_error("cora: Cholesky factor must store the diagonal first in each column");
    for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) rcount[Li[q]]++;
  }
  // ---- border: maximal trailing run of long rows
  int first_border = m;
  while (first_border > 0 && rcount[first_border - 1] > kBorderRowNnz) --first_border;
  const int nb = m - first_border;
  BorderHost &B = P.border;
  B.nb = nb;
  B.Lbb.assign(static_cast<size_t>(nb) * nb, 0.0);
  for (int k = 0; k < nb; ++k) B.out_row.push_back(row_of[first_border + k]);

  // ---- CSR of the strictly lower part restricted to non-border rows (forward, pull)
  std::vector<int32_t> rptr(first_border + 1, 0);
  for (int i = 0; i < first_border; ++i) rptr[i + 1] = rptr[i] + rcount[i];
  std::vector<int32_t> rcol(rptr[first_border]);
  std::vector<double> rval(rptr[first_border]);
  std::vector<int32_t> fill(rptr.begin(), rptr.end() - 1);
  // border rows: W part (columns < first_border) collected per row
  std::vector<std::vector<int32_t>> wc(nb);
  std::vector<std::vector<double>> wv(nb);
  for (int j = 0; j < m; ++j) {
    for (int32_t q = Lp[j] + (0); q < Lp[j + 1]; ++q) {
      const int i = Li[q];
      if (i >= first_border) {
        const int k = i - first_border;
        if (j >= first_border) B.Lbb[static_cast<size_t>(k) * nb + (j - first_border)] = Lx[q];
        else { wc[k].push_back(row_of[j]); wv[k].push_back(Lx[q]); }
      } else if (i != j) {
        rcol[fill[i]] = j;
        rval[fill[i]] = Lx[q];
        fill[i]++;
      }
    }
  }
  for (int k = 0; k < nb; ++k) {
    const int len = static_cast<int>(wc[k].size());
    for (int c0 = 0; c0 < len || (c0 == 0 && len == 0); c0 += kBorderChunk) {
      B.chunk_row.push_back(k);
      B.chunk_begin.push_back(static_cast<int32_t>(B.wcols.size()) + c0);
      B.chunk_end.push_back(static_cast<int32_t>(B.wcols.size()) + std::min(len, c0 + kBorderChunk));
      if (len == 0) break;
    }
    B.wcols.insert(B.wcols.end(), wc[k].begin(), wc[k].end());
    B.wvals.insert(B.wvals.end(), wv[k].begin(), wv[k].end());
  }

  // ---- forward levels (non-border rows): level = 1 + max level of dependencies
  std::vector<int32_t> lev(first_border, 0);
  int height = 0;
  for (int i = 0; i < first_border; ++i) {
    int l = 0;
    for (int32_t q = rptr[i]; q < rptr[i + 1]; ++q) l = std::max(l, lev[rcol[q]] + 1);
    lev[i] = l;
    height = std::max(height, l + 1);
  }
  auto emit = [&](TriHost &T, const std::vector<int32_t> &level_of, int nlev, auto row_begin, auto row_end,
                  auto col_at, auto val_at, auto diag_of, bool descending) {
    std::vector<std::vector<int32_t>> rows(nlev);
    for (int i = 0; i < first_border; ++i) rows[level_of[i]].push_back(i);
    T.rowptr.assign(1, 0);
    for (int l = 0; l < nlev; ++l) {
      const int L = descending ? nlev - 1 - l : l;
      if (rows[L].empty()) continue;
      TriLevel tl;
      tl.begin = static_cast<int32_t>(T.out_row.size());
      int maxlen = 0;
      for (int32_t i : rows[L]) {
        const int32_t b = row_begin(i), e = row_end(i);
        maxlen = std::max(maxlen, e - b);
        for (int32_t q = b; q < e; ++q) {
          T.cols.push_back(row_of[col_at(q)]);
          T.vals.push_back(val_at(q));
        }
        T.rowptr.push_back(static_cast<int32_t>(T.cols.size()));
        T.out_row.push_back(row_of[i]);
        T.dinv.push_back(1.0 / diag_of(i));
      }
      tl.end = static_cast<int32_t>(T.out_row.size());
      tl.lanes = lanes_for(maxlen);
      T.levels.push_back(tl);
    }
  };
  emit(P.fwd, lev, height, [&](int i) { return rptr[i]; }, [&](int i) { return rptr[i + 1]; },
       [&](int32_t q) { return rcol[q]; }, [&](int32_t q) { return rval[q]; },
       [&](int i) { return Lx[Lp[i]]; }, false);

  // ---- backward (L^T x = y, pull over the columns of L): x_j needs x_i for the
  // rows i > j of column j; border rows are solved first, so they count as level -1.
  std::vector<int32_t> blev(first_border, 0);
  int bheight = 0;
  for (int j = first_border - 1; j >= 0; --j) {
    int l = 0;
    for (int32_t q = Lp[j] + 1; q < Lp[j + 1]; ++q) {
      const int i = Li[q];
      if (i < first_border) l = std::max(l, blev[i] + 1);
    }
    blev[j] = l;
    bheight = std::max(bheight, l + 1);
  }
  emit(P.bwd, blev, bheight, [&](int j) { return Lp[j] + 1; }, [&](int j) { return Lp[j + 1]; },
       [&](int32_t q) { return Li[q]; }, [&](int32_t q) { return Lx[q]; },
       [&](int j) { return Lx[Lp[j]]; }, false);
  P.height = std::max(height, bheight);
}

}  // namespace cora
