#include "trisolve.h"

#include <algorithm>
#include <stdexcept>

namespace cora {

namespace {
constexpr int kBorderRowNnz = 4096;   // rows of L longer than this form the dense border
constexpr int kBorderChunk = 2048;

// lanes per supernode from the largest number of external entries of a supernode in the level
int lanes_for(int max_len) { return max_len <= 96 ? 8 : 64; }
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

  // ---- small chain supernodes over the non-border rows.  parent(j) = first off-diagonal row of column j.
  const int nr = first_border;
  std::vector<int32_t> parent(nr, -1), nchild(nr, 0);
  for (int j = 0; j < nr; ++j)
    if (Lp[j] + 1 < Lp[j + 1] && Li[Lp[j] + 1] < nr) parent[j] = Li[Lp[j] + 1];
  for (int j = 0; j < nr; ++j)
    if (parent[j] >= 0) nchild[parent[j]]++;
  std::vector<int32_t> sn_of(nr, 0), sn_first;
  for (int i = 0; i < nr; ++i) {
    const bool extend = i > 0 && parent[i - 1] == i && nchild[i] == 1 && (i - sn_first.back()) < kTriSn;
    if (!extend) sn_first.push_back(i);
    sn_of[i] = static_cast<int32_t>(sn_first.size()) - 1;
  }
  const int nsn = static_cast<int>(sn_first.size());
  sn_first.push_back(nr);

  auto emit = [&](TriHost &T, bool backward, auto row_begin, auto row_end, auto col_at, auto val_at) {
    std::vector<int32_t> lev(nsn, 0);
    int height = 0;
    auto visit = [&](int s) {
      int l = 0;
      for (int i = sn_first[s]; i < sn_first[s + 1]; ++i)
        for (int32_t q = row_begin(i); q < row_end(i); ++q) {
          const int j = col_at(q);
          if (j < nr && sn_of[j] != s) l = std::max(l, lev[sn_of[j]] + 1);
        }
      lev[s] = l;
      height = std::max(height, l + 1);
    };
    if (!backward) for (int s = 0; s < nsn; ++s) visit(s);
    else for (int s = nsn - 1; s >= 0; --s) visit(s);
    std::vector<std::vector<int32_t>> by_level(height);
    for (int s = 0; s < nsn; ++s) by_level[lev[s]].push_back(s);
    for (int l = 0; l < height; ++l) {
      TriLevel tl;
      tl.begin = static_cast<int32_t>(T.sn.size());
      int maxlen = 0;
      for (int32_t s : by_level[l]) {
        const int lo = sn_first[s], hi = sn_first[s + 1], bsz = hi - lo;
        TriSn R{};
        R.ext_begin = static_cast<int32_t>(T.cols.size());
        R.nrows = bsz;
        auto pos = [&](int i) { return backward ? hi - 1 - i : i - lo; };  // processing position of row i
        for (int t = 0; t < kTriSn; ++t) { R.out_row[t] = row_of[backward ? hi - 1 : lo]; R.dinv[t] = 0.0; }
        for (int t = 0; t < bsz; ++t) {
          const int i = backward ? hi - 1 - t : lo + t;
          for (int32_t q = row_begin(i); q < row_end(i); ++q) {
            const int j = col_at(q);
            if (j < nr && sn_of[j] == s) {
              const int pq = pos(j);
              if (pq >= t) throw std::logic_error("cora: supernode dependency out of order");
              R.lint[t * (t - 1) / 2 + pq] += val_at(q);
            } else {
              if (row_of[j] >= (1 << 28)) throw std::runtime_error("cora: too many rows for the packed triangular-solve index");
              T.cols.push_back(row_of[j] | (t << 28));  // position of the row inside its supernode in the top bits
              T.vals.push_back(val_at(q));
            }
          }
          R.out_row[t] = row_of[i];
          R.dinv[t] = 1.0 / Lx[Lp[i]];
        }
        R.ext_end = static_cast<int32_t>(T.cols.size());
        maxlen = std::max(maxlen, R.ext_end - R.ext_begin);
        T.sn.push_back(R);
      }
      tl.end = static_cast<int32_t>(T.sn.size());
      tl.lanes = lanes_for(maxlen);
      T.levels.push_back(tl);
    }
    return height;
  };
  const int hf = emit(P.fwd, false, [&](int i) { return rptr[i]; }, [&](int i) { return rptr[i + 1]; },
                      [&](int32_t q) { return rcol[q]; }, [&](int32_t q) { return rval[q]; });
  const int hb = emit(P.bwd, true, [&](int j) { return Lp[j] + 1; }, [&](int j) { return Lp[j + 1]; },
                      [&](int32_t q) { return Li[q]; }, [&](int32_t q) { return Lx[q]; });
  P.height = std::max(hf, hb);
}

}  // namespace cora
