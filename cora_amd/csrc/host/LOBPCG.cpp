#include "LOBPCG.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../../../include/cora_hip.h"
#include "dense.h"

namespace CORA {

namespace {

struct Blocks {
  cora_ctx *c;
  int m;
  std::vector<double *> owned;
  Blocks(cora_ctx *ctx, int m_) : c(ctx), m(m_) {}
  ~Blocks() {
    for (double *p : owned) cora_dev_free(c, p);
  }
  void chk(int rc, const char *w) const {
    if (rc != CORA_OK) throw std::runtime_error(std::string("LOBPCG: ") + w + ": " + cora_last_error(c));
  }
  double *alloc() {
    double *p = nullptr;
    chk(cora_dev_alloc(c, m, &p), "alloc");
    owned.push_back(p);
    return p;
  }
  Matrix gram(const double *a, const double *b) const {
    Matrix G(m, m);
    chk(cora_gram_dev(c, a, m, b, m, G.data()), "gram");
    return G;
  }
  // out = sum_i X_i C_i  (C_i: m x m host matrices)
  void combine(std::vector<const double *> xs, const std::vector<Matrix> &Cs, double *out) const {
    std::vector<int> k(xs.size(), m);
    std::vector<const double *> cp;
    for (const Matrix &C : Cs) cp.push_back(C.data());
    chk(cora_combine_dev(c, static_cast<int>(xs.size()), xs.data(), k.data(), cp.data(), m, out), "combine");
  }
};

Matrix sub(const Matrix &M, Index r0, Index c0, Index nr, Index nc) { return M.block(r0, c0, nr, nc); }

}  // namespace

LOBPCGSolver::LOBPCGSolver(cora_ctx *ctx, int N) : c_(ctx), N_(N) {}

LOBPCGSolver::~LOBPCGSolver() {
  for (double *p : owned_) cora_dev_free(c_, p);
}

Matrix LOBPCGSolver::block() const {
  if (!X_) throw std::logic_error("LOBPCGSolver::block: no run yet");
  Matrix X(N_, m_);
  if (cora_download(c_, X_, m_, X.data(), N_) != CORA_OK) throw std::runtime_error(std::string("LOBPCG: download: ") + cora_last_error(c_));
  return X;
}

Vector LOBPCGSolver::column(int j) const {
  if (!X_) throw std::logic_error("LOBPCGSolver::column: no run yet");
  if (j < 0 || j >= m_) throw std::invalid_argument("LOBPCGSolver::column: index out of range");
  // X e_j on the device (1.0 * x + 0.0 * the rest: exact), one column over the bus
  double *col = nullptr;
  if (cora_dev_alloc(c_, 1, &col) != CORA_OK) throw std::runtime_error(std::string("LOBPCG: alloc: ") + cora_last_error(c_));
  Vector v(N_, 1);
  Matrix e(m_, 1);
  e(j, 0) = 1.0;
  const double *xs[1] = {X_};
  const int ks[1] = {m_};
  const double *cs[1] = {e.data()};
  int rc = cora_combine_dev(c_, 1, xs, ks, cs, 1, col);
  if (rc == CORA_OK) rc = cora_download(c_, col, 1, v.data(), N_);
  cora_dev_free(c_, col);
  if (rc != CORA_OK) throw std::runtime_error(std::string("LOBPCG: column: ") + cora_last_error(c_));
  return v;
}

LOBPCGResult LOBPCG(cora_ctx *c, const DeviceOperator &A, const std::optional<DeviceOperator> &T, const Matrix &X0,
                    size_t nev, size_t max_iters, Scalar tau, const std::optional<LOBPCGStop> &stop) {
  LOBPCGSolver S(c, static_cast<int>(X0.rows()));
  return S.run(A, T, {HostColumns{X0.data(), static_cast<int>(X0.cols())}}, nev, max_iters, tau, stop, true);
}

LOBPCGResult LOBPCGSolver::run(const DeviceOperator &A, const std::optional<DeviceOperator> &T,
                               const std::vector<HostColumns> &start, size_t nev, size_t max_iters, Scalar tau,
                               const std::optional<LOBPCGStop> &stop, bool download) {
  cora_ctx *c = c_;
  const int N = N_;
  int m = 0;
  for (const HostColumns &h : start) m += h.cols;
  if (m < 1 || m > 24) throw std::invalid_argument("LOBPCG: block size must be in [1, 24]");
  if (static_cast<size_t>(m) < nev) throw std::invalid_argument("LOBPCG: block smaller than nev");
  if (start.size() > 4) throw std::invalid_argument("LOBPCG: at most four pieces in the start block");
  const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "      [lobpcg] %-22s %.4f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  for (double *p : owned_) cora_dev_free(c_, p);
  owned_.clear();
  m_ = m;
  X_ = nullptr;
  Blocks B(c, m);
  double *X = B.alloc(), *AX = B.alloc(), *W = B.alloc(), *AW = B.alloc(), *P = B.alloc(), *AP = B.alloc(),
         *t1 = B.alloc(), *t2 = B.alloc(), *t3 = B.alloc(), *t4 = B.alloc();
  if (start.size() == 1 && !start[0].device) {
    B.chk(cora_upload(c, start[0].data, N, m, t1), "upload");
  } else {
    // pieces side by side: Out = sum_i piece_i [0 .. I .. 0]
    struct Pieces {  // the pieces' device buffers go back whatever happens in the loop (round-3 advice)
      cora_ctx *c;
      std::vector<double *> v;
      ~Pieces() {
        for (double *d : v) cora_dev_free(c, d);
      }
    } dev{c, {}};
    std::vector<const double *> xs;
    std::vector<int> ks;
    std::vector<Matrix> sel;
    int at = 0;
    for (const HostColumns &h : start) {
      if (h.device) {  // already resident: its leading columns are picked by the selection matrix
        if (h.device_width < h.cols || h.device_width > 24) throw std::invalid_argument("LOBPCG: bad resident piece");
        Matrix E(h.device_width, m);
        for (int j = 0; j < h.cols; ++j) E(j, at + j) = 1.0;
        sel.push_back(E);
        xs.push_back(h.device);
        ks.push_back(h.device_width);
        at += h.cols;
        continue;
      }
      double *d = nullptr;
      B.chk(cora_dev_alloc(c, h.cols, &d), "alloc");
      dev.v.push_back(d);
      B.chk(cora_upload(c, h.data, N, h.cols, d), "upload");
      Matrix E(h.cols, m);
      for (int j = 0; j < h.cols; ++j) E(j, at + j) = 1.0;
      sel.push_back(E);
      xs.push_back(d);
      ks.push_back(h.cols);
      at += h.cols;
    }
    std::vector<const double *> cp;
    for (const Matrix &E : sel) cp.push_back(E.data());
    B.chk(cora_combine_dev(c, static_cast<int>(xs.size()), xs.data(), ks.data(), cp.data(), m, t1), "combine");
  }
  tick("upload of the block");

  // X <- orthonormal basis of span(X0): X0 V D^-1/2
  {
    // (columns scaled to unit length first: the start block mixes columns of very different size -- at 10^6 poses the
    // iterate's translation rows give columns of norm 1e8 next to a unit seed vector, and the Gram matrix's small
    // eigenvalue drowned in the rounding of the large ones)
    Matrix G = B.gram(t1, t1);
    std::vector<double> dsc(static_cast<size_t>(m));
    for (int j = 0; j < m; ++j) {
      if (!(G(j, j) > 0.0)) throw std::runtime_error("LOBPCG: initial block is rank deficient");
      dsc[j] = 1.0 / std::sqrt(G(j, j));
    }
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) G(i, j) *= dsc[i] * dsc[j];
    Vector ev;
    Matrix V;
    symmetricEigen(G, ev, V);
    Matrix C(m, m);
    for (int j = 0; j < m; ++j) {
      if (!(ev(j) > 1e-14 * ev(m - 1))) throw std::runtime_error("LOBPCG: initial block is rank deficient");
      for (int i = 0; i < m; ++i) C(i, j) = dsc[i] * V(i, j) / std::sqrt(ev(j));
    }
    B.combine({t1}, {C}, X);
  }
  A(X, m, AX);
  std::vector<Scalar> theta(static_cast<size_t>(m));
  {  // initial Rayleigh-Ritz
    Vector ev;
    Matrix V;
    Matrix H = B.gram(X, AX);
    for (int i = 0; i < m; ++i)
      for (int j = i + 1; j < m; ++j) H(i, j) = H(j, i) = 0.5 * (H(i, j) + H(j, i));
    symmetricEigen(H, ev, V);
    B.combine({X}, {V}, t1);
    B.combine({AX}, {V}, t2);
    std::swap(X, t1);
    std::swap(AX, t2);
    for (int i = 0; i < m; ++i) theta[i] = ev(i);
  }
  tick("start: 2 products' worth");
  bool haveP = false;
  LOBPCGResult res;
  size_t it = 0;
  for (; it < max_iters; ++it) {
    if (stop && (*stop)(it, theta, X, m)) break;
    // R = AX - X diag(theta)   (into W)
    Matrix D(m, m), I = Matrix::Identity(m, m);
    for (int i = 0; i < m; ++i) D(i, i) = -theta[i];
    B.combine({AX, X}, {I, D}, t1);
    // residual norms of the wanted pairs (without a preconditioner W is R itself: X' W is asked for in the same trip
    // -- the same kernels on the same operands, one wait fewer per iteration)
    Matrix RR(m, m), XtW_early(m, m);
    const bool early = !T;
    if (early) {
      const double *ga[2] = {t1, X}, *gb[2] = {t1, t1};
      const int km[2] = {m, m};
      double *out[2] = {RR.data(), XtW_early.data()};
      B.chk(cora_gram_batch_dev(c, 2, ga, km, gb, km, out), "gram batch");
    } else {
      RR = B.gram(t1, t1);
    }
    size_t nconv = 0;
    for (size_t k = 0; k < nev; ++k)
      if (std::sqrt(std::max(RR(k, k), 0.0)) <= tau * std::max(std::abs(theta[k]), 1e-300)) ++nconv;
    res.num_converged = nconv;
    if (tau > 0 && nconv == nev) break;
    if (T) (*T)(t1, m, W);
    else std::swap(W, t1);
    // W <- (I - X X^T) W, then orthonormalise W
    {
      Matrix XtW = early ? XtW_early : B.gram(X, W);
      for (Index i = 0; i < XtW.size(); ++i) XtW.data()[i] = -XtW.data()[i];
      B.combine({W, X}, {I, XtW}, t1);
      Matrix G = B.gram(t1, t1);  // columns scaled to unit length, as above (a vanished column stays out)
      std::vector<double> dsc(static_cast<size_t>(m));
      for (int j = 0; j < m; ++j) dsc[j] = G(j, j) > 0.0 ? 1.0 / std::sqrt(G(j, j)) : 0.0;
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) G(i, j) *= dsc[i] * dsc[j];
      Vector ev;
      Matrix V;
      symmetricEigen(G, ev, V);
      Matrix C(m, m);
      const double top = std::max(ev(m - 1), 1e-300);
      for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) C(i, j) = ev(j) > 1e-20 * top ? dsc[i] * V(i, j) / std::sqrt(ev(j)) : 0.0;
      B.combine({t1}, {C}, W);
    }
    A(W, m, AW);
    // Rayleigh-Ritz on S = [X W P]
    const int nb = haveP ? 3 : 2, ns = nb * m;
    const double *S[3] = {X, W, P}, *AS[3] = {AX, AW, AP};
    Matrix GA(ns, ns), GB(ns, ns);
    std::vector<Matrix> grams;  // (S_a' A S_b, S_a' S_b) for a <= b: one trip to the device for all of them
    {
      std::vector<const double *> ga_, gb_;
      for (int a = 0; a < nb; ++a)
        for (int b = a; b < nb; ++b) {
          ga_.push_back(S[a]); gb_.push_back(AS[b]);
          ga_.push_back(S[a]); gb_.push_back(S[b]);
        }
      grams.assign(ga_.size(), Matrix(m, m));
      std::vector<int> km(ga_.size(), m);
      std::vector<double *> out;
      for (Matrix &G : grams) out.push_back(G.data());
      B.chk(cora_gram_batch_dev(c, static_cast<int>(ga_.size()), ga_.data(), km.data(), gb_.data(), km.data(), out.data()),
            "gram batch");
    }
    size_t gi = 0;
    for (int a = 0; a < nb; ++a)
      for (int b = a; b < nb; ++b) {
        const Matrix &ga = grams[gi], &gb = grams[gi + 1];
        gi += 2;
        for (int i = 0; i < m; ++i)
          for (int j = 0; j < m; ++j) {
            GA(a * m + i, b * m + j) = ga(i, j);
            GA(b * m + j, a * m + i) = ga(i, j);
            GB(a * m + i, b * m + j) = gb(i, j);
            GB(b * m + j, a * m + i) = gb(i, j);
          }
      }
    for (int i = 0; i < ns; ++i)
      for (int j = i + 1; j < ns; ++j) {
        GA(i, j) = GA(j, i) = 0.5 * (GA(i, j) + GA(j, i));
        GB(i, j) = GB(j, i) = 0.5 * (GB(i, j) + GB(j, i));
      }
    // generalised problem GA c = theta GB c through an orthonormal basis of GB's range
    Vector eb;
    Matrix Vb;
    symmetricEigen(GB, eb, Vb);
    std::vector<int> keep;
    for (int j = 0; j < ns; ++j)
      if (eb(j) > 1e-10 * eb(ns - 1)) keep.push_back(j);
    const int nk = static_cast<int>(keep.size());
    if (nk < m) break;  // basis collapsed: converged to machine precision
    Matrix Z(ns, nk);
    for (int jj = 0; jj < nk; ++jj)
      for (int i = 0; i < ns; ++i) Z(i, jj) = Vb(i, keep[jj]) / std::sqrt(eb(keep[jj]));
    Matrix Hr = Z.transpose() * GA * Z;
    for (int i = 0; i < nk; ++i)
      for (int j = i + 1; j < nk; ++j) Hr(i, j) = Hr(j, i) = 0.5 * (Hr(i, j) + Hr(j, i));
    Vector er;
    Matrix Vr;
    symmetricEigen(Hr, er, Vr);
    const Matrix C = Z * sub(Vr, 0, 0, nk, m);  // ns x m coefficients of the new X
    std::vector<Matrix> Cb;
    for (int a = 0; a < nb; ++a) Cb.push_back(sub(C, a * m, 0, m, m));
    // new P = [W P] C_wp,  new X = X C_x + new P
    {
      std::vector<const double *> s, as;
      std::vector<Matrix> cs;
      for (int a = 1; a < nb; ++a) { s.push_back(S[a]); as.push_back(AS[a]); cs.push_back(Cb[a]); }
      B.combine(s, cs, t1);   // new P
      B.combine(as, cs, t2);  // new AP
    }
    B.combine({X, t1}, {Cb[0], I}, t3);    // new X
    B.combine({AX, t2}, {Cb[0], I}, t4);   // new AX
    std::swap(P, t1);
    std::swap(AP, t2);
    std::swap(X, t3);
    std::swap(AX, t4);
    haveP = true;
    for (int i = 0; i < m; ++i) theta[i] = er(i);
  }
  tick("iterations");
  res.num_iters = it;
  res.Theta = Vector(m, 1);
  for (int i = 0; i < m; ++i) res.Theta(i) = theta[i];
  // the blocks live on in this object (B gives up its ownership)
  owned_ = std::move(B.owned);
  B.owned.clear();
  X_ = X;
  if (download) {
    res.X = block();
    tick("download of the block");
  }
  return res;
}

}  // namespace CORA
