#define _GNU_SOURCE
#include <sched.h>
/*
 * cora_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the CORA hot path
 * (MarineRoboticsGroup/cora @ 2025-10-17).  It is the parity checker for the
 * HIP kernels and the `cpu_baseline` leg of bench.py.  Nothing under
 * cora_amd/ may include, link or call this file: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below against the reference's own MatrixMarket golden vectors
 * (tests/golden/<case>/{DataMatrix,X_rand_dim2,rand_dX,expected_egrad,
 * expected_rgrad,hessProd,S_rand,X_gt}.mm, copied as data from the
 * reference's tests/data/) and the known-answer costs in the reference's
 * tests/test_utils.cpp:210-222.
 *
 * Conventions (same as the reference, include/CORA/CORA_types.h:43-70):
 *   dense matrices are column-major double with an explicit leading dimension,
 *   the sparse data matrix Q is row-major CSR with int32 indices.
 * Variable layout (include/CORA/CORA_problem.h:151-157):
 *   rows [0, d*n)          n stacked d x p Stiefel blocks  (Y_i Y_i^T = I_d)
 *   rows [d*n, d*n+r)      r unit rows (one sphere per range measurement)
 *   rows [d*n+r, N)        translations (poses, then landmarks)
 *
 * Each function cites the reference file:line it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads used by the row-parallel loops below (SpMM, tangent projection, Hessian-vector product).
 * The reference is single-threaded (its CMakeLists.txt never enables OpenMP): bench.py times 1 thread as
 * the reference-equivalent baseline and all cores as the second column of BASELINE.md section 4.  Every
 * row is computed by one thread in the same order, so results do not depend on the thread count. */
void orc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

#define AT(M, ld, i, j) ((M)[(size_t)(j) * (size_t)(ld) + (size_t)(i)])

/* ---- a1: Problem::dataMatrixProduct, Explicit branch ------------------
 * src/CORA_problem.cpp:742-746  (`return data_matrix_ * Y;`, Eigen
 * row-major sparse x column-major dense: one dot product per (row, col)). */
void orc_spmm(int N, const int32_t *rowptr, const int32_t *col,
              const double *val, const double *X, int ldx, int k, double *out,
              int ldo) {
  for (int c = 0; c < k; ++c) {
    const double *xc = X + (size_t)c * ldx;
    double *oc = out + (size_t)c * ldo;
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q)
        s += val[q] * xc[col[q]];
      oc[i] = s;
    }
  }
}

/* ---- NUMA placement for the all-core column of bench.py -----------------
 * Linux places a page on the memory node of the thread that touches it first.  The threaded loops above and below
 * share rows out with schedule(static), so copies made with the same schedule put every thread's rows of Q and of
 * the dense operands on its own node (the destinations must be untouched memory: oracle.py maps them fresh).
 * Without this, arrays filled by numpy sit on the master thread's node and a two-socket host runs the product out
 * of one socket's memory. */
/* Thread t of the OpenMP team runs on logical CPU cpus[t] (n = 0: every thread may run anywhere again, so that
 * threads created afterwards by other code do not inherit a one-core mask from the master thread). */
int orc_bind_threads(int n, const int *cpus) {
  int failed = 0;
#ifdef _OPENMP
#pragma omp parallel reduction(+ : failed)
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    const int t = omp_get_thread_num();
    if (n > 0) {
      CPU_SET(cpus[t % n], &set);
    } else {
      for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &set);
    }
    if (sched_setaffinity(0, sizeof(set), &set) != 0) failed += 1;
  }
#else
  (void)n;
  (void)cpus;
#endif
  return failed;
}

void orc_first_touch_csr(int N, const int32_t *rowptr, const int32_t *col,
                         const double *val, int32_t *col_dst, double *val_dst) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i)
    for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      col_dst[q] = col[q];
      val_dst[q] = val[q];
    }
}
void orc_first_touch_dense(int N, int k, const double *src, int lds, double *dst,
                           int ldd) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < N; ++i)
    for (int c = 0; c < k; ++c)
      dst[(size_t)c * ldd + i] = src ? src[(size_t)c * lds + i] : 0.0;
}

/* Row-at-a-time variant of the same product (all k columns of one row per
 * pass over the row's nonzeros).  Identical results up to summation order
 * (none: each (row, col) sum visits the nonzeros in the same order).  Used
 * by the CPU baseline because it streams Q once instead of k times. */
void orc_spmm_rowwise(int N, const int32_t *rowptr, const int32_t *col,
                      const double *val, const double *X, int ldx, int k,
                      double *out, int ldo) {
#pragma omp parallel for schedule(static) if (N > 20000)
  for (int i = 0; i < N; ++i) {
    double acc[64];
    for (int c = 0; c < k; ++c) acc[c] = 0.0;
    for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      const double v = val[q];
      const double *xr = X + col[q];
      for (int c = 0; c < k; ++c) acc[c] += v * xr[(size_t)c * ldx];
    }
    for (int c = 0; c < k; ++c) out[(size_t)c * ldo + i] = acc[c];
  }
}

/* ---- a11: metric closure, src/CORA.cpp:119-122 `(V1^T V2).trace()` ---- */
double orc_inner(int N, int p, const double *A, int lda, const double *B,
                 int ldb) {
  double tr = 0.0;
  for (int c = 0; c < p; ++c) {
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += AT(A, lda, i, c) * AT(B, ldb, i, c);
    tr += s;
  }
  return tr;
}

/* ---- a2: Problem::evaluateObjective, src/CORA_problem.cpp:759-762 ------
 * 0.5 * trace(Y^T (Q Y)).  `work` is N*p doubles. */
double orc_cost(int N, const int32_t *rowptr, const int32_t *col,
                const double *val, const double *Y, int ldy, int p,
                double *work) {
  orc_spmm(N, rowptr, col, val, Y, ldy, p, work, N);
  return 0.5 * orc_inner(N, p, Y, ldy, work, N);
}

/* ---- a5/a6/a7: Problem::tangent_space_projection ----------------------
 * src/CORA_problem.cpp:782-820;
 * StiefelProduct::projectToTangentSpace include/CORA/StiefelProduct.h:79-81
 *   = V - SymBlockDiagProduct(Y, Y^T, V)   (src/StiefelProduct.cpp:38-55)
 *   row layout:  V_i - sym(Y_i V_i^T) Y_i
 * ObliqueManifold::projectToTangentSpace src/ObliqueManifold.cpp:16-27
 *   row layout:  v_j - <y_j, v_j> y_j
 * translations: copied. `out` may alias V. */
void orc_tangent_proj(int d, int n, int r, int N, int p, const double *Y,
                      int ldy, const double *V, int ldv, double *out, int ldo) {
#pragma omp parallel for schedule(static) if (n > 5000)
  for (int i = 0; i < n; ++i) {
    double P[16], S[16], tmp[4 * 64];
    const int r0 = i * d;
    /* P = Y_i V_i^T  (d x d) */
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        double s = 0.0;
        for (int c = 0; c < p; ++c)
          s += AT(Y, ldy, r0 + a, c) * AT(V, ldv, r0 + b, c);
        P[a * 4 + b] = s;
      }
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) S[a * 4 + b] = 0.5 * (P[a * 4 + b] + P[b * 4 + a]);
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < p; ++c) {
        double s = 0.0;
        for (int b = 0; b < d; ++b) s += S[a * 4 + b] * AT(Y, ldy, r0 + b, c);
        tmp[a * 64 + c] = AT(V, ldv, r0 + a, c) - s;
      }
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < p; ++c) AT(out, ldo, r0 + a, c) = tmp[a * 64 + c];
  }
  const int dn = d * n;
#pragma omp parallel for schedule(static) if (r > 5000)
  for (int j = dn; j < dn + r; ++j) {
    double ip = 0.0;
    for (int c = 0; c < p; ++c) ip += AT(Y, ldy, j, c) * AT(V, ldv, j, c);
    for (int c = 0; c < p; ++c)
      AT(out, ldo, j, c) = AT(V, ldv, j, c) - ip * AT(Y, ldy, j, c);
  }
  for (int j = dn + r; j < N; ++j)
    for (int c = 0; c < p; ++c) AT(out, ldo, j, c) = AT(V, ldv, j, c);
}

/* ---- a12: Problem::compute_Lambda_blocks, src/CORA_problem.cpp:1105-1131
 * Lst: d x (d*n) column-major (ld = d), block i = sym((QY)_i Y_i^T);
 * lob: r, lob[j] = <y_j, (QY)_j>. */
void orc_lambda_blocks(int d, int n, int r, int p, const double *Y, int ldy,
                       const double *QY, int ldq, double *Lst, double *lob) {
  for (int i = 0; i < n; ++i) {
    const int r0 = i * d;
    double P[16];
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        double s = 0.0;
        for (int c = 0; c < p; ++c)
          s += AT(QY, ldq, r0 + a, c) * AT(Y, ldy, r0 + b, c);
        P[a * 4 + b] = s;
      }
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b)
        AT(Lst, d, a, r0 + b) = 0.5 * (P[a * 4 + b] + P[b * 4 + a]);
  }
  const int dn = d * n;
  for (int j = 0; j < r; ++j) {
    double s = 0.0;
    for (int c = 0; c < p; ++c)
      s += AT(Y, ldy, dn + j, c) * AT(QY, ldq, dn + j, c);
    lob[j] = s;
  }
}

/* Lambda * X for the block-diagonal multiplier
 * (Problem::compute_Lambda_from_Lambda_blocks, src/CORA_problem.cpp:1133-1160)
 * out = Lambda X; rows >= dn+r are zero. */
static void lambda_apply(int d, int n, int r, int N, int k, const double *Lst,
                         const double *lob, const double *X, int ldx,
                         double *out, int ldo) {
  for (int c = 0; c < k; ++c) {
    for (int i = 0; i < n; ++i) {
      const int r0 = i * d;
      for (int a = 0; a < d; ++a) {
        double s = 0.0;
        for (int b = 0; b < d; ++b)
          s += AT(Lst, d, a, r0 + b) * AT(X, ldx, r0 + b, c);
        AT(out, ldo, r0 + a, c) = s;
      }
    }
    const int dn = d * n;
    for (int j = 0; j < r; ++j)
      AT(out, ldo, dn + j, c) = lob[j] * AT(X, ldx, dn + j, c);
    for (int j = dn + r; j < N; ++j) AT(out, ldo, j, c) = 0.0;
  }
}

/* ---- certificate operator S X = Q X - Lambda X -------------------------
 * Problem::get_certificate_matrix src/CORA_problem.cpp:1162-1166 applied to
 * a block of vectors (the LOBPCG operator of src/CORA_utils.cpp:83).
 * `work` is N*k doubles. */
void orc_S_apply(int d, int n, int r, int N, const int32_t *rowptr,
                 const int32_t *col, const double *val, const double *Lst,
                 const double *lob, const double *X, int ldx, int k,
                 double *out, int ldo, double *work) {
  orc_spmm_rowwise(N, rowptr, col, val, X, ldx, k, out, ldo);
  lambda_apply(d, n, r, N, k, Lst, lob, X, ldx, work, N);
  for (int c = 0; c < k; ++c)
    for (int i = 0; i < N; ++i) AT(out, ldo, i, c) -= AT(work, N, i, c);
}

/* ---- a8: Problem::Riemannian_Hessian_vector_product --------------------
 * src/CORA_problem.cpp:822-867:
 *   H = Q Ydot                                                  (:835)
 *   Stiefel rows:  Proj_{Y_i}( H_i - sym(Y_i G_i^T) Ydot_i )     (:839-850)
 *   oblique rows:  Proj_{y_j}( h_j - <g_j, y_j> ydot_j )         (:853-864)
 *   translations:  H                                            (:866)
 * G = nablaF_Y (the cached Euclidean gradient Q Y).  work: N*p doubles. */
void orc_hvp(int d, int n, int r, int N, int p, const int32_t *rowptr,
             const int32_t *col, const double *val, const double *Y, int ldy,
             const double *G, int ldg, const double *Ydot, int ldd,
             double *out, int ldo, double *work) {
  const int dn = d * n;
  double *H = work; /* N x p, ld N */
  orc_spmm_rowwise(N, rowptr, col, val, Ydot, ldd, p, H, N);
#pragma omp parallel for schedule(static) if (n > 5000)
  for (int i = 0; i < n; ++i) {
    const int r0 = i * d;
    double P[16], S[16];
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) {
        double s = 0.0;
        for (int c = 0; c < p; ++c)
          s += AT(Y, ldy, r0 + a, c) * AT(G, ldg, r0 + b, c);
        P[a * 4 + b] = s;
      }
    for (int a = 0; a < d; ++a)
      for (int b = 0; b < d; ++b) S[a * 4 + b] = 0.5 * (P[a * 4 + b] + P[b * 4 + a]);
    double upd[4 * 64];
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < p; ++c) {
        double s = 0.0;
        for (int b = 0; b < d; ++b) s += S[a * 4 + b] * AT(Ydot, ldd, r0 + b, c);
        upd[a * 64 + c] = s;
      }
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < p; ++c) AT(H, N, r0 + a, c) -= upd[a * 64 + c];
  }
#pragma omp parallel for schedule(static) if (r > 5000)
  for (int j = dn; j < dn + r; ++j) {
    double w = 0.0;
    for (int c = 0; c < p; ++c) w += AT(G, ldg, j, c) * AT(Y, ldy, j, c);
    for (int c = 0; c < p; ++c) AT(H, N, j, c) -= w * AT(Ydot, ldd, j, c);
  }
  orc_tangent_proj(d, n, r, N, p, Y, ldy, H, N, out, ldo);
}

/* ---- one-sided Jacobi SVD of a small m x k matrix (m >= k) -------------
 * The reference calls Eigen::JacobiSVD (src/StiefelProduct.cpp:29-33); Eigen
 * is a system dependency that is not in /root/reference, so this restates the
 * published Hestenes one-sided Jacobi method.  A (m x k, col-major ld m) is
 * overwritten by U*Sigma; V (k x k col-major) receives the right vectors. */
static void jacobi_svd_small(int m, int k, double *A, double *V) {
  for (int a = 0; a < k; ++a)
    for (int b = 0; b < k; ++b) V[b * k + a] = (a == b) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int a = 0; a < k - 1; ++a)
      for (int b = a + 1; b < k; ++b) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < m; ++i) {
          alpha += A[a * m + i] * A[a * m + i];
          beta += A[b * m + i] * A[b * m + i];
          gamma += A[a * m + i] * A[b * m + i];
        }
        if (gamma == 0.0) continue;
        const double denom = sqrt(alpha * beta);
        if (denom > 0 && fabs(gamma) / denom > off) off = fabs(gamma) / denom;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) /
                         (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < m; ++i) {
          const double x = A[a * m + i], y = A[b * m + i];
          A[a * m + i] = cs * x - sn * y;
          A[b * m + i] = sn * x + cs * y;
        }
        for (int i = 0; i < k; ++i) {
          const double x = V[a * k + i], y = V[b * k + i];
          V[a * k + i] = cs * x - sn * y;
          V[b * k + i] = sn * x + cs * y;
        }
      }
    if (off < 1e-15) break;
  }
}

/* ---- a9: Problem::projectToManifold, src/CORA_problem.cpp:905-934 ------
 * Stiefel rows: StiefelProduct::projectToManifold src/StiefelProduct.cpp:8-36
 *   thin SVD of the p x d block A_i^T = U S V^T  ->  U V^T  (the polar factor)
 * oblique rows: ObliqueManifold::projectToManifold src/ObliqueManifold.cpp:6-14
 *   normalise each row;  translations: copied.  `out` may alias A. */
void orc_project_manifold(int d, int n, int r, int N, int p, const double *A,
                          int lda, double *out, int ldo) {
  double B[64 * 4], V[16];
  for (int i = 0; i < n; ++i) {
    const int r0 = i * d;
    /* B = A_i^T  (p x d, col-major ld p): column a = row r0+a of A */
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < p; ++c) B[a * p + c] = AT(A, lda, r0 + a, c);
    jacobi_svd_small(p, d, B, V);
    /* normalise columns of B -> U */
    for (int a = 0; a < d; ++a) {
      double s = 0;
      for (int c = 0; c < p; ++c) s += B[a * p + c] * B[a * p + c];
      s = sqrt(s);
      if (s > 0)
        for (int c = 0; c < p; ++c) B[a * p + c] /= s;
    }
    /* (U V^T) is p x d; row layout block = (U V^T)^T = V U^T  (d x p) */
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < p; ++c) {
        double s = 0;
        for (int b = 0; b < d; ++b) s += V[b * d + a] * B[b * p + c];
        AT(out, ldo, r0 + a, c) = s;
      }
  }
  const int dn = d * n;
  for (int j = dn; j < dn + r; ++j) {
    double s = 0;
    for (int c = 0; c < p; ++c) s += AT(A, lda, j, c) * AT(A, lda, j, c);
    s = sqrt(s);
    for (int c = 0; c < p; ++c)
      AT(out, ldo, j, c) = (s > 0) ? AT(A, lda, j, c) / s : AT(A, lda, j, c);
  }
  if (out != A || ldo != lda)
    for (int j = dn + r; j < N; ++j)
      for (int c = 0; c < p; ++c) AT(out, ldo, j, c) = AT(A, lda, j, c);
}

/* ---- Problem::retract, src/CORA_problem.cpp:936-938 -------------------- */
void orc_retract(int d, int n, int r, int N, int p, const double *Y, int ldy,
                 const double *V, int ldv, double *out, int ldo) {
  for (int c = 0; c < p; ++c)
    for (int i = 0; i < N; ++i)
      AT(out, ldo, i, c) = AT(Y, ldy, i, c) + AT(V, ldv, i, c);
  orc_project_manifold(d, n, r, N, p, out, ldo, out, ldo);
}

/* ---- a10 (Jacobi branch): Problem::precondition ------------------------
 * src/CORA_problem.cpp:888-889 with the preconditioner built at :616-618
 * (diag(Q)^-1), followed by the tangent projection the solver's closure
 * applies (src/CORA.cpp:86-92). dinv = 1/diag(Q). */
void orc_precond_jacobi(int d, int n, int r, int N, int p, const double *dinv,
                        const double *Y, int ldy, const double *V, int ldv,
                        double *out, int ldo) {
  for (int c = 0; c < p; ++c)
    for (int i = 0; i < N; ++i) AT(out, ldo, i, c) = dinv[i] * AT(V, ldv, i, c);
  orc_tangent_proj(d, n, r, N, p, Y, ldy, out, ldo, out, ldo);
}

void orc_diag(int N, const int32_t *rowptr, const int32_t *col,
              const double *val, double *diag) {
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
    for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q)
      if (col[q] == i) s += val[q];
    diag[i] = s;
  }
}

/* =========================================================================
 * Sparse Cholesky (CHOLMOD stand-in for the oracle).
 *
 * The reference factors (Q + lambda I)[0:N-1, 0:N-1] with
 * Eigen::CholmodDecomposition (include/CORA/CORA_preconditioners.h:24-26,
 * src/CORA_preconditioners.cpp:16-83) and tests S + eta I for positive
 * definiteness with Eigen::CholmodSupernodalLLT (src/CORA_utils.cpp:36-51).
 * CHOLMOD (SuiteSparse, system package, unpinned: CMakeLists.txt:73) is not
 * in /root/reference; this restates the published up-looking sparse Cholesky
 * (T. A. Davis, "Direct Methods for Sparse Linear Systems", SIAM 2006, ch. 4:
 * elimination tree, ereach, up-looking numeric factorisation).  The result of
 * a solve is independent of the fill-reducing ordering up to rounding, and
 * "factorisation succeeds" <=> "matrix is numerically positive definite", so
 * any correct LL^T reproduces the reference's observable behaviour.
 *
 * Input: symmetric matrix in CSR (full pattern, both triangles), order n,
 * and a permutation perm (new -> old) or NULL for identity.
 * ========================================================================= */
typedef struct {
  int n;
  int32_t *Lp;  /* column pointers of L (CSC), n+1 */
  int32_t *Li;  /* row indices */
  double *Lx;   /* values; first entry of each column is the diagonal */
  int32_t *perm; /* new -> old */
  int32_t *iperm;
  long long nnzL;
} orc_chol;

void orc_chol_free(orc_chol *F) {
  if (!F) return;
  free(F->Lp); free(F->Li); free(F->Lx); free(F->perm); free(F->iperm);
  free(F);
}

long long orc_chol_nnz(const orc_chol *F) { return F->nnzL; }

/* Returns NULL if the matrix is not positive definite (pivot <= 0 or NaN),
 * mirroring `MChol.info() != Eigen::Success`. */
orc_chol *orc_chol_factor(int n, const int32_t *Ap, const int32_t *Ai,
                          const double *Ax, const int32_t *perm_in) {
  orc_chol *F = (orc_chol *)calloc(1, sizeof(orc_chol));
  F->n = n;
  F->perm = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  F->iperm = (int32_t *)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) F->perm[i] = perm_in ? perm_in[i] : i;
  for (int i = 0; i < n; ++i) F->iperm[F->perm[i]] = i;

  /* C = upper triangle of P A P^T in CSC == for each new column k the entries
   * (i <= k).  Because A is symmetric, column k of the upper triangle of C is
   * row perm[k] of A restricted to iperm[col] <= k. */
  int32_t *Cp = (int32_t *)calloc((size_t)n + 1, sizeof(int32_t));
  for (int k = 0; k < n; ++k) {
    const int old = F->perm[k];
    int cnt = 0;
    for (int32_t q = Ap[old]; q < Ap[old + 1]; ++q)
      if (F->iperm[Ai[q]] <= k) ++cnt;
    Cp[k + 1] = Cp[k] + cnt;
  }
  int32_t *Ci = (int32_t *)malloc(sizeof(int32_t) * (size_t)(Cp[n] > 0 ? Cp[n] : 1));
  double *Cx = (double *)malloc(sizeof(double) * (size_t)(Cp[n] > 0 ? Cp[n] : 1));
  for (int k = 0; k < n; ++k) {
    const int old = F->perm[k];
    int32_t w = Cp[k];
    for (int32_t q = Ap[old]; q < Ap[old + 1]; ++q) {
      const int i = F->iperm[Ai[q]];
      if (i <= k) { Ci[w] = i; Cx[w] = Ax[q]; ++w; }
    }
  }

  /* elimination tree (Davis, cs_etree) */
  int32_t *parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t *anc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int k = 0; k < n; ++k) {
    parent[k] = -1; anc[k] = -1;
    for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) {
      int i = Ci[q];
      while (i != -1 && i < k) {
        const int inext = anc[i];
        anc[i] = k;
        if (inext == -1) parent[i] = k;
        i = inext;
      }
    }
  }
  /* column counts by a symbolic up-looking pass (ereach per row) */
  int32_t *colcount = (int32_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int32_t));
  int32_t *flag = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int k = 0; k < n; ++k) flag[k] = -1;
  for (int k = 0; k < n; ++k) {
    flag[k] = k;
    colcount[k]++; /* diagonal */
    for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) {
      int i = Ci[q];
      while (i != -1 && i < k && flag[i] != k) {
        colcount[i]++; /* L(k,i) is nonzero */
        flag[i] = k;
        i = parent[i];
      }
    }
  }
  F->Lp = (int32_t *)malloc(sizeof(int32_t) * ((size_t)n + 1));
  long long tot = 0;
  for (int k = 0; k < n; ++k) {
    if (tot > 2000000000LL) { tot = -1; break; }
    F->Lp[k] = (int32_t)tot; tot += colcount[k];
  }
  if (tot < 0 || tot > 2000000000LL) { /* too much fill for int32 indexing */
    free(Cp); free(Ci); free(Cx); free(parent); free(anc); free(colcount);
    free(flag); free(stack); orc_chol_free(F); return NULL;
  }
  F->Lp[n] = (int32_t)tot;
  F->nnzL = tot;
  F->Li = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
  F->Lx = (double *)malloc(sizeof(double) * (size_t)(tot > 0 ? tot : 1));
  int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  double *x = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
  for (int k = 0; k < n; ++k) next[k] = F->Lp[k];
  for (int k = 0; k < n; ++k) flag[k] = -1;

  int ok = 1;
  for (int k = 0; k < n && ok; ++k) {
    /* ereach: nonzero pattern of row k of L, in topological order */
    int top = n;
    flag[k] = k;
    for (int32_t q = Cp[k]; q < Cp[k + 1]; ++q) {
      int i = Ci[q];
      x[i] = Cx[q];
      if (i == k) continue;
      int len = 0;
      while (flag[i] != k) {
        stack[len++] = i;
        flag[i] = k;
        i = parent[i];
      }
      while (len > 0) stack[--top] = stack[--len];
    }
    double dk = x[k];
    x[k] = 0.0;
    /* duplicates on the diagonal/off-diagonal are assumed pre-summed */
    for (; top < n; ++top) {
      const int i = stack[top];
      const double lki = x[i] / F->Lx[F->Lp[i]];
      x[i] = 0.0;
      for (int32_t q = F->Lp[i] + 1; q < next[i]; ++q)
        x[F->Li[q]] -= F->Lx[q] * lki;
      dk -= lki * lki;
      const int32_t w = next[i]++;
      F->Li[w] = k;
      F->Lx[w] = lki;
    }
    if (!(dk > 0.0)) { ok = 0; break; }
    const int32_t w = next[k]++;
    F->Li[w] = k;
    F->Lx[w] = sqrt(dk);
  }
  free(Cp); free(Ci); free(Cx); free(parent); free(anc); free(colcount);
  free(flag); free(stack); free(next); free(x);
  if (!ok) { orc_chol_free(F); return NULL; }
  return F;
}

/* Solve A X = B for k right-hand sides (col-major); X may alias B. */
void orc_chol_solve(const orc_chol *F, const double *B, int ldb, int k,
                    double *X, int ldx) {
  const int n = F->n;
  double *y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int c = 0; c < k; ++c) {
    for (int i = 0; i < n; ++i) y[i] = AT(B, ldb, F->perm[i], c);
    for (int j = 0; j < n; ++j) { /* L y = b */
      y[j] /= F->Lx[F->Lp[j]];
      const double yj = y[j];
      for (int32_t q = F->Lp[j] + 1; q < F->Lp[j + 1]; ++q)
        y[F->Li[q]] -= F->Lx[q] * yj;
    }
    for (int j = n - 1; j >= 0; --j) { /* L^T x = y */
      double s = y[j];
      for (int32_t q = F->Lp[j] + 1; q < F->Lp[j + 1]; ++q)
        s -= F->Lx[q] * y[F->Li[q]];
      y[j] = s / F->Lx[F->Lp[j]];
    }
    for (int i = 0; i < n; ++i) AT(X, ldx, F->perm[i], c) = y[i];
  }
  free(y);
}

/* ---- a10 (Cholesky branches): blockCholeskySolve ------------------------
 * src/CORA_preconditioners.cpp:46-83: solve the leading F->n rows, and if the
 * right-hand side has exactly one more row set that last row to zero
 * (:78-79).  Followed by the tangent projection of src/CORA.cpp:86-92. */
void orc_precond_chol(const orc_chol *F, int d, int n, int r, int N, int p,
                      const double *Y, int ldy, const double *V, int ldv,
                      double *out, int ldo) {
  orc_chol_solve(F, V, ldv, p, out, ldo);
  if (N == F->n + 1)
    for (int c = 0; c < p; ++c) AT(out, ldo, N - 1, c) = 0.0;
  orc_tangent_proj(d, n, r, N, p, Y, ldy, out, ldo, out, ldo);
}
