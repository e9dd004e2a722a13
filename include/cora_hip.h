/*
 * cora_hip.h -- C ABI of the MI355X-native CORA solver core (libcora_hip.so).
 *
 * The reference (MarineRoboticsGroup/cora) has no FFI seam: its hot path is the
 * set of `const` methods of `CORA::Problem` that `solveCORA` wraps in
 * std::function closures for the TNT optimizer (src/CORA.cpp:52-92,119-122)
 * and for LOBPCG (src/CORA_utils.cpp:83).  Each entry point below replaces one
 * of those methods; the citation is the reference file:line it stands in for.
 *
 * Conventions
 *   - every call returns a cora_status (0 = ok); no exception crosses the ABI;
 *     cora_last_error() gives the message (replaces MatrixShapeException /
 *     std::runtime_error, include/CORA/CORA_types.h:23-39).
 *   - host matrices are column-major double with an explicit leading dimension
 *     (zero-copy from Eigen::MatrixXd::data(), include/CORA/CORA_types.h:48);
 *     Q is row-major CSR with int32 indices (Eigen::SparseMatrix<double,
 *     RowMajor,int> outerIndexPtr/innerIndexPtr/valuePtr after makeCompressed(),
 *     include/CORA/CORA_types.h:70).
 *   - variable layout (include/CORA/CORA_problem.h:151-157): rows [0,d*n) are n
 *     stacked d x p Stiefel blocks, rows [d*n, d*n+r) are r unit rows, rows
 *     [d*n+r, N) are the n+l translations;  N = n(d+1) + l + r.
 *   - the caller owns host buffers; the library owns all device memory tied to
 *     the handle.  One handle = one HIP stream; a handle is not thread-safe,
 *     independent handles are.
 *   - `_dev` entry points take DEVICE pointers to the library's resident layout:
 *     row-major  rows x ld  doubles with ld = cora_ld(ctx) = cora_ld_for(k) = the number of
 *     columns k itself for every k in 2..24 (no padding columns; ld = 2 for k = 1) -- ask, do not
 *     assume -- and rows = cora_rows(ctx) in the library's internal row order (a permutation even
 *     on one GPU: cora_row_map).  Use cora_upload / cora_download to convert from / to the host layout.
 *   - there is no CPU fallback: every compute entry point fails with
 *     CORA_ERR_HIP when no gfx950 device is usable.
 */
#ifndef CORA_HIP_H_
#define CORA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cora_ctx cora_ctx;

typedef enum {
  CORA_OK = 0,
  CORA_ERR_SHAPE = 1,     /* checkMatrixShape failure */
  CORA_ERR_NOT_READY = 2, /* set_point / precond_setup missing (checkUpToDate) */
  CORA_ERR_NAN = 3,       /* NaN guard, src/CORA_problem.cpp:898-901 */
  CORA_ERR_HIP = 4,
  CORA_ERR_ARG = 5,
  CORA_ERR_NOMEM = 6
} cora_status;

/* Preconditioner kinds, include/CORA/CORA_types.h:77 */
typedef enum {
  CORA_PRECOND_NONE = 0,
  CORA_PRECOND_JACOBI = 1,
  CORA_PRECOND_BLOCK_CHOLESKY = 2,
  CORA_PRECOND_REGULARIZED_CHOLESKY = 3
} cora_precond_kind;

/* ---------------------------------------------------------------- handle */

/* Builds a device-resident problem from the assembled data matrix
 * (`Problem::data_matrix_` after updateProblemData(), src/CORA_problem.cpp:
 * 500-510, 625-712).  n_trans = n_poses + n_landmarks. */
int cora_ctx_create(int device, int d, int n_poses, int n_ranges, int n_trans,
                    const int32_t *rowptr, const int32_t *colidx,
                    const double *vals, cora_ctx **out);

/* Same, but the handle owns only the rows of partition `rank` of `world`
 * (pose-aligned, nnz-balanced row partition).  All ranks must pass
 * the same full matrix.  world == 1 is identical to cora_ctx_create. */
int cora_ctx_create_part(int device, int d, int n_poses, int n_ranges,
                         int n_trans, const int32_t *rowptr,
                         const int32_t *colidx, const double *vals, int rank,
                         int world, cora_ctx **out);
/* The same with options.  CORA_PART_WHOLE_LONG_ROWS: keep the long (landmark) rows whole on their owner, as rounds 1-2
 * did -- the owner then reads ~10^4 remote rows of X per landmark (cora_remote_rows), but a handle WITHOUT communication
 * computes complete rows of its shard by itself (the caller keeps the remote rows current).  Default (flags = 0): the
 * long rows are distributed (cora_long_rows) and need the all-reduce of cora_comm_create_* / cora_set_comm. */
#define CORA_PART_WHOLE_LONG_ROWS 1u
int cora_ctx_create_part_opts(int device, int d, int n_poses, int n_ranges, int n_trans, const int32_t *rowptr,
                              const int32_t *colidx, const double *vals, int rank, int world, unsigned flags,
                              cora_ctx **out);

void cora_ctx_destroy(cora_ctx *ctx);

/* Message of the last failed call on `ctx` (ctx may be NULL for create). */
const char *cora_last_error(const cora_ctx *ctx);

/* Problem::setRank / incrementRank, include/CORA/CORA_problem.h:327-334.
 * Invalidates the current point and the preconditioner state. */
int cora_set_rank(cora_ctx *ctx, int p);
int cora_get_rank(const cora_ctx *ctx);

/* Use an externally created hipStream_t (e.g. torch's current stream). */
int cora_set_stream(cora_ctx *ctx, void *hip_stream);

/* Layout queries for the `_dev` API. */
int cora_ld(const cora_ctx *ctx);            /* row stride (doubles) */
int cora_ld_for(int k);                      /* row stride used for k columns */
int64_t cora_rows(const cora_ctx *ctx);      /* rows of a resident vector */
int64_t cora_shard_rows(const cora_ctx *ctx);  /* rows owned per rank (padded) */
int64_t cora_shard_begin(const cora_ctx *ctx); /* first owned internal row */
int64_t cora_nnz(const cora_ctx *ctx);
int64_t cora_dim(const cora_ctx *ctx);       /* N */
/* Row permutation: internal row of API row i (length N, int32). */
int cora_row_map(const cora_ctx *ctx, int32_t *api_to_internal);
/* Partitioned handles: the internal rows OUTSIDE this rank's shard that its rows of Q reference, ascending
 * (the halo of the pose chain plus whatever couples to remote landmarks).  Only these rows of an operand
 * have to arrive before a product, so the exchange step can all-gather them instead of whole shards
 * (cora_amd/dist.py).  rows may be NULL to query the count. */
int cora_remote_rows(const cora_ctx *ctx, int32_t *rows, int64_t *count);
/* Partitioned handles: the DISTRIBUTED long rows (API row indices, the same list on every rank; empty on one GPU).
 * The columns of such a row -- a landmark's translation row -- span every rank, so every rank multiplies the part of
 * the row on the columns it owns and the partial sums are added over the ranks after the product (by the library with
 * cora_comm_create_* / cora_set_comm); the row's owner then holds the result.  A rank therefore asks for no remote rows
 * of X on behalf of these rows: cora_remote_rows() is the chain halo plus the landmarks' own rows. */
int cora_long_rows(const cora_ctx *ctx, int32_t *api_rows, int64_t *count);

/* Statistics of the device format: [0] slices, [1] padded nnz stored in
 * slices, [2] nnz in long rows, [3] long rows, [4] long-row chunks,
 * [5] local rows, [6] local nnz, [7] max slice width. */
int cora_format_stats(const cora_ctx *ctx, int64_t stats[8]);

/* Bytes of the device format of this handle's share of Q -- what ONE product reads of the matrix whatever the operand:
 * [0] values, [1] column indices and per-lane descriptors, [2] slice / chunk descriptors, row lists and head blocks,
 * [3] their sum.  (The reference's CSR, src/CORA_problem.cpp:742-746, is 12 nnz + 4 (N + 1) bytes.) */
int cora_format_bytes(const cora_ctx *ctx, int64_t bytes[4]);

/* ------------------------------------------- host-pointer operator API */

/* Problem::dataMatrixProduct (Explicit), src/CORA_problem.cpp:742-746.
 * X, out: N x k. */
int cora_data_matrix_product(cora_ctx *ctx, const double *X, int ldx, int k,
                             double *out, int ldo);

/* Problem::evaluateObjective, src/CORA_problem.cpp:759-762. Y: N x p. */
int cora_evaluate_objective(cora_ctx *ctx, const double *Y, int ldy, double *f);

/* Problem::Euclidean_gradient, src/CORA_problem.cpp:764-770. */
int cora_euclidean_gradient(cora_ctx *ctx, const double *Y, int ldy,
                            double *out, int ldo);

/* Problem::Riemannian_gradient(Y), src/CORA_problem.cpp:772-780. */
int cora_riemannian_gradient(cora_ctx *ctx, const double *Y, int ldy,
                             double *out, int ldo);

/* Problem::tangent_space_projection(Y, Ydot), src/CORA_problem.cpp:782-820. */
int cora_tangent_space_projection(cora_ctx *ctx, const double *Y, int ldy,
                                  const double *V, int ldv, double *out,
                                  int ldo);

/* Problem::Riemannian_Hessian_vector_product(Y, nablaF_Y, dotY),
 * src/CORA_problem.cpp:822-867. */
int cora_riemannian_hessian_vector_product(cora_ctx *ctx, const double *Y,
                                           int ldy, const double *nablaF_Y,
                                           int ldg, const double *dotY,
                                           int ldd, double *out, int ldo);

/* Problem::projectToManifold, src/CORA_problem.cpp:905-934. */
int cora_project_to_manifold(cora_ctx *ctx, const double *A, int lda,
                             double *out, int ldo);

/* Problem::retract(Y, V), src/CORA_problem.cpp:936-938. */
int cora_retract(cora_ctx *ctx, const double *Y, int ldy, const double *V,
                 int ldv, double *out, int ldo);

/* Problem::updatePreconditioner, src/CORA_problem.cpp:512-623.
 *  JACOBI: diag(Q)^-1 (:616-618), built on device.
 *  *_CHOLESKY: `factor` is an opaque host factorization installed with
 *  cora_precond_set_cholesky (host LL^T + device triangular solves). */
int cora_precond_setup(cora_ctx *ctx, int kind);

/* Install a sparse Cholesky factor L (CSC, int32, diagonal first in each
 * column) of P (Q + lambda I)[0:m,0:m] P^T with m = N or N-1
 * (pin_last_translation, src/CORA_problem.cpp:602-609); perm is new->old. */
int cora_precond_set_cholesky(cora_ctx *ctx, int m, const int32_t *Lp,
                              const int32_t *Li, const double *Lx,
                              const int32_t *perm);

/* Device solve plan of the installed factor: [0] stages (a solve is 4*stages - 2 sparse products),
 * [1] entries of the explicit block inverses, [2] nnz(L), [3] rows of the last stage. */
int cora_precond_stats(const cora_ctx *ctx, int64_t stats[4]);
/* Entries the preconditioner's device solve plan stores: [0] / [1] the last stage's forward / backward product,
 * [2] / [3] entry slots of the substitution blocks' forward / backward sweep (null padding included), [4] substitution
 * blocks, [5] aux rows.  (Measurement: bench.py's bytes per STPCG iteration.) */
int cora_precond_entries(const cora_ctx *ctx, int64_t s[6]);

/* Translation-implicit formulation (Formulation::Implicit, src/CORA_problem.cpp:714-753):
 *   dataMatrixProduct(Y) = Qmain Y - B L^-1 B^T Y,  L L^T = Q33[0:nt-1, 0:nt-1].
 * cora_implicit_set_cholesky installs that factor (CSC, diagonal first per column; perm is
 * new -> old index inside the translation block); cora_set_formulation(ctx, 1) then switches every
 * product (spmm / objective / set_point / hvp) to the implicit operator.  Vectors keep N rows:
 * translation rows of inputs are ignored, translation rows of outputs are zero (callers pass and
 * read the leading d*n + r rows).  cora_certificate_product* always applies the explicit S, like
 * the reference's certify_solution (:1054-1058).  cora_translation_explicit_dev =
 * getTranslationExplicitSolution (:1168-1197): out = [Y; -L^-1 B^T Y; 0].
 * Partitioned handle: every rank installs the WHOLE factor; the two products of an implicit product are partitioned
 * products and the solve between them is replicated (the right-hand side is all-gathered, cora_allgather_fn). */
int cora_implicit_set_cholesky(cora_ctx *ctx, int m, const int32_t *Lp, const int32_t *Li,
                               const double *Lx, const int32_t *perm);
int cora_set_formulation(cora_ctx *ctx, int implicit);
int cora_translation_explicit_dev(cora_ctx *ctx, const double *dY, int k, double *dOut);

/* Problem::precondition(V), src/CORA_problem.cpp:869-903 (no projection;
 * last row zeroed when the factor has N-1 rows, src/CORA_preconditioners.cpp:
 * 78-79; NaN guard -> CORA_ERR_NAN). */
int cora_precondition(cora_ctx *ctx, const double *V, int ldv, double *out,
                      int ldo);

/* Problem::compute_Lambda_blocks(Y), src/CORA_problem.cpp:1105-1131.
 * stiefel: d x (d*n) column-major (ld = d); oblique: r. */
int cora_compute_lambda_blocks(cora_ctx *ctx, const double *Y, int ldy,
                               double *stiefel, double *oblique);

/* Certificate operator  out = (Q - Lambda(Y)) X  for an N x k block, i.e.
 * Problem::get_certificate_matrix(Y) (src/CORA_problem.cpp:1162-1166) applied
 * as the LOBPCG operator (src/CORA_utils.cpp:83).  Uses the Lambda of the
 * current point (cora_set_point). */
int cora_certificate_product(cora_ctx *ctx, const double *X, int ldx, int k,
                             double *out, int ldo);

/* Metric closure <V1, V2> = trace(V1^T V2), src/CORA.cpp:119-122. */
int cora_inner_product(cora_ctx *ctx, const double *A, int lda,
                       const double *B, int ldb, int k, double *out);

/* --------------------------------------------- resident (device) API */

/* Allocate / free a resident vector of cora_rows(ctx) x cora_ld_for(k). */
int cora_dev_alloc(cora_ctx *ctx, int k, double **dptr);
int cora_dev_free(cora_ctx *ctx, double *dptr);

/* Host column-major N x k  <->  resident row-major layout. */
int cora_upload(cora_ctx *ctx, const double *host, int ld, int k, double *dptr);
int cora_download(cora_ctx *ctx, const double *dptr, int k, double *host,
                  int ld);

/* Make Y the current point: caches Y, nablaF = Q Y, f = 1/2 <Y, Q Y>,
 * Lambda(Y) and grad = Proj_Y(nablaF) on the device (the QuadraticModel
 * closure of src/CORA.cpp:58-75 + the objective of :52-55 in one pass). */
int cora_set_point(cora_ctx *ctx, const double *Y, int ldy);
int cora_set_point_dev(cora_ctx *ctx, const double *dY);

/* f(Y) = 1/2 <Y, Q Y> for a resident Y WITHOUT changing the current point (the
 * Objective closure of src/CORA.cpp:52-55, used for trial points). Synchronises. */
int cora_objective_dev(cora_ctx *ctx, const double *dY, double *f);

/* One trust-region trial step of Optimization::Riemannian::TNT (the solver src/CORA.cpp:139-140 calls; its sources
 * are a dependency absent from the reference tree) in ONE wait: dHs = Hess(s), dXprop = Retr_Y(s), and
 * out = { <grad, s>, <s, Hess s>, <s, s>, f(Xprop) }.  Same kernels and values as cora_hvp_dev + cora_dots_dev +
 * cora_retract_dev + cora_objective_dev; the product Q Xprop stays on the handle for cora_tnt_accept_dev. */
int cora_tnt_trial_dev(cora_ctx *ctx, const double *dS, double *dHs, double *dXprop, double out[4]);
/* Make dX the current point (cora_set_point_dev), dPg = precondition(grad) projected, and
 * out = { f, <grad, grad>, <Pg, Pg>, <grad, Pg> } in one wait.  When dX is the vector the last cora_tnt_trial_dev
 * filled (one GPU), Q X is taken from that call: the pointer cora_point_egrad_dev returns changes.
 * CONTRACT of the reuse: the kept product belongs to the CONTENTS dXprop had when cora_tnt_trial_dev returned.  Every entry
 * point of this header that writes a caller's vector (cora_upload, the outputs of every *_dev operation, cora_stpcg_*_dev's
 * work vectors) or can hand its address out again (cora_dev_free / cora_dev_alloc) drops it when that vector is dXprop, and
 * the accept then forms Q X again.  A write the library cannot see (the caller's own kernel, hipMemcpy into dXprop) must be
 * followed by cora_set_point_dev instead of this call. */
int cora_tnt_accept_dev(cora_ctx *ctx, const double *dX, double *dPg, double out[4]);

/* f at the current point (local shard contribution when partitioned). */
int cora_point_cost(cora_ctx *ctx, double *f);
/* Device pointers to the cached point data (valid until the next point is set: cora_set_point*, cora_tnt_accept_dev). */
const double *cora_point_Y_dev(const cora_ctx *ctx);
const double *cora_point_egrad_dev(const cora_ctx *ctx);
const double *cora_point_rgrad_dev(const cora_ctx *ctx);

/* out = Q X (k columns; ld = cora_ld_for(k)). */
int cora_spmm_dev(cora_ctx *ctx, const double *dX, int k, double *dOut);
/* out = Proj_Y((Q - Lambda) X) at the current point: the Hessian-vector
 * product of src/CORA_problem.cpp:822-867 with cached Lambda. */
int cora_hvp_dev(cora_ctx *ctx, const double *dX, double *dOut);
/* out = (Q - Lambda) X, k columns. */
int cora_certificate_product_dev(cora_ctx *ctx, const double *dX, int k,
                                 double *dOut);
/* out = Proj_Y(V) at the current point. */
int cora_tangent_space_projection_dev(cora_ctx *ctx, const double *dV,
                                      double *dOut);
/* out = Proj_Y(precondition(V)): the `precon` closure, src/CORA.cpp:86-92. */
int cora_precondition_projected_dev(cora_ctx *ctx, const double *dV,
                                    double *dOut);
/* out = projectToManifold(Y + alpha V) with Y the current point. */
int cora_retract_dev(cora_ctx *ctx, const double *dV, double alpha,
                     double *dOut);
int cora_project_to_manifold_dev(cora_ctx *ctx, const double *dA, double *dOut);
/* Vector ops on resident N x p vectors (local shard rows when partitioned). */
int cora_axpby_dev(cora_ctx *ctx, double a, const double *dX, double b,
                   double *dY); /* Y = a X + b Y */
/* Y1 += a1 X1 and Y2 += a2 X2 in one launch: the step and residual updates of an STPCG iteration
 * (s += alpha p, r += alpha Hp inside Optimization::Riemannian::TNT, called from src/CORA.cpp:139). */
int cora_axpy2_dev(cora_ctx *ctx, double a1, const double *dX1, double *dY1, double a2, const double *dX2,
                   double *dY2);
/* The inner solver of the trust-region method -- Steihaug-Toint truncated preconditioned CG on
 *   min <grad, s> + 1/2 <s, Hess s>   subject to ||s||_M <= Delta
 * (Optimization::Riemannian::TNT's STPCG, reached from src/CORA.cpp:139-140 with the closures of
 * :58-92,119-122) -- run entirely on the device at the current point: Hess = cora_hvp_dev, M^-1 =
 * cora_precondition_projected_dev, inner products and the scalar recurrences in device memory.  Stops
 * when ||r|| <= ||r0|| min(kappa_fgr, ||r0||^theta), on negative curvature or the trust-region boundary
 * (step to the boundary), or after max_iters Hessian-vector products.  dS receives the step; dR, dV,
 * dP, dHp are work vectors (cora_dev_alloc(ctx, p, ..)).  iters = Hessian-vector products used,
 * step_M_norm = ||s||_M. */
int cora_stpcg_dev(cora_ctx *ctx, const double *dGrad, double Delta, double kappa_fgr, double theta,
                   int max_iters, double *dS, double *dR, double *dV, double *dP, double *dHp, int *iters,
                   double *step_M_norm);
/* The same solve started from a preconditioned gradient the caller already holds: dPg = M^-1 grad (projected, i.e.
 * cora_precondition_projected_dev(grad)), g_g = <grad, grad>, g_Pg = <grad, dPg>.  The trust-region loop computes all
 * three for its stopping tests at every accepted point, so the inner solve needs neither a preconditioner apply nor a
 * reduction of its own to start.  dPg is read only. */
int cora_stpcg_warm_dev(cora_ctx *ctx, const double *dGrad, const double *dPg, double g_g, double g_Pg, double Delta,
                        double kappa_fgr, double theta, int max_iters, double *dS, double *dR, double *dV, double *dP,
                        double *dHp, int *iters, double *step_M_norm);
/* Same for vectors allocated with k columns (cora_dev_alloc(ctx, k, ..)). */
int cora_axpby_cols_dev(cora_ctx *ctx, int k, double a, const double *dX, double b, double *dY);
/* dX (k columns, resident layout) = numbers in (-1, 1) that depend on (seed, variable, column) only -- the same block on
 * every partition of the problem --: a start block for the eigensolver made on the device (the norm estimate of src/CORA_problem.cpp:556-578 starts from Matrix::Random). */
int cora_fill_random_dev(cora_ctx *ctx, int k, unsigned long long seed, double *dX);
int cora_copy_dev(cora_ctx *ctx, const double *dX, int k, double *dY);
int cora_dot_dev(cora_ctx *ctx, const double *dA, const double *dB, int k,
                 double *out); /* synchronises the stream */
/* Up to 4 inner products in one pass / one synchronisation. */
int cora_dots_dev(cora_ctx *ctx, int count, const double *const *dA,
                  const double *const *dB, double *out);

/* Tall-skinny block operations for the eigensolver (Rayleigh-Ritz of LOBPCG, which the reference
 * takes from libs/Optimization: src/CORA_utils.cpp:113-119).  Blocks are resident vectors with ka /
 * kb <= 24 columns.
 *   gram:    G = A^T B -> host, column-major ka x kb (synchronises)
 *   combine: Out = sum_{i<n} X_i C_i, C_i host column-major k_i x kout (ld k_i), n <= 4;
 *            Out must not alias an input. */
int cora_gram_dev(cora_ctx *ctx, const double *dA, int ka, const double *dB, int kb, double *G);
/* n <= 16 products G[e] = A_e^T B_e with one synchronisation: the numbers of n cora_gram_dev calls (same kernel, one
 * piece of the reduction buffer each) -- the twelve Gram blocks of one Rayleigh-Ritz step. */
int cora_gram_batch_dev(cora_ctx *ctx, int n, const double *const *dA, const int *ka, const double *const *dB,
                        const int *kb, double *const *G);
int cora_combine_dev(cora_ctx *ctx, int n, const double *const *dX, const int *k, const double *const *C,
                     int kout, double *dOut);

/* Timing helpers: HIP events on the handle's stream. */
int cora_timer_start(cora_ctx *ctx);
int cora_timer_stop_ms(cora_ctx *ctx, float *ms); /* synchronises */
int cora_sync(cora_ctx *ctx);

/* Test hook: executes the handle's device FORMAT (slices + long rows) on the
 * host, to validate the format conversion where no GPU exists.  Never used by
 * any compute entry point. */
/* A factor of the caller's own for the handle's vectors: (L L^T)^-1 with L (CSC, diagonal first) the Cholesky-form
 * factor of P A P^T, N rows, perm new -> old in API row order.  It may be INCOMPLETE (dropped entries).  Used by
 * fast_verification for the preconditioner of src/CORA_utils.cpp:140-156 (ILDL with pos_def_mod: L |D|^(1/2)).
 * cora_aux_solve_dev: dX = (P^T L L^T P)^-1 dB on k-column resident vectors, dB != dX.
 * Partitioned handle: the factor of the RANK'S OWN rows (m = its row count, perm holds its API rows): the solve is
 * block Jacobi over the ranks, like cora_precond_set_cholesky, and touches the rank's rows only. */
int cora_aux_set_cholesky(cora_ctx *ctx, int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                          const int32_t *perm);
int cora_aux_solve_dev(cora_ctx *ctx, const double *dB, int k, double *dX);

/* ------------------------------------------------ multi-GPU: injected communication
 *
 * A partitioned handle (cora_ctx_create_part, world > 1) owns the rows of its shard; every resident vector still
 * has cora_rows() rows, of which only the shard is kept current.  The library links no collective library: the
 * three steps that need other ranks are callbacks (RCCL through torch.distributed in bench.py, gloo or threads in
 * the tests).  With them installed every entry point of the resident API is collective -- all ranks call the same
 * sequence -- and the C++ host above (TNT, LOBPCG, solveCORA) runs unchanged, one process per GPU:
 *   exchange : make the rows of dX that this rank's part of Q reads (cora_remote_rows) current; called before
 *              every product (src/CORA_problem.cpp:742-746 and its callers), on the handle's stream
 *   allreduce: sum n host doubles over the ranks in place (inner products src/CORA.cpp:119-122, the cost
 *              src/CORA_problem.cpp:759-762, Gram matrices of LOBPCG); must return the SAME bits on every rank
 *   allgather: make every row of dX current on every rank (before a download)
 * Without callbacks a partitioned handle leaves both to the caller (operands current where read, results are
 * per-rank partial sums).  Each returns 0 on success.  Work they enqueue must be ordered after the work already on the handle's stream
 * (they may synchronise it) and complete, or be ordered on that stream, when they return. */
typedef int (*cora_exchange_fn)(void *user, double *dX, int ld);
typedef int (*cora_allreduce_fn)(void *user, double *vals, int n);
typedef int (*cora_allgather_fn)(void *user, double *dX, int ld);
int cora_set_comm(cora_ctx *ctx, cora_exchange_fn exchange, cora_allreduce_fn allreduce, cora_allgather_fn allgather,
                  void *user);
/* on != 0: a collective step on this (partitioned) handle without communication installed fails with
 * CORA_ERR_NOT_READY instead of being left to the caller.  CORA::Problem sets it on the handles it creates for a
 * partition, so that a rebuilt handle cannot silently compute per-rank partial results. */
int cora_require_comm(cora_ctx *ctx, int on);
/* 1 if cora_stpcg_dev / cora_stpcg_warm_dev can run on this handle: always on one GPU; on a partitioned handle with
 * the library's own communication as the ACTIVE transport (cora_comm_create_*: the inner products are all-reduced on
 * the device, no host round trip), the explicit formulation and up to 12 columns.  Preconditioners on a partitioned
 * handle: none, Jacobi, or a Cholesky factor of the rank's OWN diagonal block of Q + lambda I installed with
 * cora_precond_set_cholesky -- block Jacobi over the ranks with exact blocks, NOT the reference's one global factor
 * (a sequential solve does not shard; expect more inner iterations than on one GPU). */
int cora_stpcg_device_ok(const cora_ctx *ctx);
/* The library's OWN communication for a partitioned handle: the three steps above implemented natively, so that no
 * callback (and no Python) sits on the data path.  Two transports:
 *   RCCL : one process per GPU.  Rank 0 calls cora_rccl_unique_id (128 bytes), the launcher hands the bytes to every
 *          rank (torch.distributed broadcast in bench.py, MPI_Bcast, a file), every rank calls cora_comm_create_rccl:
 *          collective.  The exchange is pack kernel -> ncclAllGather -> scatter kernel, the reductions ncclAllReduce,
 *          all on the handle's stream.  librccl.so is opened at run time: single-GPU users do not depend on it.
 *   local: every rank a thread of one process with its own handle (tests on a one-GPU box; several GPUs of one process):
 *          cora_local_group_create(world) once, cora_comm_create_local(handle, group) from every rank's thread.
 * Both replace whatever cora_set_comm installed; the communicator is destroyed with the handle.
 * cora_comm_exchanged_rows: rows a rank receives per exchange (world x the longest export list). */
typedef struct cora_local_group cora_local_group;
int cora_rccl_unique_id(void *id128);
int cora_comm_create_rccl(cora_ctx *ctx, const void *id128);
cora_local_group *cora_local_group_create(int world);
void cora_local_group_destroy(cora_local_group *group);
void cora_local_group_abort(cora_local_group *group); /* a rank failed outside the library: release the others */
int cora_comm_create_local(cora_ctx *ctx, cora_local_group *group);
/* Third transport (round 6; SURVEY 8e mitigation 3: "one-shot direct all-gather"): DEVICE-SIDE collectives over peer-mapped
 * mailboxes, no RCCL and no host on the data path (cora_amd/csrc/p2p.h).  The exchange of a product is under 1 KB per rank
 * and the reductions of an STPCG iteration are 1-3 doubles, so a collective is ONE small kernel on the handle's stream: it
 * writes its payload into every peer's mailbox (stores over xGMI; the mailbox is exported with hipIpcGetMemHandle and mapped
 * by the peers, or shared by pointer between threads of one process), sets a per-(receiver, sender) sequence flag with
 * release at system scope, spins on the flags of its OWN mailbox (with a wall-clock timeout, CORA_P2P_TIMEOUT_S, default 60:
 * a dead peer raises an error count instead of hanging the GPU, and every later collective call of that rank fails with
 * CORA_ERR_HIP instead of computing on what the timed-out one delivered) and delivers -- all-reduces add in rank order, the same bits
 * as the other transports.  Ranks: one process per GPU, several processes sharing a GPU (tests), or threads.
 *   cora_comm_p2p_handle(ctx, blob)   -> creates this rank's mailbox, blob = CORA_P2P_HANDLE_BYTES to hand to the peers
 *   (launcher: all-gather of the blobs in rank order -- torch.distributed, MPI, a file; like the RCCL id)
 *   cora_comm_create_p2p(ctx, blobs)  -> collective: maps the peers, plans the exchange
 * cora_comm_counters stays at 0 + 0 on this transport (no library collective is issued); cora_comm_p2p_status:
 * out[0] collectives, [1] kernels launched for them, [2] timeouts raised, [3] mailbox memory (0 uncached, 1 fine-grained,
 * 2 ordinary; -1 no p2p transport), [4] all-gathers, [5] all-reduces. */
#define CORA_P2P_HANDLE_BYTES 128
int cora_comm_p2p_handle(cora_ctx *ctx, void *blob);
int cora_comm_create_p2p(cora_ctx *ctx, const void *blobs);
int cora_comm_p2p_status(const cora_ctx *ctx, long out[6]);
int64_t cora_comm_exchanged_rows(const cora_ctx *ctx);
/* Rows of resident vectors received through all-gathers of whole shards or of packed pieces so far (library's own
 * communication): the implicit formulation's replicated translation solve gathers the translation rows alone
 * (world x the longest shard's translation rows per product), not whole shards. */
long long cora_comm_gathered_rows(const cora_ctx *ctx);
/* What the handle's RCCL communicator itself reports: out[0] = ncclCommCount, out[1] = ncclCommUserRank (-1, -1 without an
 * RCCL communicator).  cora_comm_create_rccl fails when they differ from the handle's partition. */
int cora_comm_rccl_ranks(const cora_ctx *ctx, int out[2]);
/* Measurement switch: on = 0 leaves the collective steps to the caller again (a product then runs on whatever the
 * remote rows hold: bench.py times the kernel alone this way), on = 1 re-installs the native communication. */
int cora_comm_native_enable(cora_ctx *ctx, int on);
/* Collectives the library's own communication has issued on this handle so far: out[0] all-gathers, out[1] all-reduces.
 * A product is ONE all-gather (the exported rows of the operand and the distributed long rows' partial sums travel
 * together; the owners add the sums up in rank order); an iteration of the device-resident STPCG is that plus two
 * all-reduces (kappa; <r,r> and <r,v>): the two synchronisation points of preconditioned CG. */
int cora_comm_counters(const cora_ctx *ctx, long out[2]);

/* The phases of a partitioned handle's product, timed with HIP events on the handle's stream (library's own
 * communication, serial order; collective: every rank calls it): us[0] pack of the exported rows, [1] the distributed
 * long rows' chunks, [2] the all-gather, [3] unpack (rows scattered, long rows summed), [4] the slices.  epi: 0 Q X,
 * 1 (Q - Lambda) X, 2 the Hessian-vector product.  What a first run on several GPUs is read with (bench.py --gpus N). */
int cora_debug_product_phases(cora_ctx *ctx, const double *dX, double *dOut, int epi, int reps, double us[5]);

/* Test hook (process-wide): the pose slices of a product read X through their LDS windows from this many slices on
 * (default 2 048 = the wavefronts resident at once; below, every wavefront is resident and gathers directly).  Returns
 * the previous value; a negative argument only queries.  Results do not depend on it. */
int cora_debug_spmm_window_min_slices(int min_slices);

/* Timing hook: with on != 0 the products of a partitioned handle skip every collective step (the operand's remote rows
 * are whatever they are, the distributed long rows stay partial sums) -- the kernel alone, for bench.py's roofline leg. */
int cora_debug_local_products(cora_ctx *ctx, int on);

/* With the native communication a product of a partitioned handle overlaps the exchange of its operand with the
 * slices that read this rank's own rows only: exchange (pack, all-gather, scatter) on a second stream, interior
 * slices + long-row chunks at the same time on the handle's stream, boundary slices when the exchange has landed.
 * on = 0: always the serial order (exchange, then one launch); on = 1 (default): the split from 2 048 interior slices
 * per rank on (CORA_EXCHANGE_OVERLAP_MIN_SLICES) -- below, a second launch and two cross-stream dependencies cost more
 * than the interior slices take; on = 2: always the split.  The results are the same numbers either way.
 * cora_comm_overlap_active: 1 when products of this handle take the overlapped form. */
int cora_comm_overlap_enable(cora_ctx *ctx, int on);
int cora_comm_overlap_active(const cora_ctx *ctx);
/* Building blocks of an exchange, on the handle's stream: dPacked[k] = dX[rows[k]], dX[rows[k]] = dPacked[k],
 * dDst[rows[k]] = dSrc[rows[k]] (rows: device array of n internal rows; ld = row stride in doubles). */
int cora_pack_rows_dev(cora_ctx *ctx, const double *dX, int ld, const int32_t *d_rows, int64_t n, double *dPacked);
int cora_scatter_rows_dev(cora_ctx *ctx, const double *dPacked, int ld, const int32_t *d_rows, int64_t n, double *dX);
int cora_copy_rows_dev(cora_ctx *ctx, const double *dSrc, int ld, const int32_t *d_rows, int64_t n, double *dDst);
/* dDst[rows of shard `shard`] = dSrc[the same rows] (one contiguous block of cora_shard_rows() rows) */
int cora_copy_shard_dev(cora_ctx *ctx, const double *dSrc, int ld, int shard, double *dDst);
int cora_rank(const cora_ctx *ctx);
int cora_world(const cora_ctx *ctx);

/* Measurement hook (bench.py): HIP event pairs around the Hessian-vector product of every iteration of
 * cora_stpcg_dev, i.e. the product as it runs INSIDE the solver loop, with the preconditioner's traffic between
 * two of them (the back-to-back figure keeps Q in the Infinity Cache). */
int cora_debug_profile_stpcg(cora_ctx *ctx, int on);
int cora_debug_stpcg_hvp_us(cora_ctx *ctx, double *mean_us, int *count);
/* on = 2: events around EVERY launch of the sweep-fused iteration (one GPU).  us[k], mean over the iterations of the
 * last cora_stpcg_dev that ran: 0 product with the kappa partials | 1 kappa | 2 forward sweep | 3, 4 the last stage's
 * two products | 5 backward sweep | 6 two events in a row with nothing between them (what an event costs the stream:
 * every figure carries it); -1 where nothing was recorded.  us[7] = 1 when kappa had no launch of its own (folded into the
 * forward sweep: us[1] is then another measurement of the event's cost). */
int cora_debug_stpcg_phase_us(cora_ctx *ctx, double us[8]);
/* Form of the iteration the last cora_stpcg_dev ran: 0 one pass per operation, 1 fused vector passes, 2 vector passes
 * fused into the sweeps of the Cholesky solve (tests pin which form they compare). */
int cora_debug_stpcg_path(const cora_ctx *ctx);
/* OPT-IN (CORA_STPCG_GRAPH=1; off by default -- measured slower on this part, see stpcg_run in capi.hip): batches of
 * device-resident STPCG iterations captured once as a hipGraph and replayed (one GPU, fused forms).
 * out[0] = graphs captured so far, out[1] = batches replayed (both 0 unless the switch is on). */
int cora_debug_stpcg_graph(const cora_ctx *ctx, long out[2]);
/* EVERY environment variable the library reads (round 6: one table; `grep -rn 'getenv("CORA_' cora_amd/csrc` finds the same
 * names and no others).  NONE changes what is computed beyond the rounding of a different (equally valid) order of operations;
 * the two lab switches that produced wrong results (CORA_LAB_SKIP_RANGE_SLICES, the CORA_SUB_F32 upload) are no longer in the
 * product library: they exist only in builds with -DCORA_LAB_BUILD / -DCORA_SUB_F32=1.
 *
 * Forms of the STPCG iteration (all tested against each other, tests/test_gpu_solver.py; read per solve):
 *   CORA_NO_FUSE=1            one pass per operation instead of the fused vector passes
 *   CORA_NO_SWEEP_FUSE=1      vector passes are not folded into the sweeps of the two-stage Cholesky solve
 *   CORA_NO_INVERSE_FUSE=1    ... nor into the two products of a one-explicit-inverse plan
 *   CORA_NO_RESIDUAL_SLOTS=1  the one-explicit-inverse iteration finishes <r, r> with a ticket in its residual pass
 *   CORA_NO_KAPPA_FOLD=1, CORA_KAPPA_FOLD_MAX=n (4096)   kappa = <p, Hp> gets a launch of its own (always | above n partials)
 *   CORA_NO_TNT_FUSE=1        cora_tnt_accept_dev forms Q X again instead of taking the trial's
 * Host loop of cora_stpcg_dev (same numbers, tests run one problem under each):
 *   CORA_STPCG_DEPTH=0|1      the host waits for every iteration | runs one whole iteration ahead (default: the next
 *                             iteration's product ahead on small problems, 0 at 10^5 poses and above)
 *   CORA_STPCG_AHEAD=0|1      forces the product-ahead form off | on
 *   CORA_STPCG_BATCH=n        n iterations enqueued, all waited for (the form partitioned handles always use)
 *   CORA_STPCG_GRAPH=1        opt-in hipGraph replay of batches (measured slower on this part)
 * Partitioned handles:
 *   CORA_NO_EXCHANGE_OVERLAP=1, CORA_EXCHANGE_OVERLAP_MIN_SLICES=n (2048)   interior slices beside the exchange: never | from n
 *                             interior slices per rank
 *   CORA_IMPLICIT_WHOLE_GATHER=1   the implicit formulation gathers whole shards instead of the packed translation rows
 * Format of Q and the product's launch (same products to rounding; bit-identical where the order of sums is unchanged):
 *   CORA_CHAIN_SLICES=0       pose slices in the plain layout (all columns explicit)
 *   CORA_SLICE_LJF=...        order of the slices inside an XCD's range (longest first)
 *   CORA_SPMM_EXTRA_LDS=bytes, CORA_SPMM_WINDOW_MIN_SLICES=n   occupancy / LDS-window experiments of k_spmm
 *   CORA_FORMAT_THREADS=n, CORA_FORMAT_TIMING=1   host threads of the format builder; its phase times on stderr
 * Solve plan of a Cholesky factor (trisolve_build.cpp; every plan solves the same system, tests/test_trisolve_cpu.py):
 *   CORA_TRI_SUB=0|1          force the explicit-stage form | the substitution-block form
 *   CORA_TRI_SN_CAP=n         rows per supernode whose diagonal block is inverted (default: a pose)
 *   CORA_TRI_UNFOLD_MIN=n     aux sums folded into the last stage's first product below n extra entries
 *   CORA_TRI_LEVEL_CAP=x      (round 6; default 0 = off: measured, no gain) subtrees taller than x times the median block are
 *                             not taken whole as solve blocks
 *   CORA_TRI_THREADS=n, CORA_TRI_CHECK_ETREE=1, CORA_TRI_TIMING=1   builder threads; elimination tree computed both ways
 *                             and compared; phase times of set-up on stderr
 *   CORA_SUB_IO_LISTS=1, CORA_IO_STATS=1   row I/O of the sweeps from index lists instead of run tables; run statistics
 * Host factorisation and ordering (sparse_cholesky.cpp, CORA_problem.cpp; bit-identical factors in every setting):
 *   CORA_CHOL_THREADS=n, CORA_SYMBOLIC_THREADS=n, CORA_CHOL_NO_SYMBOLIC_CACHE=1, CORA_CHOL_NO_TRAILING_GROUP=1
 *   CORA_ND_LEAF=n            poses per leaf of the nested dissection (2)
 *   CORA_REG_CHOLESKY_MAX_COND=x   kappa_max of the regularised preconditioner (1e6, src/CORA_problem.cpp:551)
 * solveCORA (host/CORA.cpp, CORA_utils.cpp):
 *   CORA_NO_CERT_PREPARE=1    the first certification's pattern work is not prepared beside the first TNT solve
 *   CORA_NO_CERT_SPECULATION=1   the eigensolver's first stage does not run beside the host's PSD test
 *   CORA_NO_PIVOT_SEED=1      (round 6) fast_verification keeps the reference's plain order after a failed factorisation --
 *                             LOBPCG from the bootstrap block, then the ILDL branch (src/CORA_utils.cpp:112-167) -- instead of
 *                             seeding the block with the failed pivot's direction of non-positive curvature
 *   CORA_TRACE_BITS=1         the bits of every stage of the staircase on stderr (determinism bisection) */

int cora_debug_format_spmm_host(const cora_ctx *ctx, const double *X, int ldx,
                                int k, double *out, int ldo);

/* Test hook: builds the device solve plan (stages, explicit block inverses) of a Cholesky factor
 * L (CSC, diagonal first per column, m x m) and executes its products on the host, in launch order:
 * X = (L L^T)^-1 B for k column-major right-hand sides.  stats: [0] stages, [1] entries of the
 * block inverses, [2] nnz(L), [3] dense stage-0 blocks.  Never used by any compute entry point. */
int cora_debug_factor_solve_host(int m, const int32_t *Lp, const int32_t *Li, const double *Lx, int k,
                                 const double *B, double *X, int64_t stats[4]);

#ifdef __cplusplus
}
#endif
#endif /* CORA_HIP_H_ */
